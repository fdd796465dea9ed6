// engine.cpp -- see engine.hpp.  Formulas cite /root/reference/src/osqppurepy/_osqp.py ("_osqp.py:LINE"), the only
// in-tree statement of the algorithm the reference's C core executes.
#include "engine.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <thread>
#include <limits>

namespace osqp_hip {

namespace {
constexpr double kRhoMin = 1e-6, kRhoMax = 1e6, kRhoTol = 1e-4;   // _osqp.py:25-28 (RHO_EQ_OVER_RHO_INEQ = 1e3 is applied in the set_rho kernel)
constexpr double kMinScaling = 1e-4, kMaxScaling = 1e4;                                 // _osqp.py:44-45
const double kNaN = std::numeric_limits<double>::quiet_NaN();

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
double limit_scaling(double v) { return v < kMinScaling ? 1.0 : (v > kMaxScaling ? kMaxScaling : v); }   // _osqp.py:363-387
double clamp_rho(double r) { return std::min(std::max(r, kRhoMin), kRhoMax); }

template <class T>
T *dev_vec(Dev &d, size_t count) { return static_cast<T *>(be::alloc(d, std::max<size_t>(count, 1) * sizeof(T))); }

// Row blocks for the CSR-stream kernels: consecutive rows whose nnz sum to <= target (and <= kMaxRowsPerBlock rows);
// a row with more than kLongRow entries is a block of its own (reduced by the whole workgroup).
std::vector<int> build_row_blocks_target(const std::vector<int> &rowptr, int nrows, int target) {
  std::vector<int> rb; rb.push_back(0);
  int r = 0;
  while (r < nrows) {
    int len = rowptr[r + 1] - rowptr[r];
    if (len > kLongRow) { r++; rb.push_back(r); continue; }
    int start = r, acc = 0;
    while (r < nrows && r - start < kMaxRowsPerBlock) {
      int l2 = rowptr[r + 1] - rowptr[r];
      if (l2 > kLongRow || (acc + l2 > target && r > start)) break;
      acc += l2; r++;
    }
    rb.push_back(r);
  }
  return rb;
}
// Every kernel runs kGrid workgroups, so the number of row blocks is made a whole multiple k of kGrid with equal
// nnz per block (a 1172-block matrix on a 1024-workgroup grid would otherwise cost two full rounds).
// cap: most entries of a block (kChunk for the CSR-stream kernels; kF1Chunk when the one-launch PCG form is wanted: a workgroup then
// takes several blocks per launch on large problems)
std::vector<int> build_row_blocks(const std::vector<int> &rowptr, int nrows, int cap = kChunk) {
  const long nnz = nrows > 0 ? rowptr[nrows] : 0;
  long k = std::max<long>(1, (nnz + (long)kGrid * cap - 1) / ((long)kGrid * cap));
  for (;; k++) {
    int target = (int)std::max<long>(128, (nnz + kGrid * k - 1) / (kGrid * k));
    for (int attempt = 0; attempt < 40 && target <= cap; attempt++) {
      std::vector<int> rb = build_row_blocks_target(rowptr, nrows, target);
      if ((long)rb.size() - 1 <= kGrid * k) return rb;
      target = std::min<int>(cap + 1, target + std::max(1, target / 50));
    }
    if (k > 1024) return build_row_blocks_target(rowptr, nrows, cap);   // pathological (e.g. all rows long): accept
  }
}
// block descriptors {first row, end row, first nnz, end nnz}; long rows also get their run table (see DevCsr::runinfo).  Slices are the
// fixed kChunk steps the kernels take from the row's first entry
std::vector<int> block_descs(const std::vector<int> &rb, const std::vector<int> &rp, const std::vector<int> &cj, std::vector<int> &runs) {
  std::vector<int> d; d.reserve(4 * rb.size());
  runs.clear();
  for (size_t b = 0; b + 1 < rb.size(); b++) {
    const int r0 = rb[b], r1 = rb[b + 1], k0 = rp[r0], k1 = rp[r1];
    int end_row = r1;
    if (r1 - r0 == 1 && k1 - k0 > kLongRow) {
      end_row = -(1 + (int)runs.size());
      for (int base = k0; base < k1; base += kChunk) {
        const int end = std::min(k1, base + kChunk);
        bool run = true;
        for (int k = base + 1; k < end && run; k++) run = cj[k] == cj[k - 1] + 1;
        runs.push_back(run ? cj[base] : -1);
      }
    }
    d.push_back(r0); d.push_back(end_row); d.push_back(k0); d.push_back(k1);
  }
  return d;
}
}  // namespace

// ------------------------------------------------------------------------------------------------ policy
// include/osqp_hip.h OSQPHipPolicy.  The ONLY place of the library that reads the environment is policy_from_env().
namespace {
thread_local OSQPHipPolicy g_default_policy;
thread_local bool g_default_policy_set = false;

// runtime_only: refresh the fields that may change between two solves of a handle (experiments switch OSQP_HIP_SMALL_DIRECT etc. at run time)
void policy_from_env(OSQPHipPolicy &p, bool runtime_only) {
  if (!runtime_only) { if (g_default_policy_set) p = g_default_policy; else Engine::default_policy(&p); }
  auto on = [](const char *name, OSQPInt &v) { if (const char *e = std::getenv(name)) v = e[0] != '0'; };      // "0" switches off, anything else on
  auto num = [](const char *name, OSQPInt &v) { if (const char *e = std::getenv(name)) v = std::atoi(e); };
  auto real = [](const char *name, OSQPFloat &v) { if (const char *e = std::getenv(name)) v = std::atof(e); };
  on("OSQP_HIP_SMALL_DIRECT", p.small_direct); num("OSQP_HIP_DEVICE_DRIVEN", p.device_driven); on("OSQP_HIP_BATCH_REORDER", p.batch_reorder);
  num("OSQP_HIP_RHO_WINDOW", p.rho_window); real("OSQP_HIP_RHO_WINDOW_TOL", p.rho_window_tol); on("OSQP_HIP_RHO_PERSIST", p.rho_persist);
  if (const char *e = std::getenv("OSQP_HIP_RHO_TOL_EXP")) { const double v = std::atof(e); p.rho_tol_exp = v > 0 && v <= 1 ? v : 0.5; }
  real("OSQP_HIP_BUDGET_TOLERATE", p.budget_tolerate); real("OSQP_HIP_BUDGET_SIGMA", p.budget_sigma); num("OSQP_HIP_BUDGET_SLACK", p.budget_slack);
  if (std::getenv("OSQP_HIP_BUDGET_FULL")) p.budget_full = 1;
  on("OSQP_HIP_CG_ESCALATE", p.cg_escalate); on("OSQP_HIP_STALL", p.stall);
  real("OSQP_HIP_POLISH_DELTA_FLOOR", p.polish_delta_floor); real("OSQP_HIP_POLISH_PCG_TOL", p.polish_pcg_tol);
  on("OSQP_HIP_SLOT_POLL", p.slot_poll); num("OSQP_HIP_POLL_LOW", p.poll_low); real("OSQP_HIP_POLL_FIRST", p.poll_first);
  real("OSQP_HIP_POLL_FRAC", p.poll_frac); real("OSQP_HIP_POLL_WAIT", p.poll_wait);
  num("OSQP_HIP_FINISH_PAIRS", p.finish_pairs); num("OSQP_HIP_POLL_SLEEP_US", p.poll_sleep_us);
  if (std::getenv("OSQP_HIP_SLOT_LOG")) p.slot_log = 1;
  if (std::getenv("OSQP_HIP_BATCH_TIMING")) p.batch_timing = 1;
  if (std::getenv("OSQP_HIP_WB_LOG")) p.woodbury_log = 1;
  if (const char *e = std::getenv("OSQP_HIP_BATCH_VARIANT")) {
    static const char *names[] = {"", "direct", "direct256", "w64", "w256", "generic"};
    p.batch_variant = 0;
    for (int k = 1; k <= 5; k++) if (!std::strcmp(e, names[k])) p.batch_variant = k;
  }
  if (runtime_only) return;
  on("OSQP_HIP_WOODBURY_FUSED", p.woodbury_fused); real("OSQP_HIP_WOODBURY_DIRECT_TOL", p.woodbury_direct_tol);
  on("OSQP_HIP_WOODBURY", p.woodbury); on("OSQP_HIP_WOODBURY_DIRECT", p.woodbury_direct); on("OSQP_HIP_WOODBURY_LARGE", p.woodbury_large);
  num("OSQP_HIP_REORDER", p.reorder);
  on("OSQP_HIP_GRAPH", p.graph); on("OSQP_HIP_SLOTS", p.slots); on("OSQP_HIP_PCG_FUSED", p.pcg_fused); on("OSQP_HIP_F1", p.f1); on("OSQP_HIP_WINDOW", p.window);
  real("OSQP_HIP_EXTRAP", p.extrap);
  if (const char *e = std::getenv("OSQP_HIP_RHO_EQ_FACTOR")) { const double v = std::atof(e); if (v >= 1.0) p.rho_eq_factor = v; }
  if (std::getenv("OSQP_HIP_SETUP_TIMING")) p.setup_timing = 1;
}
}  // namespace

void Engine::default_policy(OSQPHipPolicy *p) {
  if (!p) return;
  *p = OSQPHipPolicy();
  p->graph = p->slots = p->pcg_fused = p->f1 = p->window = p->device_driven = p->small_direct = p->batch_reorder = 1; p->batch_variant = 0;
  p->extrap = 0.9; p->rho_eq_factor = 0.0;
  p->rho_window = 10; p->rho_window_tol = 0.1; p->rho_persist = 1; p->rho_tol_exp = 0.5;
  p->budget_tolerate = 0.0; p->budget_sigma = 3.0; p->budget_slack = 0; p->budget_full = 0; p->cg_escalate = 1; p->stall = 1;
  p->polish_delta_floor = 1e-3; p->polish_pcg_tol = 1e-15; p->woodbury = 1; p->woodbury_direct = 1; p->woodbury_large = 1;
  p->slot_poll = 1; p->poll_low = 6; p->poll_first = 0.8; p->poll_frac = 0.75; p->poll_wait = 0.7;
  p->finish_pairs = 12; p->poll_sleep_us = 30;
  p->reorder = 1; p->woodbury_fused = 1; p->woodbury_direct_tol = 1e-6; p->debug_fail_refactor = 0;
}
void Engine::set_default_policy(const OSQPHipPolicy *p) {
  g_default_policy_set = p != nullptr;
  if (p) g_default_policy = *p;
}
int Engine::get_policy(OSQPHipPolicy *p) const { if (!p) return OSQP_DATA_VALIDATION_ERROR; *p = pol_; return OSQP_NO_ERROR; }
int Engine::set_policy(const OSQPHipPolicy *p) {
  if (!p) return OSQP_DATA_VALIDATION_ERROR;
  if (!(p->extrap >= 0 && p->extrap <= 2) || p->rho_window < 0 || !(p->rho_window_tol > 0) || !(p->rho_tol_exp > 0 && p->rho_tol_exp <= 1) ||
      !(p->budget_sigma >= 0) || p->finish_pairs < 1 || p->batch_variant < 0 || p->batch_variant > 5 ||
      !(p->rho_eq_factor == 0 || p->rho_eq_factor >= 1) || p->reorder < 0 || p->reorder > 2 || !(p->woodbury_direct_tol > 0 && p->woodbury_direct_tol < 1) || !(p->polish_delta_floor > 0) || !(p->polish_pcg_tol > 0 && p->polish_pcg_tol < 1))
    return OSQP_SETTINGS_VALIDATION_ERROR;
  const OSQPHipPolicy old = pol_;
  pol_ = *p; pol_explicit_ = true;
  // [setup] fields keep the value the handle was built with
  pol_.slots = old.slots; pol_.pcg_fused = old.pcg_fused; pol_.f1 = old.f1; pol_.window = old.window; pol_.woodbury = old.woodbury; pol_.woodbury_direct = old.woodbury_direct; pol_.woodbury_large = old.woodbury_large; pol_.reorder = old.reorder; pol_.woodbury_fused = old.woodbury_fused; pol_.woodbury_direct_tol = old.woodbury_direct_tol;
  if (pol_.graph != old.graph) { use_graph_ = pol_.graph != 0; if (dev_ready_) { be::activate(d_); be::sync(d_); drop_graphs(); } }
  if (dev_ready_) d_.theta = pol_.extrap;
  if (dev_ready_ && d_.wb.dbg && pol_.debug_fail_refactor != old.debug_fail_refactor) { be::activate(d_); const int v = pol_.debug_fail_refactor; be::h2d(d_, d_.wb.dbg, &v, sizeof(int)); }
  if (dev_ready_ && pol_.rho_eq_factor >= 1.0 && pol_.rho_eq_factor != old.rho_eq_factor) return set_rho_eq_factor(pol_.rho_eq_factor);
  return OSQP_NO_ERROR;
}

Engine::Engine() {
  pub.settings = &settings; pub.solution = &solution; pub.info = &info; pub.work = reinterpret_cast<OSQPWorkspace *>(this);
  policy_from_env(pol_, false);
  use_graph_ = pol_.graph != 0; use_slots_ = pol_.slots != 0;
  if (pol_.rho_eq_factor >= 1.0) { eq_factor_mixed_ = pol_.rho_eq_factor; eq_factor_env_ = true; }
}
Engine::~Engine() { free_all(); }

void Engine::drop_graphs() {
  for (auto &kv : graphs_) be::graph_free(d_, kv.second);
  graphs_.clear();
  for (auto &kv : sgraphs_) be::graph_free(d_, kv.second);
  sgraphs_.clear();
}

void Engine::sync_graph_scalars() {
  const double sig[6] = {d_.theta, d_.alpha, d_.sigma, d_.rho_eq_factor, d_.rho_eq_mixed, (double)d_.eq_from_cnt};
  if (std::memcmp(sig, graph_sig_, sizeof(sig)) == 0) return;
  if (!graphs_.empty() || !sgraphs_.empty()) { be::sync(d_); drop_graphs(); }
  std::memcpy(graph_sig_, sig, sizeof(sig));
}

void Engine::free_all() {
  if (!dev_ready_) return;
  try { be::activate(d_); be::ext_wait(d_); be::sync(d_); } catch (const DeviceError &) {}       // runs in the destructor: release what we can
  drop_graphs();
  if (bbuf_) { be::dfree(d_, bbuf_); bbuf_ = nullptr; bbuf_cap_ = 0; }
  if (d_batch_order_) { be::dfree(d_, d_batch_order_); d_batch_order_ = nullptr; batch_order_cap_ = 0; }
  if (d_batch_iters_) { be::dfree(d_, d_batch_iters_); d_batch_iters_ = nullptr; d_batch_iters_n_ = 0; }
  batch_order_.clear();
  if (ckpt_) { be::dfree(d_, ckpt_); ckpt_ = nullptr; }
  free_batch_direct();
  if (d_.f1.va) d_.Minv = d_.xs = d_.p = d_.r = d_.s = nullptr;      // (these point into the F1 arena, freed as one block below)
  void *ptrs[] = {d_.A.rowptr, d_.A.col, d_.A.blkdesc, d_.A.val, d_.B.rowptr, d_.B.col, d_.B.blkdesc, d_.B.val, d_.Bdiag, d_.A.runinfo, d_.B.runinfo, d_.A.blkwin, d_.B.blkwin, d_.A.lcol, d_.B.lcol, d_.qraw, d_.lraw, d_.uraw, d_.cnt,
                  d_.q, d_.l, d_.u, d_.D, d_.Dinv, d_.E, d_.Einv, d_.rho, d_.rho_inv, d_.ctype, d_.x, d_.z, d_.y, d_.dx,
                  d_.dy, d_.xs, d_.xg, d_.xsp, d_.ztg, d_.zt, d_.t0, d_.v, d_.r, d_.uu, d_.p, d_.s, d_.w, d_.t, d_.Minv, d_.uu2, d_.ms, d_.part, d_.res,
                  d_.scal, d_.flags, d_.slot, d_.Praw, d_.Araw, d_.cs, d_.Pi, d_.Pj, d_.Pm1, d_.Pm2, d_.Ai, d_.Aj, d_.AmA, d_.AmB,
                  d_.wb.AL.rowptr, d_.wb.AL.col, d_.wb.AL.blkdesc, d_.wb.AL.runinfo, d_.wb.AL.val, d_.wb.ALT.rowptr, d_.wb.ALT.col, d_.wb.ALT.blkdesc, d_.wb.ALT.runinfo, d_.wb.ALT.val,
                  d_.wb.al_src, d_.wb.alt_src, d_.wb.islong, d_.wb.rows, d_.wb.WT, d_.wb.S, d_.wb.Sinv, d_.wb.g, d_.wb.h, d_.wb.Dinv0, d_.wb.colmap, d_.wb.W, d_.wb.pv, d_.wb.info, d_.wb.dbg, d_.wb.x.tile, d_.wb.x.tile2, d_.wb.x.partG, d_.wb.x.partZ, d_.wb.x.ls0, d_.wb.x.ls1, d_.wb.x.sc_ptr, d_.wb.x.sc_row, d_.wb.x.sc_src, d_.wb.x.sc_val, d_.wb.x.bjj,
                  d_.ctl, d_.f1.blk, d_.f1.ent, d_.f1.cptr, d_.f1.prp, d_.f1.pcol, d_.f1.psrc, d_.f1.pval, d_.f1.va, d_pc_, d_pr_};
  for (void *p : ptrs) if (p) be::dfree(d_, p);
  be::destroy(d_);
  d_ = Dev(); d_pc_ = d_pr_ = nullptr;
  dev_ready_ = false;
}

// ------------------------------------------------------------------------------------------------ settings
int Engine::validate_settings(const OSQPSettings *s, bool at_setup) {
  if (!s) return OSQP_SETTINGS_VALIDATION_ERROR;
  bool ok = s->device >= 0 && s->scaling >= 0 && s->rho > 0 && s->sigma > 0 && s->alpha > 0 && s->alpha < 2 &&
            s->cg_max_iter > 0 && s->cg_tol_reduction > 0 && s->cg_tol_fraction > 0 && s->cg_tol_fraction < 1 &&
            s->adaptive_rho_interval >= 0 && s->adaptive_rho_fraction > 0 && s->adaptive_rho_tolerance >= 1 &&
            s->max_iter > 0 && s->eps_abs >= 0 && s->eps_rel >= 0 && (s->eps_abs > 0 || s->eps_rel > 0) &&
            s->eps_prim_inf > 0 && s->eps_dual_inf > 0 && s->check_termination >= 0 && s->time_limit > 0 &&
            s->delta > 0 && s->polish_refine_iter >= 0 &&
            (s->cg_precond == OSQP_NO_PRECONDITIONER || s->cg_precond == OSQP_DIAGONAL_PRECONDITIONER);
  for (int flag : {s->verbose, s->warm_starting, s->polishing, s->rho_is_vec, s->adaptive_rho, s->scaled_termination, s->check_dualgap})
    ok = ok && (flag == 0 || flag == 1);
  if (!ok) return OSQP_SETTINGS_VALIDATION_ERROR;
  if (at_setup && s->linsys_solver != OSQP_INDIRECT_SOLVER) return OSQP_LINSYS_SOLVER_INIT_ERROR;   // GPU engine is PCG only
  return OSQP_NO_ERROR;
}

int Engine::auto_rho_interval() const {
  if (settings.adaptive_rho_interval > 0) return settings.adaptive_rho_interval;
  return settings.check_termination > 0 ? 2 * settings.check_termination : 50;
}

// ------------------------------------------------------------------------------------------------ scaling
// Ruiz equilibration + cost normalisation, _osqp.py:389-497 (host, once per setup).
void Engine::compute_scaling(std::vector<double> &Px, std::vector<double> &Ax, std::vector<double> &qs) {
  D_.assign(n, 1.0); E_.assign(m, 1.0); c_ = 1.0;
  std::vector<double> dt(n), et(m), nP(n);
  auto p_col_norms = [&](std::vector<double> &out) {      // columns of the full symmetric P from its upper triangle
    std::fill(out.begin(), out.end(), 0.0);
    for (int j = 0; j < n; j++)
      for (int k = P_.p[j]; k < P_.p[j + 1]; k++) {
        double a = std::fabs(Px[k]); int i = P_.i[k];
        out[j] = std::max(out[j], a); out[i] = std::max(out[i], a);
      }
  };
  for (int it = 0; it < settings.scaling; it++) {
    p_col_norms(dt);                                                    // _norm_KKT_cols :348-361
    std::fill(et.begin(), et.end(), 0.0);
    for (int j = 0; j < n; j++)
      for (int k = A_.p[j]; k < A_.p[j + 1]; k++) {
        double a = std::fabs(Ax[k]);
        dt[j] = std::max(dt[j], a); et[A_.i[k]] = std::max(et[A_.i[k]], a);
      }
    for (int j = 0; j < n; j++) dt[j] = 1.0 / std::sqrt(limit_scaling(dt[j]));      // :419-421
    for (int i = 0; i < m; i++) et[i] = 1.0 / std::sqrt(limit_scaling(et[i]));
    for (int j = 0; j < n; j++) {                                                    // :432-439
      for (int k = P_.p[j]; k < P_.p[j + 1]; k++) Px[k] *= dt[P_.i[k]] * dt[j];
      for (int k = A_.p[j]; k < A_.p[j + 1]; k++) Ax[k] *= et[A_.i[k]] * dt[j];
      qs[j] *= dt[j]; D_[j] *= dt[j];
    }
    for (int i = 0; i < m; i++) E_[i] *= et[i];
    p_col_norms(nP);                                                                 // :443-468
    double mean = 0; for (int j = 0; j < n; j++) mean += nP[j];
    mean /= std::max(n, 1);
    double nq = 0; for (int j = 0; j < n; j++) nq = std::max(nq, std::fabs(qs[j]));
    double ct = 1.0 / limit_scaling(std::max(limit_scaling(nq), mean));
    for (auto &v : Px) v *= ct;
    for (auto &v : qs) v *= ct;
    c_ *= ct;
  }
  Dinv_.resize(n); Einv_.resize(m);
  for (int j = 0; j < n; j++) Dinv_[j] = 1.0 / D_[j];
  for (int i = 0; i < m; i++) Einv_[i] = 1.0 / E_[i];
  cinv_ = 1.0 / c_;
}

// P <- c D P D, A <- E A D with the stored scaling (_osqp.py:1443, :1463)
void Engine::scale_matrix_values(std::vector<double> &Px, std::vector<double> &Ax) const {
  Px.resize(P_.nnz()); Ax.resize(A_.nnz());
  for (int j = 0; j < n; j++) {
    for (int k = P_.p[j]; k < P_.p[j + 1]; k++) Px[k] = c_ * D_[P_.i[k]] * D_[j] * P_.x[k];
    for (int k = A_.p[j]; k < A_.p[j + 1]; k++) Ax[k] = E_[A_.i[k]] * D_[j] * A_.x[k];
  }
}

// constraint classes, _osqp.py:505-518 (on the SCALED bounds, as the reference does).
//
// Weight of equality rows.  The reference sets rho_i = 1e3 * rho_bar on equality rows (RHO_EQ_OVER_RHO_INEQ,
// _osqp.py:27,521) -- free for a direct KKT solve, but for the reduced-KKT PCG it puts weights 1 and 1000 side by side in
// K = P + sigma I + A' diag(rho) A: the Jacobi-preconditioned condition number becomes ~1e3 (measured: 35 CG iterations
// per decade on the banded QPs), the capped PCG stops converging and ADMM itself slows down (config 2: 1275 ADMM
// iterations with the cap binding in every iteration vs 575 for the direct path).  Measured on the simulator with the
// factor as the only change (DESIGN.md "Equality weight"): 10 instead of 1e3 gives the same ADMM iteration counts on
// every MIXED problem tried and 2-11x fewer PCG iterations; when ALL active rows are equalities the ratio cannot
// affect the conditioning, and the large value is what makes ADMM fast (feasibility_test.py: 475 vs 4250 iterations).
// Hence: 1e3 (the reference's value) if no inequality row is active, eq_factor_mixed_ (10) otherwise.  The ADMM fixed
// point, i.e. the solution, does not depend on it.  osqp_hip_set_rho_eq_factor() / OSQP_HIP_RHO_EQ_FACTOR override it.
void Engine::classify_constraints(const std::vector<double> &ls, const std::vector<double> &us) {
  ctype_.resize(m);
  int n_ineq = 0;
  for (int i = 0; i < m; i++) {
    int t;
    if (ls[i] < -OSQP_INFTY * kMinScaling && us[i] > OSQP_INFTY * kMinScaling) t = -1;
    else if (us[i] - ls[i] < kRhoTol) t = 1;
    else t = 0;
    if (!settings.rho_is_vec) t = 0;
    ctype_[i] = t;
    n_ineq += (t == 0);
  }
  d_.rho_eq_factor = (n_ineq == 0) ? 1e3 : mixed_eq_factor();
  d_.eq_from_cnt = 0;                                   // host classification: k_set_rho takes the factor from rho_eq_factor
}

// Device-side counterpart of upload_q / upload_bounds_and_types: scaling and classification kernels over the resident raw vectors
void Engine::device_scale_vectors(bool q, bool bounds) {
  if (q) be::scale_q(d_, c_);
  if (bounds) {
    d_.rho_eq_mixed = mixed_eq_factor(); d_.eq_from_cnt = 1;
    be::scale_bounds(d_, settings.rho_is_vec);
    scaled_stale_ = true;
  }
}

void Engine::ensure_host_vectors() {
  if (raw_stale_) {                                     // the last update came through device pointers
    be::d2h(d_, q0_.data(), d_.qraw, sizeof(double) * n);
    if (m > 0) { be::d2h(d_, l0_.data(), d_.lraw, sizeof(double) * m); be::d2h(d_, u0_.data(), d_.uraw, sizeof(double) * m); }
    raw_stale_ = false;
  }
  if (scaled_stale_) {
    ls_.resize(m); us_.resize(m);
    for (int i = 0; i < m; i++) { ls_[i] = E_[i] * l0_[i]; us_[i] = E_[i] * u0_[i]; }
    const int keep = d_.eq_from_cnt; const double keepf = d_.rho_eq_factor;
    classify_constraints(ls_, us_);                     // (host copy of ctype only: the device already holds its own)
    d_.eq_from_cnt = keep; if (keep) d_.rho_eq_factor = keepf;
    scaled_stale_ = false;
  }
}

// Equality weight on problems with inequality rows.  The value 10 (see above) is what the PCG needs at scale; a problem with at
// most kSmallEqN variables is solved by CG in at most n steps whatever the conditioning, and on small LPs / rank-deficient QPs the
// reference's 1e3 is what ADMM itself needs (fuzz: 3 of 150 random small problems reach max_iter with 10 and solve in 500-5000
// iterations, like the oracle, with 1e3).  osqp_hip_set_rho_eq_factor() overrides both.
double Engine::mixed_eq_factor() const {
  constexpr int kSmallEqN = 256;
  return (!eq_factor_set_ && !eq_factor_env_ && n <= kSmallEqN) ? 1e3 : eq_factor_mixed_;
}

int Engine::set_rho_eq_factor(double f) {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  if (!(f >= 1.0)) return OSQP_SETTINGS_VALIDATION_ERROR;
  be::activate(d_);
  eq_factor_mixed_ = f; eq_factor_set_ = true;
  ensure_host_vectors();
  upload_bounds_and_types();
  be::set_rho(d_, rho_bar_);
  be::precond(d_, settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER);
  return OSQP_NO_ERROR;
}

void Engine::upload_bounds_and_types() {
  std::vector<double> ls(m), us(m);
  for (int i = 0; i < m; i++) { ls[i] = E_[i] * l0_[i]; us[i] = E_[i] * u0_[i]; }       // _osqp.py:435-436, :1357-1358
  apply_scaled_bounds(ls, us);
}

void Engine::apply_scaled_bounds(const std::vector<double> &ls, const std::vector<double> &us) {
  ls_ = ls; us_ = us;
  classify_constraints(ls, us);
  be::h2d(d_, d_.l, ls.data(), sizeof(double) * m);
  be::h2d(d_, d_.u, us.data(), sizeof(double) * m);
  be::h2d(d_, d_.ctype, ctype_.data(), sizeof(int) * m);
}

void Engine::upload_q() {
  std::vector<double> qs(n);
  for (int j = 0; j < n; j++) qs[j] = c_ * D_[j] * q0_[j];                              // _osqp.py:1328
  be::h2d(d_, d_.q, qs.data(), sizeof(double) * n);
}

void Engine::fill_matrix_values(const std::vector<double> &Px, const std::vector<double> &Ax) {
  std::fill(Bval_.begin(), Bval_.end(), 0.0);
  for (int j = 0; j < n; j++) Bval_[bdiag_[j]] = settings.sigma;
  for (int k = 0; k < P_.nnz(); k++) {
    Bval_[Pmap1_[k]] += Px[k];
    if (Pmap2_[k] >= 0) Bval_[Pmap2_[k]] = Px[k];
  }
  for (int k = 0; k < A_.nnz(); k++) { Aval_[AmapA_[k]] = Ax[k]; Bval_[AmapB_[k]] = Ax[k]; }
  be::h2d(d_, d_.A.val, Aval_.data(), sizeof(double) * Aval_.size());
  be::h2d(d_, d_.B.val, Bval_.data(), sizeof(double) * Bval_.size());
}


// ------------------------------------------------------------------------------------------------ reordering
// A QP whose band structure is hidden by the order in which its variables and constraints happen to be numbered takes the slow
// path (global gathers, two launches per PCG iteration) although a permutation would make it banded.  compute_reorder finds one:
//   1. breadth-first order of the COLUMNS through the bipartite graph of A (column -> its rows -> their columns) joined with P's
//      pattern, started from a pseudo-peripheral column (two sweeps), component by component (Cuthill-McKee levels);
//   2. three barycentre sweeps -- a row sits at the mean rank of its columns, a column moves to the mean position of its rows,
//      ranks are renewed by sorting -- which straighten the arbitrary order inside the BFS levels (measured on config 2 with shuffled
//      rows and columns: window of a 1000-entry row block 287 columns as generated, 331 after the BFS, 286 after two sweeps);
//   3. rows sorted by the middle of their (new) column range.
// O(nnz) per sweep + two sorts of n / m keys; runs only when the natural order does not admit the one-launch form.
void Engine::compute_reorder(const std::vector<int> &Arp, const std::vector<int> &Arj, const std::vector<int> &Brp, const std::vector<int> &Bj) {
  std::vector<int> cstamp(n, 0), rstamp(m, 0), comp_done(n, 0), order, sweep, best;
  order.reserve(n);
  int stamp = 0;
  auto bfs = [&](int start, std::vector<int> &out) {
    out.clear(); stamp++;
    out.push_back(start); cstamp[start] = stamp;
    for (size_t h = 0; h < out.size(); h++) {
      const int j = out[h];
      for (int k = A_.p[j]; k < A_.p[j + 1]; k++) {
        const int i = A_.i[k];
        if (rstamp[i] == stamp) continue;
        rstamp[i] = stamp;
        for (int e = Arp[i]; e < Arp[i + 1]; e++) { const int c = Arj[e]; if (cstamp[c] != stamp) { cstamp[c] = stamp; out.push_back(c); } }
      }
      for (int k = Brp[j]; k < Brp[j + 1] && Bj[k] < n; k++) { const int c = Bj[k]; if (cstamp[c] != stamp) { cstamp[c] = stamp; out.push_back(c); } }
    }
  };
  for (int s0 = 0; s0 < n; s0++) {
    if (comp_done[s0]) continue;
    bfs(s0, sweep);
    if (sweep.size() > 2) { bfs(sweep.back(), best); bfs(best.back(), sweep); }      // pseudo-peripheral start: the far end of the far end
    for (int c : sweep) { comp_done[c] = 1; order.push_back(c); }
  }
  std::vector<double> rank(n), prow(m), pcol(n);
  for (int k = 0; k < n; k++) rank[order[k]] = k;
  std::vector<int> idx(n);
  for (int it = 0; it < 3; it++) {
    for (int i = 0; i < m; i++) {
      const int cnt = Arp[i + 1] - Arp[i];
      double s = 0; for (int e = Arp[i]; e < Arp[i + 1]; e++) s += rank[Arj[e]];
      prow[i] = cnt ? s / cnt : 0.0;
    }
    for (int j = 0; j < n; j++) {
      const int cnt = A_.p[j + 1] - A_.p[j];
      double s = 0; for (int k = A_.p[j]; k < A_.p[j + 1]; k++) s += prow[A_.i[k]];
      pcol[j] = cnt ? s / cnt : rank[j];
    }
    for (int j = 0; j < n; j++) idx[j] = j;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return pcol[a] < pcol[b] || (pcol[a] == pcol[b] && rank[a] < rank[b]); });
    for (int k = 0; k < n; k++) rank[idx[k]] = k;
  }
  pc_.assign(n, 0); ipc_.assign(n, 0);
  for (int j = 0; j < n; j++) { ipc_[j] = (int)rank[j]; pc_[(int)rank[j]] = j; }
  std::vector<long> key(m);
  for (int i = 0; i < m; i++) {
    int lo = INT32_MAX, hi = -1;
    for (int e = Arp[i]; e < Arp[i + 1]; e++) { lo = std::min(lo, ipc_[Arj[e]]); hi = std::max(hi, ipc_[Arj[e]]); }
    key[i] = hi >= 0 ? (long)lo + hi : 2L * n;                       // (empty rows last)
  }
  pr_.resize(m);
  for (int i = 0; i < m; i++) pr_[i] = i;
  std::stable_sort(pr_.begin(), pr_.end(), [&](int a, int b) { return key[a] < key[b]; });
  ipr_.assign(m, 0);
  for (int i = 0; i < m; i++) ipr_[pr_[i]] = i;
}

// P_, A_, q0_, l0_, u0_ <- the permuted problem; PvalMap_ / AvalMap_ = where each of the caller's stored entries went.  Entries keep the
// caller's relative order inside a (row, column) pair (a CSC may repeat an entry), columns come out with ascending row indices.
void Engine::apply_reorder() {
  auto permute_csc = [&](HostCsc &M, const std::vector<int> &rmap, const std::vector<int> &cmap, bool upper, std::vector<int> &vmap) {
    const int nz = M.nnz(), nr = M.nr, nc = M.nc;
    std::vector<int> ri(nz), ci(nz);
    for (int j = 0; j < nc; j++)
      for (int k = M.p[j]; k < M.p[j + 1]; k++) {
        int r = rmap[M.i[k]], c = cmap[j];
        if (upper && r > c) std::swap(r, c);
        ri[k] = r; ci[k] = c;
      }
    // stable counting sort by row, then by column: sorted by (column, row), ties in the caller's order
    std::vector<int> byrow(nz), cnt(std::max(nr, nc) + 1, 0);
    for (int k = 0; k < nz; k++) cnt[ri[k] + 1]++;
    for (int r = 0; r < nr; r++) cnt[r + 1] += cnt[r];
    for (int k = 0; k < nz; k++) byrow[cnt[ri[k]]++] = k;
    HostCsc O; O.nr = nr; O.nc = nc; O.p.assign(nc + 1, 0); O.i.resize(nz); O.x.resize(nz);
    for (int k = 0; k < nz; k++) O.p[ci[k] + 1]++;
    for (int c = 0; c < nc; c++) O.p[c + 1] += O.p[c];
    std::vector<int> cur(O.p.begin(), O.p.end() - 1);
    vmap.assign(nz, 0);
    for (int t = 0; t < nz; t++) { const int k = byrow[t], pos = cur[ci[k]]++; O.i[pos] = ri[k]; O.x[pos] = M.x[k]; vmap[k] = pos; }
    M = std::move(O);
  };
  permute_csc(P_, ipc_, ipc_, true, PvalMap_);
  permute_csc(A_, ipr_, ipc_, false, AvalMap_);
  q0_ = to_internal_n(q0_.data()); l0_ = to_internal_m(l0_.data()); u0_ = to_internal_m(u0_.data());
  reordered_ = true;
}

void Engine::clear_reorder() {
  reordered_ = false;
  pc_.clear(); pr_.clear(); ipc_.clear(); ipr_.clear(); PvalMap_.clear(); AvalMap_.clear();
}

// ------------------------------------------------------------------------------------------------ F1 plan
// One launch per PCG iteration (backend.h DevF1): symbolic data, built once at setup from the row blocks of A.  The form applies when
// every row block of A has a column window of at most kF1Win columns and at most kF1MaxRows rows, the windows of blocks g and g + D
// never overlap for some D <= kF1MaxD (banded / block-banded A -- as given, or after Engine::reorder has found the band), and the
// columns can be dealt out to the blocks as OWN columns -- consecutive ranges [cs[g], cs[g+1]) inside the block's window, at most
// kF1MaxOwn of them with at most kF1PChunk entries of P + sigma I.  Anything else keeps the two-kernel form.  OSQPHipPolicy::f1 = 0
// switches the plan off.  plan_f1 is host-only (no device state is touched: setup may try several row blockings / orderings).
bool Engine::plan_f1(const std::vector<int> &rb, const std::vector<int> &Arp, const std::vector<int> &Arj,
                     const std::vector<int> &Brp, const std::vector<int> &Bj, F1Plan &pl) {
  pl = F1Plan();
  if (!pol_.f1) return false;
  const int nb = (int)rb.size() - 1;
  if (!be::device_assembly() || m == 0 || nb < kGrid / 4) return false;      // (few blocks: most workgroups would idle in the vector update)
  std::vector<int> a0(nb), wl(nb), lo0(nb), hi0(nb);
  for (int b = 0; b < nb; b++) {
    const int r0 = rb[b], r1 = rb[b + 1], k0 = Arp[r0], k1 = Arp[r1];
    if (k1 == k0 || r1 - r0 > kF1MaxRows || k1 - k0 > kF1Chunk || (r1 - r0 == 1 && k1 - k0 > kLongRow)) return false;
    int lo = INT32_MAX, hi = -1;
    for (int k = k0; k < k1; k++) { lo = std::min(lo, Arj[k]); hi = std::max(hi, Arj[k]); }
    if (hi - lo + 1 > kF1Win) return false;
    lo0[b] = lo; hi0[b] = hi;
  }
  // own columns: cs[g] follows the rows (rb[g] n / m: on a band of slope n / m these are the columns under the block) and is clamped into
  // what the neighbouring windows allow -- a column left of block g's window belongs to an earlier block, one right of block g - 1's
  // window to a later one; a column no window holds goes to the block in front of the gap
  std::vector<int> cs(nb + 1);
  cs[0] = 0;
  for (int g = 1; g < nb; g++) {
    const int ideal = (int)((long)rb[g] * n / m);
    const int lo = std::max(cs[g - 1], std::min(lo0[g], n)), up = hi0[g - 1] + 1;
    cs[g] = lo <= up ? std::min(std::max(ideal, lo), up) : lo;
    cs[g] = std::min(std::max(cs[g], cs[g - 1]), n);
  }
  cs[nb] = n;
  for (int b = 0; b < nb; b++) {
    // (the scatter window also covers the block's own columns: (P + sigma I) u of those joins the block's slice of A' t; a column
    //  without entries of the block's rows simply has an empty segment)
    int lo = lo0[b], hi = hi0[b];
    if (cs[b + 1] > cs[b]) { lo = std::min(lo, cs[b]); hi = std::max(hi, cs[b + 1] - 1); }
    if (hi - lo + 1 > kF1Win) return false;
    a0[b] = lo; wl[b] = hi - lo + 1;
  }
  int D = 0;
  for (int t = 1; t <= kF1MaxD && !D; t++) {
    bool ok = true;
    for (int g = 0; g + t < nb && ok; g++) ok = a0[g] + wl[g] <= a0[g + t];
    if (ok) D = t;
  }
  if (!D) return false;
  // the compact CSR of P + sigma I (row j of B up to its first A' entry)
  std::vector<int> &prp = pl.prp; prp.assign(n + 1, 0);
  for (int j = 0; j < n; j++) { int c = 0; for (int k = Brp[j]; k < Brp[j + 1] && Bj[k] < n; k++) c++; prp[j + 1] = prp[j] + c; }
  const int pnnz = prp[n];
  std::vector<int> &pcol = pl.pcol, &psrc = pl.psrc; pcol.assign(std::max(pnnz, 1), 0); psrc.assign(std::max(pnnz, 1), 0);
  for (int j = 0; j < n; j++) for (int k = Brp[j], o = prp[j]; k < Brp[j + 1] && Bj[k] < n; k++, o++) { pcol[o] = Bj[k]; psrc[o] = k; }
  std::vector<int> &blk = pl.blk; blk.assign(16 * (size_t)nb, 0);
  std::vector<unsigned int> &ent = pl.ent; ent.assign(Arj.size(), 0u);
  std::vector<unsigned short> &cptr = pl.cptr; cptr.clear();
  std::vector<int> order, tpos;
  for (int b = 0; b < nb; b++) {
    if (cs[b + 1] - cs[b] > kF1MaxOwn || prp[cs[b + 1]] - prp[cs[b]] > kF1PChunk) return false;
    const int r0 = rb[b], r1 = rb[b + 1], k0 = Arp[r0], k1 = Arp[r1], cnt = k1 - k0;
    int *w = &blk[16 * (size_t)b];
    w[0] = r0; w[1] = r1; w[2] = k0; w[3] = k1;
    w[4] = b < D ? 0 : a0[b]; w[5] = b + D < nb ? a0[b + D] : n; w[6] = cs[b]; w[7] = cs[b + 1];
    w[8] = (int)cptr.size(); w[9] = prp[cs[b]]; w[10] = prp[cs[b + 1]]; w[11] = 0;
    // gather window: the columns of the block's rows of A, together with those of its own rows of P + sigma I when that widens the
    // window by at most a quarter (every window column costs 4 + D vector loads; a P entry outside the window costs as many, once)
    int g0 = a0[b], g1 = a0[b] + wl[b];
    for (int k = prp[cs[b]]; k < prp[cs[b + 1]]; k++) { g0 = std::min(g0, pcol[k]); g1 = std::max(g1, pcol[k] + 1); }
    if (g1 - g0 > kF1Win || 4 * (g1 - g0) > 5 * wl[b]) { g0 = a0[b]; g1 = a0[b] + wl[b]; }
    w[12] = g0; w[13] = g1 - g0; w[14] = a0[b]; w[15] = wl[b];
    // column-major order of the block's entries: stable by local column (rows ascending within a column)
    order.resize(cnt);
    for (int e = 0; e < cnt; e++) order[e] = e;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return Arj[k0 + x] < Arj[k0 + y]; });
    tpos.resize(cnt);
    for (int t = 0; t < cnt; t++) tpos[order[t]] = t;
    for (int r = r0; r < r1; r++)
      for (int k = Arp[r]; k < Arp[r + 1]; k++)
        ent[k] = (unsigned)(Arj[k] - g0) | ((unsigned)(r - r0) << 9) | ((unsigned)tpos[k - k0] << 18);
    const size_t base = cptr.size();
    cptr.resize(base + wl[b] + 1, 0);
    for (int e = 0; e < cnt; e++) cptr[base + (Arj[k0 + e] - a0[b]) + 1]++;
    for (int c = 0; c < wl[b]; c++) cptr[base + c + 1] = (unsigned short)(cptr[base + c + 1] + cptr[base + c]);
  }
  pl.D = D; pl.pnnz = pnnz; pl.ok = true;
  return true;
}

void Engine::upload_f1(const F1Plan &pl) {
  d_.f1 = DevF1();
  if (!pl.ok) return;
  auto up_i = [&](const std::vector<int> &h) { int *p = dev_vec<int>(d_, h.size()); if (!h.empty()) be::h2d(d_, p, h.data(), sizeof(int) * h.size()); return p; };
  DevF1 &f = d_.f1;
  f.D = pl.D; f.pnnz = pl.pnnz;
  f.blk = up_i(pl.blk); f.prp = up_i(pl.prp); f.pcol = up_i(pl.pcol); f.psrc = up_i(pl.psrc);
  f.ent = dev_vec<unsigned int>(d_, pl.ent.size()); be::h2d(d_, f.ent, pl.ent.data(), sizeof(unsigned int) * pl.ent.size());
  f.cptr = dev_vec<unsigned short>(d_, pl.cptr.size()); be::h2d(d_, f.cptr, pl.cptr.data(), sizeof(unsigned short) * pl.cptr.size());
  f.pval = dev_vec<double>(d_, pl.pnnz);
  f.ns = ((size_t)n + 31) / 32 * 32;                       // 256-byte aligned vectors
  f.va = dev_vec<double>(d_, (7 + 2 * (size_t)pl.D) * f.ns);
  f.on = 1;
}

// ------------------------------------------------------------------------------------------------ Woodbury plan
// backend.h DevWb: the rows of A with more than kLongRow entries, when there are between 1 and kWbMaxRows of them, are treated exactly
// in the preconditioner.  Symbolic data: the long rows as their own CSR (r x n), its transpose (n x r), where each entry sits in A.val.
void Engine::prepare_wb(const std::vector<int> &Arp, const std::vector<int> &Arj) {
  d_.wb = DevWb();
  if (!pol_.woodbury || !be::wb_supported() || settings.cg_precond != OSQP_DIAGONAL_PRECONDITIONER || m == 0) return;
  std::vector<int> rows;
  for (int i = 0; i < m; i++) if (Arp[i + 1] - Arp[i] > kLongRow) rows.push_back(i);
  const int r = (int)rows.size();
  if (r < 1) return;
  // many long rows: dense S on the device (backend.h kWbLargeMax) -- when the libraries load, the dense blocks fit comfortably (W, S, S^-1:
  // 8 (r ct + 2 r^2) bytes against a budget of 24 GiB of the 288) and the long rows carry most of A (else Jacobi is not the problem)
  bool large = false;
  std::vector<int> colmap;
  int ct = 0;
  if (r > kWbMaxRows) {
    if (r > kWbLargeMax || !pol_.woodbury_large || !be::wb_large_supported()) return;
    size_t nz_long = 0;
    for (int i : rows) nz_long += (size_t)(Arp[i + 1] - Arp[i]);
    if (2 * nz_long < (size_t)Arp[m]) return;
    colmap.assign(n, -1);
    for (int i : rows) for (int k = Arp[i]; k < Arp[i + 1]; k++) colmap[Arj[k]] = 0;
    for (int j = 0; j < n; j++) if (colmap[j] == 0) colmap[j] = ct++;
    if (8.0 * ((double)r * ct + 2.0 * (double)r * r) > 24.0 * 1024 * 1024 * 1024) return;
    large = true;
  }
  std::vector<unsigned char> islong(m, 0);
  std::vector<int> lrp(r + 1, 0), lcol, lsrc;
  for (int a = 0; a < r; a++) {
    const int i = rows[a]; islong[i] = 1;
    for (int k = Arp[i]; k < Arp[i + 1]; k++) { lcol.push_back(Arj[k]); lsrc.push_back(k); }
    lrp[a + 1] = (int)lcol.size();
  }
  std::vector<int> trp(n + 1, 0), tcol(lcol.size()), tsrc(lcol.size());
  for (int c : lcol) trp[c + 1]++;
  for (int j = 0; j < n; j++) trp[j + 1] += trp[j];
  { std::vector<int> cur(trp.begin(), trp.end() - 1);
    for (int a = 0; a < r; a++) for (int k = lrp[a]; k < lrp[a + 1]; k++) { const int pos = cur[lcol[k]]++; tcol[pos] = a; tsrc[pos] = lsrc[k]; } }
  auto up_i = [&](const std::vector<int> &h) { int *p = dev_vec<int>(d_, h.size()); if (!h.empty()) be::h2d(d_, p, h.data(), sizeof(int) * h.size()); return p; };
  auto up_csr = [&](DevCsr &M, int nr, int nc, const std::vector<int> &rp, const std::vector<int> &cj) {
    std::vector<int> rb;
    if (&M == &d_.wb.AL) { for (int a = 0; a <= nr; a++) rb.push_back(a); }      // every long row is a block of its own
    else rb = build_row_blocks(rp, nr);
    std::vector<int> runs;
    M.nrows = nr; M.ncols = nc; M.nnz = (int)cj.size(); M.nblk = (int)rb.size() - 1; M.split = nc; M.single = 0; M.nwin = 0;
    M.rowptr = up_i(rp); M.col = up_i(cj); M.blkdesc = up_i(block_descs(rb, rp, cj, runs)); M.runinfo = up_i(runs);
    M.val = dev_vec<double>(d_, cj.size());
  };
  DevWb &w = d_.wb;
  w.r = r;
  up_csr(w.AL, r, n, lrp, lcol); up_csr(w.ALT, n, r, trp, tcol);
  w.al_src = up_i(lsrc); w.alt_src = up_i(tsrc); w.rows = up_i(rows);
  w.islong = dev_vec<unsigned char>(d_, m); be::h2d(d_, w.islong, islong.data(), m);
  w.S = dev_vec<double>(d_, (size_t)r * r); w.Sinv = dev_vec<double>(d_, (size_t)r * r);
  w.g = dev_vec<double>(d_, r); w.h = dev_vec<double>(d_, r); w.Dinv0 = dev_vec<double>(d_, n);
  if (large) {
    w.large = 1; w.ct = ct; w.colmap = up_i(colmap);
    w.W = dev_vec<double>(d_, (size_t)r * ct);                  // (zero-filled by the allocator: only the pattern's positions are ever written)
    w.pv = dev_vec<double>(d_, (size_t)n + m + n + 4 + r);
  } else w.WT = dev_vec<double>(d_, (size_t)n * r);
  w.info = dev_vec<int>(d_, 2); w.dbg = dev_vec<int>(d_, 1);
  if (pol_.debug_fail_refactor > 0) { const int v = pol_.debug_fail_refactor; be::h2d(d_, w.dbg, &v, sizeof(int)); }
  w.on = 1;
  // K0 diagonal <=> P has diagonal entries only and every short row of A has exactly one entry: then M = K (backend.h DevWb::exact)
  bool diag = pol_.woodbury_direct != 0;
  for (int j = 0; j < n && diag; j++) for (int k = P_.p[j]; k < P_.p[j + 1]; k++) if (P_.i[k] != j) { diag = false; break; }
  for (int i = 0; i < m && diag; i++) if (!islong[i] && Arp[i + 1] - Arp[i] > 1) diag = false;
  w.exact = diag ? 1 : 0;
  // (large mode: decided numerically after every factorisation -- two-entry rows whose contributions to K0's off-diagonal cancel, as in
  //  the lasso's  -t <= x <= t , are as good as one-entry rows)
  if (large) { w.probe = pol_.woodbury_direct != 0; w.exact = 0; w.log = pol_.woodbury_log; w.exact_tol = pol_.woodbury_direct_tol > 0 ? pol_.woodbury_direct_tol : 1e-6; }
  // The direct mode in two launches per ADMM iteration (backend.h DevWbx): additionally every short row has EXACTLY one entry (an empty
  // row would have no column to be updated with) and the problem is small enough for the per-workgroup partials (n <= kWbxMaxN)
  if (w.exact && !large && pol_.woodbury_fused && be::wbx_supported() && n <= kWbxMaxN) {
    bool ok = true;
    for (int i = 0; i < m && ok; i++) if (!islong[i] && Arp[i + 1] - Arp[i] != 1) ok = false;
    if (ok) {
      std::vector<int> sc_ptr(n + 1, 0), sc_row, sc_src;
      for (int j = 0; j < n; j++) {
        for (int k = A_.p[j]; k < A_.p[j + 1]; k++) if (!islong[A_.i[k]]) { sc_row.push_back(A_.i[k]); sc_src.push_back(AmapA_[k]); }
        sc_ptr[j + 1] = (int)sc_row.size();
      }
      DevWbx &x = w.x;
      x.G = (n + kWbxCols - 1) / kWbxCols; x.nsc = (int)sc_row.size();
      x.tile = dev_vec<double>(d_, (size_t)x.G * kWbMaxRows * kWbxCols);       // (zero-filled by the allocator: only the pattern's positions are ever written)
      x.tile2 = dev_vec<double>(d_, (size_t)x.G * kWbMaxRows * kWbxCols);
      x.partG = dev_vec<double>(d_, (size_t)x.G * kWbMaxRows); x.partZ = dev_vec<double>(d_, (size_t)x.G * kWbMaxRows);
      x.ls0 = dev_vec<double>(d_, 3 * (size_t)kWbMaxRows); x.ls1 = dev_vec<double>(d_, 3 * (size_t)kWbMaxRows);
      x.sc_ptr = up_i(sc_ptr); x.sc_row = up_i(sc_row); x.sc_src = up_i(sc_src); x.sc_val = dev_vec<double>(d_, sc_row.size());
      x.bjj = dev_vec<double>(d_, n);
      be::wbx_init(d_);
      x.on = 1;
    }
  }
}

// ------------------------------------------------------------------------------------------------ setup
int Engine::setup(const OSQPCscMatrix *P, const double *q, const OSQPCscMatrix *A, const double *l, const double *u,
                  int m_, int n_, const OSQPSettings *s) {
  double t0 = now_s();
  const bool ptime = pol_.setup_timing != 0;
  double tl = t0;
  auto lap = [&](const char *what) { if (ptime) { double t = now_s(); std::fprintf(stderr, "[osqp_hip setup] %-28s %8.2f ms\n", what, 1e3 * (t - tl)); tl = t; } };
  // ---- data validation (the C core's validate_data; error numbering bindings.cpp.in:364-375) ----
  if (!P || !A || !q || n_ <= 0 || m_ < 0) return OSQP_DATA_VALIDATION_ERROR;
  if (m_ > 0 && (!l || !u)) return OSQP_DATA_VALIDATION_ERROR;
  if (P->m != n_ || P->n != n_ || A->m != m_ || A->n != n_) return OSQP_DATA_VALIDATION_ERROR;
  auto csc_ok = [](const OSQPCscMatrix *M) {
    if (!M->p) return false;
    if (M->p[0] != 0) return false;
    for (int j = 0; j < M->n; j++) if (M->p[j + 1] < M->p[j]) return false;
    int nz = M->p[M->n];
    if (nz > 0 && (!M->i || !M->x)) return false;
    for (int k = 0; k < nz; k++) if (M->i[k] < 0 || M->i[k] >= M->m) return false;
    return true;
  };
  if (!csc_ok(P) || !csc_ok(A)) return OSQP_DATA_VALIDATION_ERROR;
  for (int j = 0; j < n_; j++)
    for (int k = P->p[j]; k < P->p[j + 1]; k++) if (P->i[k] > j) return OSQP_DATA_VALIDATION_ERROR;   // upper triangular only
  for (int i = 0; i < m_; i++) if (!(l[i] <= u[i])) return OSQP_DATA_VALIDATION_ERROR;
  int err = validate_settings(s, true);
  if (err) return err;

  free_all();
  n = n_; m = m_; settings = *s;
  rho_bar_ = clamp_rho(settings.rho); settings.rho = rho_bar_;                            // _osqp.py:503
  auto copy_csc = [](HostCsc &H, const OSQPCscMatrix *M) {
    H.nr = M->m; H.nc = M->n; int nz = M->p[M->n];
    H.p.assign(M->p, M->p + M->n + 1); H.i.assign(M->i, M->i + nz); H.x.assign(M->x, M->x + nz);
  };
  copy_csc(P_, P); copy_csc(A_, A);
  q0_.assign(q, q + n); l0_.assign(l, l + m); u0_.assign(u, u + m);
  lap("validate + copy");

  // ---- device ----
  err = be::init(d_, settings.device);
  if (err) return err;
  lap("device init");
  dev_ready_ = true;
  // ---- scaling: on the device once the matrices are assembled there; the test-only host simulator
  //      keeps the driver's host restatement of _osqp.py:389-497 ----
  const bool dev_asm = be::device_assembly();
  std::vector<double> Px, Ax, qs;
  d_.n = n; d_.m = m; d_.sigma = settings.sigma; d_.alpha = settings.alpha;

  const int nzA = A_.nnz(), nzP = P_.nnz();
  std::vector<int> Arp, Arj, Brp, Bj;
  int nzB = 0;
  // (a lambda: setup may build the structure twice -- as given, and for the reordered problem)
  auto build_structure = [&]() {
    // A as CSR (the incoming CSC is CSR(A')) + map CSC index -> CSR position
    Arp.assign(m + 1, 0); Arj.assign(nzA, 0);
    AmapA_.resize(nzA);
    for (int k = 0; k < nzA; k++) Arp[A_.i[k] + 1]++;
    for (int i = 0; i < m; i++) Arp[i + 1] += Arp[i];
    {
      std::vector<int> cur(Arp.begin(), Arp.end() - 1);
      for (int j = 0; j < n; j++)
        for (int k = A_.p[j]; k < A_.p[j + 1]; k++) { int pos = cur[A_.i[k]]++; Arj[pos] = j; AmapA_[k] = pos; }
    }
    // B = [P + sigma I | A'] as CSR with n rows; row j = (lower part of row j of P) (diag) (upper part) (column j of A)
    Brp.assign(n + 1, 0);
    std::vector<char> hasdiag(n, 0);
    for (int j = 0; j < n; j++) {
      Brp[j + 1] += 1 + (A_.p[j + 1] - A_.p[j]);
      for (int k = P_.p[j]; k < P_.p[j + 1]; k++) {
        int i = P_.i[k];
        if (i == j) hasdiag[j] = 1; else { Brp[j + 1]++; Brp[i + 1]++; }
      }
    }
    for (int j = 0; j < n; j++) Brp[j + 1] += Brp[j];
    nzB = Brp[n];
    Bj.assign(nzB, 0); std::vector<int> &bdiag = bdiag_; bdiag.assign(n, 0);
    Pmap1_.assign(nzP, -1); Pmap2_.assign(nzP, -1); AmapB_.resize(nzA);
    {
      std::vector<int> cur(Brp.begin(), Brp.end() - 1);
      for (int j = 0; j < n; j++) {
        for (int k = P_.p[j]; k < P_.p[j + 1]; k++) {
          int i = P_.i[k];
          if (i == j) continue;
          int p1 = cur[j]++; Bj[p1] = i; Pmap1_[k] = p1;      // (j, i): lower part of row j
        }
        bdiag[j] = cur[j]++; Bj[bdiag[j]] = j;
        for (int k = P_.p[j]; k < P_.p[j + 1]; k++)           // every stored (j, j) entry -- valid CSC may repeat it -- adds into the one slot
          if (P_.i[k] == j) Pmap1_[k] = bdiag[j];
        for (int k = P_.p[j]; k < P_.p[j + 1]; k++) {
          int i = P_.i[k];
          if (i == j) continue;
          int p2 = cur[i]++; Bj[p2] = j; Pmap2_[k] = p2;      // (i, j): upper part of row i (its diagonal is already placed)
        }
      }
      for (int j = 0; j < n; j++)
        for (int k = A_.p[j]; k < A_.p[j + 1]; k++) { int pos = cur[j]++; Bj[pos] = n + A_.i[k]; AmapB_[k] = pos; }
    }
  };
  build_structure();
  lap("CSR(A), B structure, maps");
  std::vector<int> rbA = build_row_blocks(Arp, m), rbB = build_row_blocks(Brp, n);
  lap("row blocks");
  // One launch per PCG iteration (F1 form): wants row blocks of A of at most kF1Chunk entries -- on large problems (n = 1M: the default
  // blocks hold ~2000 entries) A is re-blocked for it, a workgroup then loops over several blocks per launch; when the plan does not
  // apply the default blocks stay
  bool has_long = false;
  for (int i = 0; i < m && !has_long; i++) has_long = Arp[i + 1] - Arp[i] > kLongRow;
  const bool want_f1 = pol_.pcg_fused && use_slots_ && pol_.window != 0 && pol_.f1 && !has_long;
  F1Plan plan;
  auto try_plan = [&]() {
    if (!want_f1) return false;
    if (plan_f1(rbA, Arp, Arj, Brp, Bj, plan)) return true;
    if ((long)nzA > (long)kGrid * kF1Chunk) {
      std::vector<int> rbF = build_row_blocks(Arp, m, kF1Chunk);
      if (plan_f1(rbF, Arp, Arj, Brp, Bj, plan)) { rbA.swap(rbF); return true; }
    }
    return false;
  };
  bool f1ok = try_plan();
  // Reordering (OSQPHipPolicy::reorder; Engine::compute_reorder): 1 = when the one-launch form does not apply to the problem as given,
  // look for a permutation under which it does and keep it only then; 2 = always work on the permuted problem (tests of the plumbing)
  clear_reorder();
  const int reorder = no_reorder_ ? 0 : pol_.reorder;
  if (m > 0 && (reorder == 2 || (reorder == 1 && want_f1 && !f1ok && be::device_assembly() && (int)rbA.size() - 1 >= kGrid / 4))) {
    const double tr = now_s();
    compute_reorder(Arp, Arj, Brp, Bj);
    HostCsc P0 = P_, A0 = A_; std::vector<double> q00 = q0_, l00 = l0_, u00 = u0_;
    apply_reorder();
    build_structure();
    rbA = build_row_blocks(Arp, m); rbB = build_row_blocks(Brp, n);
    f1ok = try_plan();
    if (!f1ok && reorder != 2) {                     // no gain: the problem stays as the caller numbered it
      P_ = std::move(P0); A_ = std::move(A0); q0_ = std::move(q00); l0_ = std::move(l00); u0_ = std::move(u00);
      clear_reorder();
      build_structure();
      rbA = build_row_blocks(Arp, m); rbB = build_row_blocks(Brp, n);
      f1ok = try_plan();
    }
    reorder_ms_ = 1e3 * (now_s() - tr);
    lap("reordering");
  }
  if (!dev_asm) { Px = P_.x; Ax = A_.x; qs = q0_; compute_scaling(Px, Ax, qs); lap("Ruiz scaling (host)"); }
  Arp_ = Arp; Arj_ = Arj; Brp_ = Brp; Bj_ = Bj;
  d_.fused = pol_.pcg_fused ? 1 : 0;                 // 0 selects the 3-kernel sequence
  prepare_wb(Arp, Arj);
  if (d_.wb.on) d_.fused = 0;                        // (the Woodbury-corrected preconditioner lives in the three-kernel PCG form)
  d_.f1 = DevF1();
  if (d_.fused && f1ok) upload_f1(plan);
  if (reordered_) {
    d_pc_ = dev_vec<int>(d_, n); d_pr_ = dev_vec<int>(d_, m);
    be::h2d(d_, d_pc_, pc_.data(), sizeof(int) * n); be::h2d(d_, d_pr_, pr_.data(), sizeof(int) * m);
  }
  lap("F1 / Woodbury plans");
  // block descriptors; long rows also get their run table (see DevCsr::runinfo).  Slices are the fixed kChunk steps the kernels
  // take from the row's first entry (cutting them at run starts instead adds short slices that cost more than the saved
  // index bytes: lasso PCG pair 208 us vs 220 us)
  auto descs = [](const std::vector<int> &rb, const std::vector<int> &rp, const std::vector<int> &cj, std::vector<int> &runs) {
    std::vector<int> d; d.reserve(4 * rb.size());
    runs.clear();
    for (size_t b = 0; b + 1 < rb.size(); b++) {
      const int r0 = rb[b], r1 = rb[b + 1], k0 = rp[r0], k1 = rp[r1];
      int end_row = r1;
      if (r1 - r0 == 1 && k1 - k0 > kLongRow) {
        end_row = -(1 + (int)runs.size());
        for (int base = k0; base < k1; base += kChunk) {
          const int end = std::min(k1, base + kChunk);
          bool run = true;
          for (int k = base + 1; k < end && run; k++) run = cj[k] == cj[k - 1] + 1;
          runs.push_back(run ? cj[base] : -1);
        }
      }
      d.push_back(r0); d.push_back(end_row); d.push_back(k0); d.push_back(k1);
    }
    return d;
  };

  // column windows of the (short-row) blocks, see DevCsr::blkwin.  OSQPHipPolicy::window = 0 turns the windowed path off (A/B runs).
  const bool win_on = pol_.window != 0;
  auto windows = [win_on](const std::vector<int> &rb, const std::vector<int> &rp, const std::vector<int> &cj, int split,
                    std::vector<int> &win, std::vector<unsigned short> &lcol) {
    const size_t nb = rb.size() - 1;
    win.assign(4 * nb, 0); lcol.assign(std::max<size_t>(cj.size(), 1), 0);
    int nwin = 0;
    for (size_t b = 0; b < nb; b++) {
      const int r0 = rb[b], r1 = rb[b + 1], k0 = rp[r0], k1 = rp[r1];
      win[4 * b + 1] = -1;
      if (!win_on || (r1 - r0 == 1 && k1 - k0 > kLongRow) || k1 == k0) continue;
      int lo0 = INT32_MAX, hi0 = -1, lo1 = INT32_MAX, hi1 = -1;
      for (int k = k0; k < k1; k++) {
        const int c = cj[k];
        if (c < split) { lo0 = std::min(lo0, c); hi0 = std::max(hi0, c); } else { lo1 = std::min(lo1, c - split); hi1 = std::max(hi1, c - split); }
      }
      const long len0 = hi0 >= 0 ? (long)hi0 - lo0 + 1 : 0, len1 = hi1 >= 0 ? (long)hi1 - lo1 + 1 : 0;
      if (len0 + len1 > kWinCap) continue;
      if (len0 == 0) lo0 = 0;
      if (len1 == 0) lo1 = 0;
      win[4 * b] = lo0; win[4 * b + 1] = (int)len0; win[4 * b + 2] = lo1; win[4 * b + 3] = (int)len1;
      for (int k = k0; k < k1; k++) {
        const int c = cj[k];
        lcol[k] = (unsigned short)(c < split ? c - lo0 : len0 + (c - split - lo1));
      }
      nwin++;
    }
    return nwin;
  };
  auto up_i = [&](const std::vector<int> &h) { int *p = dev_vec<int>(d_, h.size()); be::h2d(d_, p, h.data(), sizeof(int) * h.size()); return p; };
  auto up_win = [&](DevCsr &M, const std::vector<int> &rb, const std::vector<int> &rp, const std::vector<int> &cj, int split) {
    std::vector<int> win; std::vector<unsigned short> lcol;
    M.split = split;
    M.single = (int)rb.size() - 1 <= kGrid;
    for (size_t b = 0; b + 1 < rb.size(); b++) if (rb[b + 1] - rb[b] > kBlock) M.single = 0;
    M.nwin = windows(rb, rp, cj, split, win, lcol);
    M.blkwin = up_i(win);
    M.lcol = dev_vec<unsigned short>(d_, lcol.size());
    be::h2d(d_, M.lcol, lcol.data(), sizeof(unsigned short) * lcol.size());
  };
  d_.A.nrows = m; d_.A.ncols = n; d_.A.nnz = nzA; d_.A.nblk = (int)rbA.size() - 1;
  d_.A.rowptr = up_i(Arp); d_.A.col = up_i(Arj); { std::vector<int> runs; d_.A.blkdesc = up_i(descs(rbA, Arp, Arj, runs)); d_.A.runinfo = up_i(runs); } d_.A.val = dev_vec<double>(d_, nzA);
  d_.B.nrows = n; d_.B.ncols = n + m; d_.B.nnz = nzB; d_.B.nblk = (int)rbB.size() - 1;
  d_.B.rowptr = up_i(Brp); d_.B.col = up_i(Bj); { std::vector<int> runs; d_.B.blkdesc = up_i(descs(rbB, Brp, Bj, runs)); d_.B.runinfo = up_i(runs); } d_.B.val = dev_vec<double>(d_, nzB);
  d_.Bdiag = up_i(bdiag_);
  up_win(d_.A, rbA, Arp, Arj, n); up_win(d_.B, rbB, Brp, Bj, n);
  lap("upload structure");
  auto dv = [&](size_t cnt) { return dev_vec<double>(d_, cnt); };
  d_.q = dv(n); d_.l = dv(m); d_.u = dv(m); d_.D = dv(n); d_.Dinv = dv(n); d_.E = dv(m); d_.Einv = dv(m);
  d_.rho = dv(m); d_.rho_inv = dv(m); d_.ctype = dev_vec<int>(d_, m);
  d_.x = dv(n); d_.z = dv(m); d_.y = dv(m); d_.dx = dv(n); d_.dy = dv(m); d_.zt = dv(m); d_.t0 = dv(m); d_.v = dv(m);
  d_.xg = dv(n); d_.xsp = dv(n); d_.ztg = dv(m);
  d_.theta = pol_.extrap;                            // PCG start extrapolation (backend.h Dev::xg)
  d_.uu = dv(n); d_.w = dv(n); d_.t = dv(m); d_.uu2 = dv(n); d_.ms = dv(2 * (size_t)n);
  if (d_.f1.on) { const size_t ns = d_.f1.ns; double *va = d_.f1.va; d_.Minv = va; d_.xs = va + ns; d_.p = va + 2 * ns; d_.r = va + 3 * ns; d_.s = va + 5 * ns; }   // backend.h DevF1::va
  else { d_.r = dv(n); d_.p = dv(n); d_.s = dv(n); d_.Minv = dv(n); d_.xs = dv(n); }
  d_.part = dv((size_t)kPartSlots * kGrid); d_.res = dv(R_COUNT); d_.scal = dv(S_HIST + 3 * (kMaxCg + 1)); d_.flags = dev_vec<int>(d_, F_COUNT); d_.slot = dev_vec<int>(d_, be::kSlotInts);
  d_.ctl = be::device_assembly() ? static_cast<Ctl *>(be::alloc(d_, sizeof(Ctl))) : nullptr;      // (the host simulator processes every boundary on the host)
  if (dev_asm) {
    // the caller's values go up once, in their own (CSC) order; every later (re)assembly and the equilibration run on the device
    std::vector<int> Pj(nzP), Aj(nzA);
    for (int j = 0; j < n; j++) { for (int k = P_.p[j]; k < P_.p[j + 1]; k++) Pj[k] = j; for (int k = A_.p[j]; k < A_.p[j + 1]; k++) Aj[k] = j; }
    d_.nzP = nzP; d_.nzA = nzA;
    d_.Praw = dv(nzP); d_.Araw = dv(nzA); d_.cs = dv(2);
    d_.Pi = up_i(P_.i); d_.Pj = up_i(Pj); d_.Pm1 = up_i(Pmap1_); d_.Pm2 = up_i(Pmap2_);
    d_.Ai = up_i(A_.i); d_.Aj = up_i(Aj); d_.AmA = up_i(AmapA_); d_.AmB = up_i(AmapB_);
    be::h2d(d_, d_.Praw, P_.x.data(), sizeof(double) * nzP); be::h2d(d_, d_.Araw, A_.x.data(), sizeof(double) * nzA);
    be::h2d(d_, d_.q, q0_.data(), sizeof(double) * n);
    be::assemble(d_, 0, 1.0, 0);                                     // unscaled, sigma added after the equilibration
    c_ = be::ruiz(d_, settings.scaling);                             // _osqp.py:389-497
    cinv_ = 1.0 / c_;
    be::f1_refresh(d_); be::wb_refresh(d_); be::wbx_refresh(d_);
    D_.resize(n); E_.resize(m); Dinv_.resize(n); Einv_.resize(m);
    be::d2h(d_, D_.data(), d_.D, sizeof(double) * n); be::d2h(d_, Dinv_.data(), d_.Dinv, sizeof(double) * n);
    if (m > 0) { be::d2h(d_, E_.data(), d_.E, sizeof(double) * m); be::d2h(d_, Einv_.data(), d_.Einv, sizeof(double) * m); }
    lap("assembly + Ruiz scaling (device)");
  } else {
    Aval_.assign(nzA, 0.0); Bval_.assign(nzB, 0.0);
    fill_matrix_values(Px, Ax);
    be::h2d(d_, d_.D, D_.data(), sizeof(double) * n); be::h2d(d_, d_.Dinv, Dinv_.data(), sizeof(double) * n);
    be::h2d(d_, d_.E, E_.data(), sizeof(double) * m); be::h2d(d_, d_.Einv, Einv_.data(), sizeof(double) * m);
    lap("matrix values (host-scaled)");
  }
  d_.qraw = dv(n); d_.lraw = dv(m); d_.uraw = dv(m); d_.cnt = dev_vec<int>(d_, 2);
  raw_stale_ = scaled_stale_ = false;
  if (be::device_vec_updates()) {
    be::copy_in(d_, d_.qraw, q0_.data(), sizeof(double) * n, 0);
    be::copy_in(d_, d_.lraw, l0_.data(), sizeof(double) * m, 0); be::copy_in(d_, d_.uraw, u0_.data(), sizeof(double) * m, 0);
    device_scale_vectors(true, true);
  } else {
    upload_q();
    upload_bounds_and_types();
  }
  be::set_rho(d_, rho_bar_);                                       // _osqp.py:499-524
  try { be::precond(d_, settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER); }
  catch (const DeviceError &err) {
    // the first device-side factorisation of the large-rank correction failed (a dense-library call, not this engine's kernels): the
    // handle falls back to plain Jacobi -- said loudly, and visible in OSQPHipStats::woodbury_rows = 0
    if (!(d_.wb.on && d_.wb.large)) throw;
    std::fprintf(stderr, "osqp_hip: large-rank Woodbury correction switched off for this handle (%s)\n", err.what());
    d_.wb.on = 0;
    be::precond(d_, settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER);
  }
  be::init_iterates(d_, 1);

  sol_x_.assign(n, kNaN); sol_y_.assign(m, kNaN); sol_pc_.assign(m, kNaN); sol_dc_.assign(n, kNaN);
  solution.x = sol_x_.data(); solution.y = sol_y_.data(); solution.prim_inf_cert = sol_pc_.data(); solution.dual_inf_cert = sol_dc_.data();
  std::memset(&info, 0, sizeof(info));
  set_status(OSQP_UNSOLVED);
  cg_budget_ = 0; have_tol_ = false; first_run_ = true; slot_pred_[0] = slot_pred_[1] = 6.0; slot_pred_[2] = 14.0;
  stats_ = OSQPHipStats(); stats_.nnzA = nzA; stats_.nnzB = nzB;
  be::sync(d_);
  lap("vectors, rho, preconditioner");
  info.setup_time = now_s() - t0;
  if (settings.verbose) {
    std::printf("-----------------------------------------------------------------\n");
    std::printf("  OSQP ADMM engine for AMD MI355X (%s), indirect (PCG) solver\n", be::name());
    std::printf("-----------------------------------------------------------------\n");
    std::printf("problem:  variables n = %d, constraints m = %d\n          nnz(P) + nnz(A) = %d\n", n, m, nzP + nzA);
    std::printf("settings: eps_abs = %.1e, eps_rel = %.1e, rho = %.2e%s, sigma = %.2e, alpha = %.2f,\n          max_iter = %d, scaling = %d, check_termination = %d, cg_max_iter = %d\n\n",
                settings.eps_abs, settings.eps_rel, settings.rho, settings.adaptive_rho ? " (adaptive)" : "", settings.sigma,
                settings.alpha, settings.max_iter, settings.scaling, settings.check_termination, settings.cg_max_iter);
  }
  return OSQP_NO_ERROR;
}

// ------------------------------------------------------------------------------------------------ driver
void Engine::set_status(int st) {
  info.status_val = st;
  const char *s = "unsolved";
  switch (st) {
    case OSQP_SOLVED: s = "solved"; break;
    case OSQP_SOLVED_INACCURATE: s = "solved inaccurate"; break;
    case OSQP_PRIMAL_INFEASIBLE: s = "primal infeasible"; break;
    case OSQP_PRIMAL_INFEASIBLE_INACCURATE: s = "primal infeasible inaccurate"; break;
    case OSQP_DUAL_INFEASIBLE: s = "dual infeasible"; break;
    case OSQP_DUAL_INFEASIBLE_INACCURATE: s = "dual infeasible inaccurate"; break;
    case OSQP_MAX_ITER_REACHED: s = "maximum iterations reached"; break;
    case OSQP_TIME_LIMIT_REACHED: s = "run time limit reached"; break;
    case OSQP_NON_CVX: s = "problem non convex"; break;
    case OSQP_SIGINT: s = "interrupted"; break;
    default: break;
  }
  std::snprintf(info.status, sizeof(info.status), "%s", s);
}

// One chunk = `niter` ADMM iterations, each  KB, budget x (K1,K2,Kv), KA  -- enqueued eagerly or replayed from a
// hipGraph captured once per (niter, budget).
void Engine::run_chunk(int niter, int budget) {
  const bool fused = be::pcg_fused(d_), wb = d_.wb.on != 0;
  const bool xy = wb && d_.wb.exact && d_.wb.x.on;       // the direct mode in two launches per ADMM iteration (wbdirect_hip.hip)
  auto enqueue = [&](int count) {
    if (xy) { be::wbx_chunk(d_, count); return; }
    for (int it = 0; it < count; it++) {
      be::kb_rhs(d_);
      if (wb) be::wb_apply(d_, 0, d_.wb.exact);
      if (wb && d_.wb.exact) { be::ka(d_, budget); continue; }      // M^-1 r_0 is the solve (x~ formed by the last kernel of M^-1)
      for (int i = 0; i < budget; i++) { be::k1(d_, i); be::k2(d_, i); if (!fused || i == budget - 1) { be::kv(d_, i); if (wb) be::wb_apply(d_, (i + 1) & 1); } }
      be::ka(d_, budget);
    }
  };
  stats_.kernel_launches += xy ? 2.0 * niter + 1 : (double)niter * ((wb && d_.wb.exact) ? 5 : (fused ? 3 + 2 * budget : 2 + 3 * budget + (wb ? 3 * (budget + 1) : 0)));
  if (!(use_graph_ && be::graphs_supported())) { enqueue(niter); return; }
  // one executable graph per (ADMM iterations, PCG budget); graphs are kept below kMaxGraphNodes kernel nodes (a
  // check_termination = 0 solve would otherwise capture max_iter * (2 + 3*budget) nodes in one graph)
  constexpr int kMaxGraphNodes = 8192;
  const int per = std::max(1, kMaxGraphNodes / (2 + 3 * budget + (wb ? 3 * (budget + 1) : 0)));
  for (int left = niter; left > 0;) {
    const int cnt = std::min(left, per);
    auto key = std::make_pair(cnt, budget | ((wb && d_.wb.exact) ? (1 << 24) : 0) | (xy ? (1 << 25) : 0));      // (the direct mode is another launch sequence: it may come and go with rho in the large-rank form)
    auto it = graphs_.find(key);
    if (it == graphs_.end()) {
      be::graph_begin(d_);
      enqueue(cnt);
      it = graphs_.emplace(key, be::graph_end(d_)).first;
    }
    be::graph_launch(d_, it->second);
    stats_.graph_launches += 1;
    left -= cnt;
  }
}

// Slot form of a chunk (backend_hip.hip "slot kernels"): begin_target > 0 starts a chunk of that many ADMM iterations (eager one-thread
// launch: target and PCG cap travel in the phase record), then `pairs` (B slot, A slot) launches follow as replays of captured
// strings of 256 / 192 / 128 / 96 / ... / 3 / 2 / 1 pairs -- the same sixteen graphs serve every chunk, whatever its length; begin_target == 0 tops up
// a chunk that has not finished.
void Engine::run_slots(int begin_target, int pairs, int cap) {
  stats_.kernel_launches += 2.0 * pairs + (begin_target > 0 ? 1 : 0);
  if (begin_target > 0) be::slot_begin(d_, begin_target, cap);
  if (!(use_graph_ && be::graphs_supported())) { for (int k = 0; k < pairs; k++) be::slot_pair(d_); return; }
  for (int left = pairs; left > 0;) {
    int unit = 1;
    // (captured string lengths: a denser set than powers of two -- every replay boundary is a bubble of a few microseconds, 63 pairs are
    //  48 + 12 + 3, not 32 + 16 + 8 + 4 + 2 + 1)
    for (int u : {256, 192, 128, 96, 64, 48, 32, 24, 16, 12, 8, 6, 4, 3, 2}) if (left >= u) { unit = u; break; }
    const std::array<int, 3> key = {unit, 0, 0};
    auto it = sgraphs_.find(key);
    if (it == sgraphs_.end()) {
      be::graph_begin(d_);
      for (int k = 0; k < unit; k++) be::slot_pair(d_);
      it = sgraphs_.emplace(key, be::graph_end(d_)).first;
    }
    be::graph_launch(d_, it->second);
    stats_.graph_launches += 1;
    left -= unit;
  }
}

double Engine::rho_estimate(const double *res) const {                                   // _osqp.py:880-908 (scaled quantities)
  double pri = res[R_PRI_S] / (std::max(res[R_AX_S], res[R_Z_S]) + 1e-10);
  double dua = res[R_DUA_S] / (std::max(std::max(res[R_ATY_S], res[R_PX_S]), res[R_QN_S]) + 1e-10);
  return clamp_rho(rho_bar_ * std::sqrt(pri / (dua + 1e-10)));
}

// _osqp.py:998-1077.  Returns 1 when a terminal status was set.
int Engine::check_termination(const double *res, bool approximate) {
  double ea = settings.eps_abs, er = settings.eps_rel, epi = settings.eps_prim_inf, edi = settings.eps_dual_inf;
  if (approximate) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
  const bool unsc = settings.scaling && !settings.scaled_termination;
  if (info.prim_res > OSQP_INFTY || info.dual_res > OSQP_INFTY || std::isnan(info.prim_res) || std::isnan(info.dual_res)) {
    set_status(OSQP_NON_CVX); info.obj_val = kNaN; return 1;                            // :1025-1028
  }
  bool pri_ok = false, dua_ok = false, prim_inf = false, dual_inf = false;
  if (m == 0) pri_ok = true;
  else {
    double eps_pri = ea + er * (unsc ? std::max(res[R_AX_U], res[R_Z_U]) : std::max(res[R_AX_S], res[R_Z_S]));   // :728-751
    if (info.prim_res < eps_pri) pri_ok = true;
    else {                                                                              // is_primal_infeasible :796-820
      double nd = unsc ? res[R_DY_U] : res[R_DY_S];
      if (nd > epi && res[R_PINF_LHS] < -epi * nd) {
        be::infeas_primal(d_);
        double r2[R_COUNT]; be::fetch_res(d_, r2);
        prim_inf = (unsc ? r2[R_ATDY_U] : r2[R_ATDY_S]) < epi * nd;
      }
    }
  }
  double mx = unsc ? cinv_ * std::max(std::max(res[R_ATY_U], res[R_PX_U]), res[R_QN_U])
                   : std::max(std::max(res[R_ATY_S], res[R_PX_S]), res[R_QN_S]);             // :766-794
  if (info.dual_res < ea + er * mx) dua_ok = true;
  else {                                                                                // is_dual_infeasible :822-878
    double nd = unsc ? res[R_DX_U] : res[R_DX_S], sc = unsc ? c_ : 1.0;
    if (nd > edi && res[R_QDX] < -sc * edi * nd) {
      be::infeas_dual(d_, edi * nd, unsc ? 1 : 0);
      double r2[R_COUNT]; be::fetch_res(d_, r2);
      if ((unsc ? r2[R_PDX_U] : r2[R_PDX_S]) < sc * edi * nd && r2[R_ADX_VIOL] == 0.0) dual_inf = true;
    }
  }
  // check_dualgap (bindings.cpp.in:442): additionally |duality gap| < eps_abs + eps_rel max(|obj|, |dual obj|)   [UPSTREAM-UNVERIFIED form]
  const bool gap_ok = !settings.check_dualgap ||
                      std::fabs(info.duality_gap) < ea + er * std::max(std::fabs(info.obj_val), std::fabs(info.dual_obj_val));
  if (pri_ok && dua_ok && gap_ok) { set_status(approximate ? OSQP_SOLVED_INACCURATE : OSQP_SOLVED); return 1; }
  if (prim_inf) { set_status(approximate ? OSQP_PRIMAL_INFEASIBLE_INACCURATE : OSQP_PRIMAL_INFEASIBLE); info.obj_val = OSQP_INFTY; return 1; }
  if (dual_inf) { set_status(approximate ? OSQP_DUAL_INFEASIBLE_INACCURATE : OSQP_DUAL_INFEASIBLE); info.obj_val = -OSQP_INFTY; return 1; }
  return 0;
}

// The v1 info fields beyond purepy's (bindings.cpp.in:475, 478, 491-492; defined by the un-vendored C core, so the formulas
// are this engine's reading of their names [UPSTREAM-UNVERIFIED]):
//   dual_obj_val   -1/2 x'Px - sup_{l <= z <= u} y'z   (the support function of the box at y; finite where y respects infinite bounds)
//   duality_gap    obj_val - dual_obj_val
//   rel_kkt_error  max( prim_res / max(||Ax||, ||z||),  dual_res / max(||Px||, ||A'y||, ||q||),  |gap| / max(|obj|, |dual obj|) )
//   primdual_int   integral over the solve time of |duality_gap| (accumulated at the termination checks)
// t0 < 0: no time integration (polish).
void Engine::update_gap_info(const double *res, double t0) {
  const bool unsc = settings.scaling && !settings.scaled_termination;
  const double ci = settings.scaling ? cinv_ : 1.0;
  info.dual_obj_val = (-0.5 * res[R_XPX] - res[R_SUPP]) * ci;
  info.duality_gap = info.obj_val - info.dual_obj_val;
  const double pn = unsc ? std::max(res[R_AX_U], res[R_Z_U]) : std::max(res[R_AX_S], res[R_Z_S]);
  const double dn = unsc ? cinv_ * std::max(std::max(res[R_ATY_U], res[R_PX_U]), res[R_QN_U]) : std::max(std::max(res[R_ATY_S], res[R_PX_S]), res[R_QN_S]);
  const double gn = std::max(std::fabs(info.obj_val), std::fabs(info.dual_obj_val));
  const double tiny = 1e-10;
  info.rel_kkt_error = std::max(std::max(m == 0 ? 0.0 : info.prim_res / (pn + tiny), info.dual_res / (dn + tiny)), std::fabs(info.duality_gap) / (gn + tiny));
  if (t0 >= 0) {
    const double t = now_s() - t0;
    info.primdual_int += std::fabs(info.duality_gap) * std::max(0.0, t - gap_time_);
    gap_time_ = t;
  }
}

int Engine::solve() {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  be::activate(d_);
  const double t0 = now_s();
  if (!pol_explicit_) policy_from_env(pol_, true);
  if (clear_update_time_) { info.update_time = 0; }
  info.update_time += update_time_acc_; update_time_acc_ = 0;
  if (!settings.warm_starting) cold_start();                                            // _osqp.py:1204-1205
  info.rho_updates = 0; info.status_polish = 0; info.polish_time = 0; info.primdual_int = 0; gap_time_ = 0;
  set_status(OSQP_UNSOLVED);
  if (small_direct_applicable()) {
    const int err = solve_small_direct(t0);
    if (err != OSQP_FUNC_NOT_IMPLEMENTED) return err;
  }
  stats_.pcg_iters_total = stats_.pcg_iters_max = stats_.pcg_unconverged = 0;
  stats_.kernel_launches = stats_.graph_launches = 0; stats_.cg_cap_escalations = 0; stats_.slot_topups = 0;
  stats_.woodbury_factorisations = 0; stats_.woodbury_factor_ms = 0;
  double res[R_COUNT];
  admm_core(t0, res);
  info.rho_estimate = rho_estimate(res);                                                 // :1275
  info.solve_time = now_s() - t0;
  if (settings.polishing && info.status_val == OSQP_SOLVED) polish();                   // :1278-1279
  store_solution();
  be::sync(d_);
  info.run_time = (first_run_ ? info.setup_time : info.update_time) + info.solve_time + info.polish_time;   // :1284-1289
  first_run_ = false; clear_update_time_ = true;
  if (settings.verbose)
    std::printf("\nstatus:               %s\n%snumber of iterations: %d\noptimal objective:    %.4f\nrun time:             %.2es\noptimal rho estimate: %.2e\n\n",
                info.status, info.status_polish == 1 ? "solution polish:      successful\n" : (info.status_polish == -1 ? "solution polish:      unsuccessful\n" : ""),
                info.iter, info.obj_val, info.run_time, info.rho_estimate);
  return OSQP_NO_ERROR;
}

// The ADMM loop proper (_osqp.py:1208-1266) on the current device iterates with the current settings; sets info.{iter,
// obj_val, prim_res, dual_res, status*}; leaves the residual block of the last check in res.
//
// The loop is a sequence of CHUNKS of ADMM iterations (up to the next termination check / rho adaptation point / start of the tight
// PCG window in front of one); what happens at a chunk boundary -- termination test, adaptive rho, PCG tolerance and budget -- is
// policy.h, one text for host and device.  Two ways of running it:
//   host-synchronous  (exec_chunk_sync + ctl_boundary on the host): the launch-per-iteration form, the host simulator, verbose
//                     solves, and the FIRST chunk of every solve (it is checkpointed and may be repeated with a larger PCG cap);
//   device-driven     (run_device_driven): the host only feeds strings of slot launches and boundary groups; the device applies
//                     policy.h itself (k_decide) and the host reads the state block when it says "done" or "need host" (second
//                     stage of an infeasibility test, approximate tolerances at max_iter).
void Engine::ctl_setup() {
  Ctl &c = ctl_;
  c = Ctl();
  const int ari = settings.adaptive_rho ? auto_rho_interval() : 0;
  c.ct = settings.check_termination; c.ari = ari; c.max_iter = settings.max_iter;
  c.tightW = (ari > 1 && pol_.rho_window > 0) ? std::min(pol_.rho_window, ari - 1) : 0;
  c.tightF = pol_.rho_window_tol; c.persist = pol_.rho_persist; c.tol_exp = pol_.rho_tol_exp;
  c.m = m; c.scaling = settings.scaling; c.scaled_termination = settings.scaled_termination; c.check_dualgap = settings.check_dualgap;
  c.has_quad = 0;
  for (double v : P_.x) if (v != 0.0) { c.has_quad = 1; break; }      // LPs adapt rho by the setting's literal tolerance (policy.h ctl_rho_rule)
  c.esc_on = pol_.cg_escalate; c.stall_on = pol_.stall; c.full_budget = pol_.budget_full; c.cap_max = kMaxCg;
  c.cg_tol_fraction = settings.cg_tol_fraction; c.cg_tol_reduction = settings.cg_tol_reduction; c.rho_tolerance = settings.adaptive_rho_tolerance;
  c.eps_abs = settings.eps_abs; c.eps_rel = settings.eps_rel; c.eps_pinf = settings.eps_prim_inf; c.eps_dinf = settings.eps_dual_inf;
  c.c = c_; c.cinv = cinv_;
  c.budget_tolerate = pol_.budget_tolerate; c.budget_sigma = pol_.budget_sigma; c.budget_slack = pol_.budget_slack; c.budget_min = d_.wb.on ? 1 : 2;
  c.iter = 0; c.cap = std::min(settings.cg_max_iter, kMaxCg);
  // start with the full budget: a starved PCG in the first chunks costs far more ADMM iterations than the launches it saves
  c.budget[0] = c.budget[1] = c.cap;
  c.stall = 1.0; c.best_dua = INFINITY; c.prev_aobj = INFINITY; c.rho_bar = rho_bar_;
  c.status = CTL_RUNNING;
}

void Engine::apply_rho(double rho) {
  rho_bar_ = rho; settings.rho = rho;
  be::set_rho(d_, rho_bar_);
  const double tf = d_.wb.on ? now_s() : 0.0;
  struct Tally { Engine *e; double t0; ~Tally() { if (t0 > 0) { e->stats_.woodbury_factorisations += 1; e->stats_.woodbury_factor_ms += 1e3 * (now_s() - t0); } } } tally{this, tf};
  try { be::precond(d_, settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER); }
  catch (const DeviceError &err) {
    // a re-factorisation of the Woodbury correction failed in the middle of a solve (dense-library call, or S not positive definite at
    // this rho): the handle continues with plain Jacobi, as setup does -- said loudly, visible in OSQPHipStats::woodbury_rows = 0
    if (!d_.wb.on) throw;
    std::fprintf(stderr, "osqp_hip: Woodbury correction switched off for this handle at rho = %.3e (%s)\n", rho, err.what());
    d_.wb.on = 0; d_.wb.exact = 0;
    be::sync(d_); drop_graphs();
    be::precond(d_, settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER);
  }
}

// info fields of the last check from the state block (+ the time integral of |gap|, accumulated where the host sees a check)
void Engine::info_from_ctl(double t0) {
  const Ctl &c = ctl_;
  info.iter = c.iter; info.obj_val = c.obj_val; info.prim_res = c.prim_res; info.dual_res = c.dual_res;
  info.dual_obj_val = c.dual_obj_val; info.duality_gap = c.duality_gap; info.rel_kkt_error = c.rel_kkt_error;
  info.rho_updates = c.rho_updates;
  if (c.rho_estimate > 0) info.rho_estimate = c.rho_estimate;
  if (t0 >= 0) {
    const double t = now_s() - t0;
    info.primdual_int += std::fabs(info.duality_gap) * std::max(0.0, t - gap_time_);
    gap_time_ = t;
  }
}

// One chunk of `cnt` ADMM iterations with at most `lim` PCG iterations per solve, host-synchronous; afterwards `flags` holds the chunk's
// PCG statistics and, if with_res, `res` the residual block of its last iterate.  Slot form (backend_hip.hip "slot kernels"): the
// chunk is a string of slot launches sized from the PCG iterations the previous chunk of this kind needed; the host watches the
// chunk's progress on a side stream and tops the string up before it runs dry.
void Engine::exec_chunk_sync(int cnt, int lim, bool with_res, int kind, double *res, int *flags) {
  const bool slots = use_slots_ && be::slots_supported(d_);
  if (!slots) {
    run_chunk(cnt, lim);
    if (with_res) { be::residuals(d_); be::fetch_res_flags(d_, res, flags); } else be::fetch_flags(d_, flags);
    return;
  }
  double *pred = slot_pred_;
  int tot[F_COUNT] = {0}, f[F_COUNT];
  int launched_pairs = 0;
  // slot pairs (two launches each) that `its` ADMM iterations with `pcg` PCG iterations each need (backend.h slot_launches)
  auto pairs_for = [&](double its, double pcg) { return 0.5 * its * be::slot_launches(d_, pcg); };
  const double t_chunk = now_s();
  // every launch of an unfinished chunk advances it, so `cnt` iterations under the cap `lim` never need more than this many pairs:
  // the bound on what the host enqueues (a record that stops advancing must not make it enqueue for ever), not a count of top-ups
  // -- one long PCG (the polish: up to kMaxCg iterations at a relative tolerance of 1e-15) legitimately takes many of them
  const int max_pairs = (int)std::ceil(pairs_for(cnt, lim)) + 8;
  if (pol_.slot_poll) {
    const int kLow = pol_.poll_low; const double kFirst = pol_.poll_first, kFrac = pol_.poll_frac, kWait = pol_.poll_wait;
    const double p0 = std::min<double>(pred[kind], lim);
    { const int np = (int)std::ceil(pairs_for(cnt, 0)) + std::max(2, (int)std::floor(kFirst * (pairs_for(cnt, p0) - pairs_for(cnt, 0)))); run_slots(cnt, np, lim); launched_pairs += np; }
    double pair_s = 9e-6, t_prev = now_s();                        // duration of a slot pair, re-estimated from the progress between two polls
    int seq_prev = 0;
    for (int seq = 0, done = 0;;) {
      be::slot_poll(d_, &seq, &done);
      if (done >= cnt) break;
      const double t_now = now_s();
      // (bounded: a record that stops advancing must not make the host enqueue launches for ever -- the synchronising fetch below
      //  then reports what the device did)
      if (t_now - t_chunk > settings.time_limit || launched_pairs > max_pairs) break;
      if (seq - seq_prev >= 8) { pair_s = std::max(5e-6, 2.0 * (t_now - t_prev) / (seq - seq_prev)); t_prev = t_now; seq_prev = seq; }
      const int ahead = launched_pairs - seq / 2;                  // pairs enqueued and not yet executed
      if (ahead > kLow) {
        // (every poll is a small copy that has to squeeze in between the chunk's kernels: poll when the queue can have run low at the
        // earliest, not continuously -- and sleep, not spin: a solving handle must not pin a host core)
        std::this_thread::sleep_for(std::chrono::duration<double>(std::min(2e-3, kWait * (ahead - kLow) * pair_s)));
        continue;
      }
      const int rem = cnt - done;
      const double rate = done > 0 ? std::min<double>(pairs_for(1, lim), (0.5 * seq) / done) : pairs_for(1, p0);      // pairs per ADMM iteration so far
      const int need = (int)std::ceil(rem * rate) + 1 - ahead;
      // (an iteration that has outrun the prediction by far -- none finished yet, more than twice the chunk's predicted need consumed:
      //  grow geometrically)
      const bool outrun = done == 0 && launched_pairs > 2 * (int)std::ceil(pairs_for(cnt, p0)) + 8;
      const int np = std::max(std::max(2, need > 12 ? (int)std::ceil(kFrac * need) : need), outrun ? launched_pairs / 4 : 0);
      run_slots(0, np, lim); launched_pairs += np;
      stats_.slot_topups += 1;
    }
  } else {
    const double pm = std::min<double>(pred[kind], lim);
    const int np = (int)std::ceil(pairs_for(cnt, 0) + 1.05 * (pairs_for(cnt, pm) - pairs_for(cnt, 0))) + 2; run_slots(cnt, np, lim); launched_pairs += np;
  }
  for (;;) {
    if (with_res) { be::residuals(d_); be::fetch_res_flags(d_, res, f); } else be::fetch_flags(d_, f);
    tot[F_STAT_SUM] += f[F_STAT_SUM]; tot[F_STAT_SUMSQ] += f[F_STAT_SUMSQ]; tot[F_STAT_N] += f[F_STAT_N]; tot[F_STAT_UNCONV] += f[F_STAT_UNCONV]; tot[F_STAT_STAG] += f[F_STAT_STAG];
    tot[F_STAT_MAX] = std::max(tot[F_STAT_MAX], f[F_STAT_MAX]);
    const int done = be::slot_done(d_);
    if (be::slot_seq(d_) != 2 * launched_pairs) {        // the record hand-over between the slot launches is broken: nothing computed since is trustworthy
      char msg[160];
      std::snprintf(msg, sizeof(msg), "osqp_hip: slot record hand-over broken: %d slots launched, record counts %d", 2 * launched_pairs, be::slot_seq(d_));
      throw DeviceError(msg);
    }
    if (done >= cnt) break;
    if (launched_pairs > max_pairs) throw DeviceError("osqp_hip: a chunk of ADMM iterations does not finish");
    const int rem = cnt - done;
    const double seen = tot[F_STAT_N] > 0 ? (double)tot[F_STAT_SUM] / tot[F_STAT_N] : pred[kind];
    const double pm = std::min<double>(std::max(seen, pred[kind]), lim);
    const int np = std::max((int)std::ceil(pairs_for(rem, 0) + 1.25 * (pairs_for(rem, pm) - pairs_for(rem, 0))) + 8, launched_pairs / 2);
    run_slots(0, np, lim); launched_pairs += np;
    stats_.slot_topups += 1;
  }
  for (int k = 0; k < F_COUNT; k++) flags[k] = tot[k];
  if (pol_.slot_log)
    std::fprintf(stderr, "chunk %p it %d cnt %d kind %d lim %d pred %.2f used-mean %.2f topups %d unconv %d rho %.4e\n", (void *)this, ctl_.iter, cnt, kind, lim, pred[kind],
                 tot[F_STAT_N] > 0 ? (double)tot[F_STAT_SUM] / tot[F_STAT_N] : 0.0, (int)stats_.slot_topups, tot[F_STAT_UNCONV], rho_bar_);
  if (tot[F_STAT_N] > 0) pred[kind] = (double)tot[F_STAT_SUM] / tot[F_STAT_N];
}

// Device-driven chunks from the state block's current chunk on: returns CTL_DONE, CTL_NEED_HOST, or -2 when the time limit passed.
// The stream carries   [slots] [slots] .. [boundary group] [slots] ..   -- a group acts only when the chunk in flight has finished, the
// slots after a finished chunk idle until a group has set the next one up, and everything idles once the state block says the solve
// is over: what is computed never depends on how the host sizes or times the strings.  The host watches the progress on a side
// stream (be::ctl_poll: no wait on the solve's stream) and keeps the queue a few slot pairs deep: most of what the chunk in flight
// still needs at the rate observed so far; when little is left, the rest plus a boundary group plus the first part of the NEXT
// chunk (by then the device has set it up itself).  Between polls the host sleeps.
// A count of slot pairs that is ONE captured string (run_slots' unit sizes), at least `want` and at most `limit` -- or `want` itself when no
// unit fits between the two.  Every replay boundary is a bubble (1-2 us; ~9 us under a profiler): where the chunk in flight will consume
// the launches anyway, the next larger single string beats an exact count made of three.
static int one_string(int want, int limit) {
  static const int units[] = {2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256};
  for (int u : units) if (u >= want) return u <= limit ? u : want;
  return want;
}

int Engine::run_device_driven(double t0, double *res, int *flags) {
  Ctl &c = ctl_;
  c.status = CTL_RUNNING; c.chunk_done = 0; c.rho_flag = 0; c.stage2 = 0;
  be::ctl_upload(d_, c);
  be::ctl_begin(d_);
  stats_.kernel_launches += 1;
  const int diagonal = settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER;
  auto pairs_for = [&](double its, double pcg) { return 0.5 * its * be::slot_launches(d_, pcg); };
  auto pred_for = [&](const Ctl &s, int kind, int tight) {
    const double pm = s.kind_n[kind] > 0 ? s.kind_sum[kind] / s.kind_n[kind] : slot_pred_[kind];
    return std::min<double>(pm, s.budget[tight]);
  };
  Ctl snap = c;
  long launched = 0;                                 // slot PAIRS enqueued
  int seq = 0, done = 0;                             // slot launches executed / ADMM iterations completed in the chunk in flight (polled)
  // The queue is kept deep enough for a SLOW host as well: the low-water mark follows the measured duration of a poll + top-up cycle
  // (three cycles' worth of slot pairs, never below poll_low) -- under a profiler, or on a loaded host, every runtime call costs a multiple
  // of its usual time, and a 6-pair queue would run dry between two top-ups.
  int kLow = pol_.poll_low, kFinish = pol_.finish_pairs;
  double cycle_s = 0.0, pair_s = 22e-6, t_cycle = now_s(), t_rate = t_cycle; long seq_rate = 0;
  bool timed_out = false;
  auto cycle_done = [&]() {                          // one poll + top-up cycle of the host has ended
    const double t_now = now_s(), c = t_now - t_cycle;
    cycle_s = cycle_s > 0 ? 0.7 * cycle_s + 0.3 * c : c; t_cycle = t_now;
    kLow = std::min(96, std::max<int>(pol_.poll_low, (int)std::ceil(3.0 * cycle_s / pair_s)));
    kFinish = std::max<int>(pol_.finish_pairs, 2 * kLow);
  };
  // chunk in flight as of the last poll, and the progress counters at the first poll that saw it (rate estimate)
  { const int cnt = snap.ch_next - snap.iter;
    const double full = pairs_for(cnt, pred_for(snap, snap.ch_kind, snap.ch_tight));
    const int np = one_string(std::max(2, (int)std::floor(pol_.poll_first * full)), (int)std::floor(0.95 * full));
    run_slots(0, np, 0); launched += np; }
  for (;;) {
    be::ctl_poll(d_, &snap, &seq, &done);
    if (snap.status != CTL_RUNNING) break;
    if (now_s() - t0 > settings.time_limit) { timed_out = true; break; }
    const long ahead = launched - seq / 2;           // pairs enqueued and not yet executed
    { const double t_now = now_s();
      if (seq - seq_rate >= 16) { pair_s = std::max(5e-6, 2.0 * (t_now - t_rate) / (double)(seq - seq_rate)); t_rate = t_now; seq_rate = seq; } }
    if (ahead > kLow) { std::this_thread::sleep_for(std::chrono::microseconds(pol_.poll_sleep_us)); t_cycle = now_s(); continue; }
    const int cnt = snap.ch_next - snap.iter, rem = std::max(0, cnt - done);
    const double pm = pred_for(snap, snap.ch_kind, snap.ch_tight);
    // pairs per ADMM iteration: what THIS chunk has consumed so far per finished iteration (a premature boundary group costs two orders of
    // magnitude more than a late one -- everything queued behind it idles until the next group -- hence the two pairs of slack)
    const double rate = done >= 2 ? std::min<double>(pairs_for(1, snap.budget[snap.ch_tight]), 0.5 * (double)std::max(0, seq - snap.seq_begin) / done) : 1.1 * pairs_for(1, pm);
    const int need = (int)std::ceil(rem * rate) + 2 - (int)ahead;
    if (need > kFinish) {                            // far from the chunk's end: most of what is missing
      const int np = one_string(std::max(2, (int)std::ceil(pol_.poll_frac * need)), need - 2);      // (never beyond what the chunk still needs)
      run_slots(0, np, 0); launched += np;
      stats_.slot_topups += 1;
      cycle_done();
      continue;
    }
    // the chunk's end is within reach: the rest, the boundary group, and the first part of the next chunk
    const int np = std::max(1, need);
    run_slots(0, np, 0); launched += np;
    run_group(diagonal);
    {
      Ctl nx = snap; nx.iter = snap.ch_next; ctl_next_chunk(nx);
      const int ncnt = nx.ch_next - nx.iter;
      if (ncnt > 0) {
        const double full = pairs_for(ncnt, pred_for(snap, nx.ch_kind, nx.ch_tight));
        const int nq = one_string(std::max(2, (int)std::floor(pol_.poll_first * full)), (int)std::floor(0.95 * full));
        run_slots(0, nq, 0); launched += nq;
      }
    }
    cycle_done();
    if (pol_.slot_log) std::fprintf(stderr, "group: device iter %d chunk %d..%d kind %d done %d rem %d rate %.2f ahead %ld pairs %d (launched %ld, seq %d) budget %d/%d tol %.3e rho %.4e; host cycle %.0f us, pair %.1f us, low-water %d\n",
                                    snap.iter, snap.iter, snap.ch_next, snap.ch_kind, done, rem, rate, ahead, np, launched, seq, snap.budget[0], snap.budget[1], snap.tol_abs, snap.rho_bar, 1e6 * cycle_s, 1e6 * pair_s, kLow);
  }
  be::sync(d_);                                      // (what is still queued idles: the state block says the solve is over -- or, after
  be::ctl_download(d_, &c);                          //  a time-out, runs to the end of the strings)
  for (int q = 0; q < R_COUNT; q++) res[q] = c.res[q];
  for (int q = 0; q < F_COUNT; q++) flags[q] = c.last_flags[q];
  for (int k = 0; k < 3; k++) if (c.kind_n[k] > 0) slot_pred_[k] = c.kind_sum[k] / c.kind_n[k];
  if (c.rho_bar != rho_bar_) { rho_bar_ = c.rho_bar; settings.rho = rho_bar_; }      // (applied on the device)
  if (timed_out && c.status == CTL_RUNNING) return -2;
  return c.status;
}

void Engine::run_group(int diagonal) {
  stats_.kernel_launches += 11 + (diagonal ? 1 : 0);
  if (!(use_graph_ && be::graphs_supported())) { be::ctl_group(d_, diagonal); return; }
  const std::array<int, 3> key = {0, 1, diagonal};
  auto it = sgraphs_.find(key);
  if (it == sgraphs_.end()) {
    be::graph_begin(d_);
    be::ctl_group(d_, diagonal);
    it = sgraphs_.emplace(key, be::graph_end(d_)).first;
  }
  be::graph_launch(d_, it->second);
  stats_.graph_launches += 1;
}

void Engine::admm_core(double t0, double *res) {
  Ctl &c = ctl_;
  ctl_setup();
  sync_graph_scalars();
  be::zero(d_, d_.flags + F_STAT_SUM, sizeof(int) * (F_COUNT - F_STAT_SUM));      // (a device-driven solve leaves its last chunk's statistics behind)
  {
    double r0[R_COUNT];
    be::residuals(d_); be::fetch_res(d_, r0);
    ctl_init_tol(c, r0);                               // tolerance and budget restart with every solve: a solve is a deterministic function of (data, iterates)
  }
  have_tol_ = false;
  ctl_next_chunk(c);
  if (settings.verbose) std::printf("iter   objective    prim res   dual res   rho        cg   time\n");
  int flags[F_COUNT] = {0};
  for (;;) {
    int st;
    // (evaluated at every boundary: the Woodbury direct mode may leave the slot form in the middle of a solve -- NEED_REFACTOR below)
    // (the Woodbury direct mode takes the slot form only when asked to, device_driven = 2: its two launches per ADMM iteration are so short
    //  that a boundary group of sixteen launches costs what the host round trip it replaces costs, and the fixed captured strings of the
    //  host-synchronous loop carry no idle launches -- 34 against 41 ms on the portfolio QP)
    d_.wb.x.slots = (pol_.device_driven >= 2 && !settings.verbose) ? 1 : 0;
    const bool slots = use_slots_ && be::slots_supported(d_);
    const bool device_driven = slots && pol_.device_driven && !settings.verbose && be::ctl_supported(d_);
    const bool first_chunk = c.iter == 0;
    if (first_chunk || !device_driven) {
      be::set_pcg_tol(d_, c.tol_rel, ctl_chunk_tol_abs(c));
      // The FIRST chunk is checkpointed: if the PCG starves at the cap in most of its solves, the chunk is repeated from the same
      // iterates with a four times larger cap (ADMM steps taken with stagnating inner solves derail exactly the problems --
      // unbounded / rank-deficient ones -- whose status the first checks decide: tools/fuzz_gpu.py, tests/test_gpu_fuzz.py).
      const bool ckpt_chunk = first_chunk && c.cap < kMaxCg && c.esc_on;
      if (ckpt_chunk) {
        if (!ckpt_) ckpt_ = dev_vec<double>(d_, 2 * (size_t)n + 2 * (size_t)m);
        be::copy_in(d_, ckpt_, d_.x, sizeof(double) * n, 1); be::copy_in(d_, ckpt_ + n, d_.xs, sizeof(double) * n, 1);
        be::copy_in(d_, ckpt_ + 2 * (size_t)n, d_.z, sizeof(double) * m, 1); be::copy_in(d_, ckpt_ + 2 * (size_t)n + m, d_.y, sizeof(double) * m, 1);
      }
      const int lim = c.budget[c.ch_tight], tight = c.ch_tight;
      const double rho_was = rho_bar_;
      exec_chunk_sync(c.ch_next - c.iter, lim, c.ch_at_check && !ckpt_chunk, c.ch_kind, res, flags);
      cg_budget_ = lim;
      if (ckpt_chunk) {
        if (lim >= c.cap && flags[F_STAT_STAG] * 2 > std::max(1, flags[F_STAT_N])) {
          be::copy_in(d_, d_.x, ckpt_, sizeof(double) * n, 1); be::copy_in(d_, d_.xs, ckpt_ + n, sizeof(double) * n, 1);
          be::copy_in(d_, d_.z, ckpt_ + 2 * (size_t)n, sizeof(double) * m, 1); be::copy_in(d_, d_.y, ckpt_ + 2 * (size_t)n + m, sizeof(double) * m, 1);
          be::zero(d_, d_.dx, sizeof(double) * n); be::zero(d_, d_.dy, sizeof(double) * m);
          be::init_iterates(d_, 0);
          c.cap = std::min(kMaxCg, 4 * c.cap); c.budget[0] = c.budget[1] = c.cap;
          c.escalations += 1;
          stats_.kernel_launches = 0; stats_.graph_launches = 0;
          continue;                                       // same chunk again (iter is still 0)
        }
        if (c.ch_at_check) { be::residuals(d_); be::fetch_res(d_, res); }      // (the chunk's PCG statistics are already in flags)
      }
      const bool was_check = c.ch_at_check;
      st = ctl_boundary(c, res, flags);
      if (st == CTL_RUNNING && c.stage2) {               // second stage of the infeasibility tests (two more SpMVs), then the rest of the boundary
        if (c.stage2 & NEED_PINF) be::infeas_primal(d_);
        if (c.stage2 & NEED_DINF) be::infeas_dual(d_, c.inf_thr_d, c.inf_unscaled);
        double r2[R_COUNT]; be::fetch_res(d_, r2);
        for (int q = R_ATDY_U; q <= R_ADX_VIOL; q++) res[q] = r2[q];
        st = ctl_boundary_stage2(c, res, flags);
      }
      if (was_check) {
        info_from_ctl(t0);
        if (settings.verbose)
          std::printf("%4d  %11.4e   %8.2e   %8.2e   %8.2e  %3d  %8.2es   (cg mean %.1f budget %d unconv %d; rho est %.2e)\n", c.iter, info.obj_val, info.prim_res,
                      info.dual_res, rho_was, flags[F_STAT_MAX], now_s() - t0, flags[F_STAT_SUM] / (double)std::max(1, flags[F_STAT_N]), lim,
                      flags[F_STAT_UNCONV], pol_rho_estimate(rho_was, res));
      }
      (void)tight;
      if (st == CTL_RUNNING && c.rho_flag) apply_rho(c.rho_bar);
    } else {
      st = run_device_driven(t0, res, flags);
      info_from_ctl(t0);
      if (st == -2) { set_status(OSQP_TIME_LIMIT_REACHED); break; }
    }
    if (st == CTL_DONE) {
      const int os = c.osqp_status;
      set_status(os);
      if (os == OSQP_NON_CVX) info.obj_val = kNaN;
      else if (os == OSQP_PRIMAL_INFEASIBLE || os == OSQP_PRIMAL_INFEASIBLE_INACCURATE) info.obj_val = OSQP_INFTY;
      else if (os == OSQP_DUAL_INFEASIBLE || os == OSQP_DUAL_INFEASIBLE_INACCURATE) info.obj_val = -OSQP_INFTY;
      break;
    }
    if (st == CTL_NEED_HOST && (c.need & NEED_REFACTOR)) {
      // Woodbury direct mode, device-driven: at the rho update of the last boundary the device-side inversion of S missed the accuracy the
      // direct mode needs (backend k_wb_invert; the chunk it had begun was cancelled).  The solve continues on the host-synchronous path
      // with the corrected preconditioner inside the PCG: rho is applied again (this time through the host's bookkeeping).
      c.need &= ~NEED_REFACTOR; c.status = CTL_RUNNING;
      d_.wb.exact = 0;
      be::sync(d_); drop_graphs();
      apply_rho(c.rho_bar);
      c.rho_flag = 0;
      continue;
    }
    if (st == CTL_NEED_HOST) {                            // max_iter without convergence: the approximate-tolerance pass (:1264-1266)
      if (!check_termination(res, true)) set_status(OSQP_MAX_ITER_REACHED);
      break;
    }
    if (now_s() - t0 > settings.time_limit) { set_status(OSQP_TIME_LIMIT_REACHED); break; }
  }
  info.rho_updates = c.rho_updates;
  stats_.pcg_iters_total = c.pcg_total; stats_.pcg_iters_max = c.pcg_max; stats_.pcg_unconverged = c.pcg_unconv;
  stats_.cg_cap_escalations = c.escalations;
}

// Solution polish (_osqp.py:1710-1828) on the multi-kernel (PCG) path.  The reference guesses the active constraints from (z, y)
// (:1719-1720), solves the reduced KKT system of the equality-constrained QP on that active set
//     [ P   Aa' ] [x ]   [ -q ]
//     [ Aa  0   ] [ya] = [ ba ]            regularised by  diag(+delta I, -delta I)   (:1740-1754)
// with a direct factorisation, repairs the regularisation's error by `polish_refine_iter` steps of iterative refinement
//     s <- s + (K + dK)^-1 (rhs - K s)                                                   (:1692-1708)
// and keeps the result if it improves the residuals (:1786-1793).  Eliminating ya from one refinement step gives
//     (P + delta I + Aa' Aa / delta) x+ = -q + delta x - Aa' ya + Aa' ba / delta ,    ya+ = ya + (Aa x+ - ba) / delta
// -- the proximal method of multipliers with parameter delta, and at the same time ONE ADMM iteration of this engine (alpha = 1) on
// the problem whose active rows are equalities at their bound with weight rho_i = 1 / delta and whose other rows are free: KB forms
// exactly that right-hand side from (x, z = ba, y = ya), the PCG solves the system, KA's y-update is the multiplier step.  So the
// polish IS the reference's recurrence, run by the engine's own kernels with the inner systems solved to a relative residual of 1e-15
// (the right-hand side carries the 1 / delta_eff weights: 1e-12 there leaves 1e-8 in the dual residual).  What differs:
//   * 1 / delta = 1e6 (the default) is out of reach of a Jacobi-preconditioned PCG (condition number ~ ||Aa||^2 / (delta lambda_min)):
//     the recurrence runs with delta_eff = max(delta, 1e-3).  The fixed point -- the solution of the unregularised reduced KKT system
//     -- does not depend on delta; only the contraction per step does (~ delta / mu instead of 1e-6 / mu), so
//   * `polish_refine_iter` is the MINIMUM number of refinement steps: the recurrence continues (at most kPolishMaxSteps) until the
//     reduced system's residuals stop improving, which is where the reference's few steps at delta = 1e-6 end up as well
//     (tests/test_gpu_polish.py compares with the oracle's polish, pinned to the reference, to 1e-8);
//   * the proximal term uses sigma (already on B's diagonal) in place of delta: any positive weight has the same fixed point.
void Engine::polish() {
  constexpr int kPolishMaxSteps = 30;
  const double tp = now_s();
  ensure_host_vectors();
  const bool unsc = settings.scaling && !settings.scaled_termination;
  std::vector<double> z(m), y(m);
  be::d2h(d_, z.data(), d_.z, sizeof(double) * m);
  be::d2h(d_, y.data(), d_.y, sizeof(double) * m);
  // keep the ADMM result
  const OSQPInfo info0 = info;
  const OSQPHipStats stats0 = stats_;
  const double rho0 = rho_bar_, alpha0 = d_.alpha;
  const int eq_from_cnt0 = d_.eq_from_cnt; const double eq_factor0 = d_.rho_eq_factor;
  const double pred0[3] = {slot_pred_[0], slot_pred_[1], slot_pred_[2]};
  const std::vector<double> ls0 = ls_, us0 = us_, y0 = y, z0 = z;
  std::vector<double> hx(n);
  be::d2h(d_, hx.data(), d_.x, sizeof(double) * n);
  // active set (:1719-1720) on the scaled iterates; equality rows are always active
  std::vector<double> lp(m), up(m);
  for (int i = 0; i < m; i++) {
    const bool low = (z[i] - ls0[i] < -y[i]) || ctype_[i] == 1, upp = !low && (us0[i] - z[i] < y[i]);
    if (low) { lp[i] = up[i] = ls0[i]; z[i] = ls0[i]; }
    else if (upp) { lp[i] = up[i] = us0[i]; z[i] = us0[i]; }
    else { lp[i] = -OSQP_INFTY; up[i] = OSQP_INFTY; y[i] = 0.0; }
  }
  auto restore = [&]() {
    d_.alpha = alpha0; drop_graphs();
    for (int k = 0; k < 3; k++) slot_pred_[k] = pred0[k];
    stats_ = stats0;
  };
  double res[R_COUNT];
  try {
    apply_scaled_bounds(lp, up);                      // active rows: equalities at their bound; the others: loose (rho = 1e-6, y = 0)
    be::h2d(d_, d_.z, z.data(), sizeof(double) * m);
    be::h2d(d_, d_.y, y.data(), sizeof(double) * m);
    const double de = std::max(settings.delta, pol_.polish_delta_floor);
    d_.rho_eq_factor = 1.0; d_.eq_from_cnt = 0;       // rho_i = rho_bar = 1 / delta_eff on the active rows
    rho_bar_ = clamp_rho(1.0 / de);
    be::set_rho(d_, rho_bar_);
    be::precond(d_, settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER);
    be::init_iterates(d_, 0);
    d_.alpha = 1.0; drop_graphs();                    // (no relaxation: the refinement recurrence; alpha is baked into captured launches)
    int flags[F_COUNT];
    double best = std::numeric_limits<double>::infinity();
    int worse = 0;
    const int min_steps = 1 + std::max(0, settings.polish_refine_iter);
    for (int s = 0; s < kPolishMaxSteps; s++) {
      be::set_pcg_tol(d_, pol_.polish_pcg_tol, 1e-15);  // ||r|| <= polish_pcg_tol ||rhs||
      exec_chunk_sync(1, kMaxCg, true, 1, res, flags);
      // residuals of the reduced KKT system: Aa x - ba (active rows) and P x + q + Aa' ya, in the scaled space
      const double err = std::max(res[R_PRI_S] / (std::max(res[R_AX_S], res[R_Z_S]) + 1e-30),
                                  res[R_DUA_S] / (std::max(std::max(res[R_ATY_S], res[R_PX_S]), res[R_QN_S]) + 1e-30));
      if (!(err < 0.5 * best)) worse++; else worse = 0;
      best = std::min(best, err);
      if (s + 1 >= min_steps && (err < 1e-13 || worse >= 2)) break;
    }
  } catch (...) {                       // a device failure mid-polish must not leave the polish's weights / bounds on the handle
    restore();
    info = info0; rho_bar_ = rho0; settings.rho = rho0; ls_ = ls0; us_ = us0;
    classify_constraints(ls_, us_);
    throw;
  }
  restore();
  // polished point against the ORIGINAL problem: z = A x, then the normal-cone projection of (z, y)  (:1773-1780)
  info = info0;
  apply_scaled_bounds(ls0, us0);
  be::init_iterates(d_, 1);                        // z = A x_pol
  be::project_normalcone(d_);                      // tmp = z + y; z = clip(tmp, l, u); y = tmp - z
  be::residuals(d_); be::fetch_res(d_, res);
  const double pol_pri = (m == 0) ? 0.0 : (unsc ? res[R_PRI_U] : res[R_PRI_S]);
  const double pol_dua = unsc ? cinv_ * res[R_DUA_U] : res[R_DUA_S];
  const double pol_obj = (0.5 * res[R_XPX] + res[R_QX]) * (settings.scaling ? cinv_ : 1.0);
  const bool ok = (pol_pri < info0.prim_res && pol_dua < info0.dual_res) || (pol_pri < info0.prim_res && info0.dual_res < 1e-10) ||
                  (pol_dua < info0.dual_res && info0.prim_res < 1e-10);                 // :1786-1793
  rho_bar_ = rho0; settings.rho = rho0;
  d_.eq_from_cnt = eq_from_cnt0; if (eq_from_cnt0) d_.rho_eq_factor = eq_factor0;
  be::set_rho(d_, rho_bar_);
  be::precond(d_, settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER);
  if (ok) {
    info.obj_val = pol_obj; info.prim_res = pol_pri; info.dual_res = pol_dua; info.status_polish = 1;       // :1797-1807
    update_gap_info(res, -1.0);
  } else {                                          // keep the ADMM solution (:1813-1814)
    info.status_polish = -1;
    be::h2d(d_, d_.x, hx.data(), sizeof(double) * n);
    be::h2d(d_, d_.y, y0.data(), sizeof(double) * m);
    be::init_iterates(d_, 1);                       // xs = x (PCG warm start), z = A x ...
    be::h2d(d_, d_.z, z0.data(), sizeof(double) * m);   // ... then the ADMM z iterate itself
  }
  be::init_iterates(d_, 0);
  info.polish_time = now_s() - tp;
  if (settings.verbose) std::printf("plsh  %11.4e   %8.2e   %8.2e   --------  (%s)\n", pol_obj, pol_pri, pol_dua, ok ? "accepted" : "rejected");
}

void Engine::store_solution() {                                                          // _osqp.py:1098-1115
  const int st = info.status_val;
  const bool pinf = st == OSQP_PRIMAL_INFEASIBLE || st == OSQP_PRIMAL_INFEASIBLE_INACCURATE;
  const bool dinf = st == OSQP_DUAL_INFEASIBLE || st == OSQP_DUAL_INFEASIBLE_INACCURATE;
  const bool unsc = settings.scaling && !settings.scaled_termination;
  std::fill(sol_pc_.begin(), sol_pc_.end(), kNaN); std::fill(sol_dc_.begin(), sol_dc_.end(), kNaN);
  if (!pinf && !dinf) {
    be::d2h(d_, sol_x_.data(), d_.x, sizeof(double) * n);
    be::d2h(d_, sol_y_.data(), d_.y, sizeof(double) * m);
    if (settings.scaling) {
      for (int j = 0; j < n; j++) sol_x_[j] *= D_[j];
      for (int i = 0; i < m; i++) sol_y_[i] *= cinv_ * E_[i];
    }
  } else {
    std::fill(sol_x_.begin(), sol_x_.end(), kNaN); std::fill(sol_y_.begin(), sol_y_.end(), kNaN);
    if (pinf) {
      be::d2h(d_, sol_pc_.data(), d_.dy, sizeof(double) * m);
      if (unsc) for (int i = 0; i < m; i++) sol_pc_[i] *= E_[i];                         // :1065-1066
    } else {
      be::d2h(d_, sol_dc_.data(), d_.dx, sizeof(double) * n);
      if (unsc) for (int j = 0; j < n; j++) sol_dc_[j] *= D_[j];                         // :1074-1075
    }
  }
  if (reordered_) {                                   // back to the caller's numbering of variables and constraints
    auto back = [](std::vector<double> &v, const std::vector<int> &perm) { std::vector<double> o(v.size()); for (size_t k = 0; k < v.size(); k++) o[perm[k]] = v[k]; v.swap(o); };
    back(sol_x_, pc_); back(sol_dc_, pc_); back(sol_y_, pr_); back(sol_pc_, pr_);
    solution.x = sol_x_.data(); solution.y = sol_y_.data(); solution.prim_inf_cert = sol_pc_.data(); solution.dual_inf_cert = sol_dc_.data();
  }
}

// ------------------------------------------------------------------------------------------------ updates
int Engine::cold_start() {                                                               // _osqp.py:636-642
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  be::activate(d_);
  be::zero(d_, d_.x, sizeof(double) * n); be::zero(d_, d_.z, sizeof(double) * m); be::zero(d_, d_.y, sizeof(double) * m);
  be::init_iterates(d_, 1);
  return OSQP_NO_ERROR;
}

int Engine::warm_start(const double *x, const double *y, bool keep_z) {                  // _osqp.py:1493-1545
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  be::activate(d_);
  settings.warm_starting = 1;
  std::vector<double> xi, yi;
  if (reordered_) { if (x) { xi = to_internal_n(x); x = xi.data(); } if (y) { yi = to_internal_m(y); y = yi.data(); } }
  if (be::device_vec_updates()) {                      // raw vectors go up as they are; x = Dinv x, y = c Einv y on the device
    double *sx = d_.w, *sy = d_.t;                     // PCG work vectors are free between solves
    if (x) be::copy_in(d_, sx, x, sizeof(double) * n, 0);
    if (y) be::copy_in(d_, sy, y, sizeof(double) * m, 0);
    be::scale_warm(d_, x ? sx : nullptr, y ? sy : nullptr, c_);
  } else {
    if (x) {
      std::vector<double> xs(n);
      for (int j = 0; j < n; j++) xs[j] = x[j] * Dinv_[j];
      be::h2d(d_, d_.x, xs.data(), sizeof(double) * n);
    }
    if (y) {
      std::vector<double> ys(m);
      for (int i = 0; i < m; i++) ys[i] = y[i] * Einv_[i] * c_;   // inverse of y = cinv E y_scaled (:1112); the C core includes c
      be::h2d(d_, d_.y, ys.data(), sizeof(double) * m);
    }
  }
  be::init_iterates(d_, keep_z ? 2 : 1);                            // z = A x (:1509); keep_z: the caller has put the z iterate in place
  return OSQP_NO_ERROR;
}

int Engine::warm_start_device(const double *x, const double *y, void *stream) {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  if (!be::device_vec_updates()) return OSQP_FUNC_NOT_IMPLEMENTED;
  be::activate(d_);
  be::ext_wait(d_);                                 // a batch kernel on a caller's stream may still read this solver's vectors
  settings.warm_starting = 1;
  be::stream_wait(d_, stream);
  if (reordered_) {                                 // the caller's numbering -> the engine's, on the device (PCG work vectors are free between solves)
    if (x) { be::gather(d_, d_.w, x, d_pc_, n); x = d_.w; }
    if (y) { be::gather(d_, d_.t, y, d_pr_, m); y = d_.t; }
  }
  be::scale_warm(d_, x, y, c_);
  be::init_iterates(d_, 1);
  return OSQP_NO_ERROR;
}

int Engine::update_data_vec(const double *q, const double *l, const double *u) {          // _osqp.py:1312-1367
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  be::activate(d_);
  be::ext_wait(d_);                                 // a batch kernel on a caller's stream may still read the bounds / q
  double t0 = now_s();
  std::vector<double> qi, li_, ui_;
  if (reordered_) {
    if (q) { qi = to_internal_n(q); q = qi.data(); }
    if (l) { li_ = to_internal_m(l); l = li_.data(); }
    if (u) { ui_ = to_internal_m(u); u = ui_.data(); }
  }
  if (l || u) {
    if (raw_stale_) ensure_host_vectors();
    for (int i = 0; i < m; i++) {
      double li = l ? l[i] : l0_[i], ui = u ? u[i] : u0_[i];
      if (!(li <= ui)) return OSQP_DATA_VALIDATION_ERROR;                                // :1348-1349
    }
  }
  const bool dev = be::device_vec_updates();
  if (q) { q0_.assign(q, q + n); if (dev) be::copy_in(d_, d_.qraw, q, sizeof(double) * n, 0); else upload_q(); }
  if (l) { l0_.assign(l, l + m); if (dev) be::copy_in(d_, d_.lraw, l, sizeof(double) * m, 0); }
  if (u) { u0_.assign(u, u + m); if (dev) be::copy_in(d_, d_.uraw, u, sizeof(double) * m, 0); }
  if (dev) device_scale_vectors(q != nullptr, l || u);
  else if (l || u) upload_bounds_and_types();                                           // update_rho_vec :526-562
  if (l || u) {
    be::set_rho(d_, rho_bar_);
    be::precond(d_, settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER);
  }
  set_status(OSQP_UNSOLVED);                                                             // reset_info :932-941
  if (!dev) be::sync(d_);                           // (device path: everything is stream-ordered; the next solve waits for it)
  update_time_acc_ += now_s() - t0;
  return OSQP_NO_ERROR;
}

// q / l / u given by DEVICE pointer (parametric re-solve with the data produced on the GPU, nn/torch.py:136-140): one device-to-device
// copy per vector, then the same kernels.  The bounds are validated on the device BEFORE anything changes (one 4-byte read-back).
int Engine::update_data_vec_device(const double *q, const double *l, const double *u, void *stream) {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  if (!be::device_vec_updates()) return OSQP_FUNC_NOT_IMPLEMENTED;
  be::activate(d_);
  be::ext_wait(d_);
  double t0 = now_s();
  be::stream_wait(d_, stream);
  if (reordered_ && (l || u) && !(l && u)) {        // one bound in the caller's numbering against the resident other one: bring it over first
    double *tmp = d_.t;                               // (a rejected call leaves the resident vectors untouched: staged in a PCG work vector)
    be::gather(d_, tmp, l ? l : u, d_pr_, m);
    if (be::count_bad_bounds(d_, l ? tmp : d_.lraw, u ? tmp : d_.uraw) > 0) return OSQP_DATA_VALIDATION_ERROR;
  } else
  if ((l || u) && be::count_bad_bounds(d_, l ? l : d_.lraw, u ? u : d_.uraw) > 0) return OSQP_DATA_VALIDATION_ERROR;
  if (reordered_) {                                 // the resident raw vectors are kept in the engine's numbering: gathers instead of copies
    if (q) be::gather(d_, d_.qraw, q, d_pc_, n);
    if (l) be::gather(d_, d_.lraw, l, d_pr_, m);
    if (u) be::gather(d_, d_.uraw, u, d_pr_, m);
  } else {
    if (q) be::copy_in(d_, d_.qraw, q, sizeof(double) * n, 1);
    if (l) be::copy_in(d_, d_.lraw, l, sizeof(double) * m, 1);
    if (u) be::copy_in(d_, d_.uraw, u, sizeof(double) * m, 1);
  }
  if (q || l || u) raw_stale_ = true;
  device_scale_vectors(q != nullptr, l || u);
  if (l || u) {
    be::set_rho(d_, rho_bar_);
    be::precond(d_, settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER);
  }
  set_status(OSQP_UNSOLVED);
  update_time_acc_ += now_s() - t0;
  return OSQP_NO_ERROR;
}

int Engine::update_data_mat(const double *Px, const int *Px_idx, int P_n, const double *Ax, const int *Ax_idx, int A_n) {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  be::activate(d_);
  be::ext_wait(d_);
  double t0 = now_s();
  const int nzP = P_.nnz(), nzA = A_.nnz();
  // bindings.cpp.in:240-281: idx == NULL means all entries in order.  Both arguments are validated BEFORE anything is changed:
  // a rejected call leaves the host copies (and therefore the next upload) untouched.
  if (Px) {
    if (Px_idx) { for (int k = 0; k < P_n; k++) if (Px_idx[k] < 0 || Px_idx[k] >= nzP) return OSQP_DATA_VALIDATION_ERROR; }
    else if (P_n != nzP && P_n != 0) return OSQP_DATA_VALIDATION_ERROR;
  }
  if (Ax) {
    if (Ax_idx) { for (int k = 0; k < A_n; k++) if (Ax_idx[k] < 0 || Ax_idx[k] >= nzA) return OSQP_DATA_VALIDATION_ERROR; }
    else if (A_n != nzA && A_n != 0) return OSQP_DATA_VALIDATION_ERROR;
  }
  // (reordered problem: the caller's positions in its own CSC arrays -> where those entries live in the permuted ones)
  if (Px) for (int k = 0; k < (Px_idx ? P_n : nzP); k++) { const int c = Px_idx ? Px_idx[k] : k; P_.x[reordered_ ? PvalMap_[c] : c] = Px[k]; }
  if (Ax) for (int k = 0; k < (Ax_idx ? A_n : nzA); k++) { const int c = Ax_idx ? Ax_idx[k] : k; A_.x[reordered_ ? AvalMap_[c] : c] = Ax[k]; }
  if (be::device_assembly()) {                                                           // _osqp.py:1443,:1463 on the device
    if (Px) be::h2d(d_, d_.Praw, P_.x.data(), sizeof(double) * nzP);
    if (Ax) be::h2d(d_, d_.Araw, A_.x.data(), sizeof(double) * nzA);
    be::assemble(d_, 1, c_, 1);
    be::f1_refresh(d_); be::wb_refresh(d_); be::wbx_refresh(d_);
  } else {
    std::vector<double> Pxs, Axs;
    scale_matrix_values(Pxs, Axs);
    fill_matrix_values(Pxs, Axs);
  }
  be::precond(d_, settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER);                   // the "refactor" of :1446,:1466,:1488
  be::init_iterates(d_, 0);                                                              // z~, t0 depend on A; iterates untouched
  set_status(OSQP_UNSOLVED);
  be::sync(d_);
  update_time_acc_ += now_s() - t0;
  return OSQP_NO_ERROR;
}

int Engine::update_rho(double rho) {                                                     // _osqp.py:1579-1597
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  be::activate(d_);
  if (!(rho > 0)) return OSQP_SETTINGS_VALIDATION_ERROR;
  rho_bar_ = clamp_rho(rho); settings.rho = rho_bar_;
  be::set_rho(d_, rho_bar_);
  be::precond(d_, settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER);
  return OSQP_NO_ERROR;
}

int Engine::update_settings(const OSQPSettings *s) {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  be::activate(d_);
  int err = validate_settings(s, false);
  if (err) return err;
  // settings that can change after setup (the reference: "These can be changed without running setup", _osqp.py:128-143)
  settings.max_iter = s->max_iter; settings.eps_abs = s->eps_abs; settings.eps_rel = s->eps_rel;
  settings.eps_prim_inf = s->eps_prim_inf; settings.eps_dual_inf = s->eps_dual_inf; settings.alpha = s->alpha;
  settings.scaled_termination = s->scaled_termination; settings.check_termination = s->check_termination;
  settings.check_dualgap = s->check_dualgap; settings.time_limit = s->time_limit; settings.warm_starting = s->warm_starting;
  settings.verbose = s->verbose; settings.polishing = s->polishing; settings.delta = s->delta;
  settings.polish_refine_iter = s->polish_refine_iter; settings.adaptive_rho = s->adaptive_rho;
  settings.adaptive_rho_interval = s->adaptive_rho_interval; settings.adaptive_rho_fraction = s->adaptive_rho_fraction;
  settings.adaptive_rho_tolerance = s->adaptive_rho_tolerance; settings.cg_max_iter = s->cg_max_iter;
  settings.cg_tol_reduction = s->cg_tol_reduction; settings.cg_tol_fraction = s->cg_tol_fraction;
  if (s->cg_precond != settings.cg_precond) { settings.cg_precond = s->cg_precond; be::precond(d_, settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER); }
  if (d_.alpha != settings.alpha) { d_.alpha = settings.alpha; drop_graphs(); }     // alpha is baked into captured launches
  have_tol_ = false; cg_budget_ = 0;
  return OSQP_NO_ERROR;
}

// Batch of nbatch QPs that share this solver's (P, A, scaling, settings) and differ in q / l / u -- the reference's
// update-style batching (nn/torch.py:136-164: update(q,l,u) + solve() per element) as ONE kernel launch.
// q: nbatch x n, l/u: nbatch x m (row-major; NULL = this solver's current vector for every problem);
// x: nbatch x n, y: nbatch x m (in: unscaled warm start if warm != 0; out: solution, or certificate for infeasible ones);
// rec: nbatch x kBatchRec = {status_val, iter, obj_val, prim_res, dual_res, rho, rho_updates, pcg_iters, status_polish, polish_time, rho_estimate, reserved}.


// ------------------------------------------------------------------------------------------------ LinSysSolver slot
// The reduced-KKT PCG as a stand-alone linear solver (include/osqp_hip.h), built from the same backend
// operations as the ADMM loop:  kb_rhs  forms  rhs = sigma x - q + A' v  and the PCG start residual, so with  x = 0,
// q = -rhs_x,  v = rho .* rhs_z  it forms exactly the right-hand side of the reduced system;  k1/k2/kv  are the PCG
// iterations (three-kernel form: a solve may be continued past its first budget);  init_iterates(0)  leaves  z~ = A x~.
int Engine::ls_setup(const OSQPCscMatrix *P, const OSQPCscMatrix *A, const double *rho_vec, const OSQPSettings *s) {
  if (!P || !A || !rho_vec || !s) return OSQP_DATA_VALIDATION_ERROR;
  OSQPSettings st = *s;
  st.scaling = 0; st.linsys_solver = OSQP_INDIRECT_SOLVER; st.verbose = 0; st.polishing = 0;   // the matrices arrive scaled
  const int nn = P->n, mm = A->m;
  std::vector<double> q(nn, 0.0), l(mm, -OSQP_INFTY), u(mm, OSQP_INFTY);
  no_reorder_ = true;                                 // (the slot's vectors -- rhs, rho_vec, warm start -- are exchanged in the caller's numbering)
  int err = setup(P, q.data(), A, l.data(), u.data(), mm, nn, &st);
  if (err) return err;
  d_.fused = 0; d_.f1.on = 0; d_.wb.on = 0;
  return ls_set_rho_vec(rho_vec);
}

int Engine::ls_set_rho_vec(const double *rho_vec) {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  if (!rho_vec) return OSQP_DATA_VALIDATION_ERROR;
  be::activate(d_);
  ls_rho_.assign(rho_vec, rho_vec + m);
  std::vector<double> rinv(m);
  for (int i = 0; i < m; i++) { if (!(ls_rho_[i] > 0)) return OSQP_DATA_VALIDATION_ERROR; rinv[i] = 1.0 / ls_rho_[i]; }
  be::h2d(d_, d_.rho, ls_rho_.data(), sizeof(double) * m);
  be::h2d(d_, d_.rho_inv, rinv.data(), sizeof(double) * m);
  be::precond(d_, settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER);
  be::init_iterates(d_, 0);                        // t0 = rho .* (A x~) must match the new rho
  be::sync(d_);
  return OSQP_NO_ERROR;
}

int Engine::ls_warm_start(const double *x) {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  if (!x) return OSQP_DATA_VALIDATION_ERROR;
  be::activate(d_);
  be::h2d(d_, d_.xs, x, sizeof(double) * n);
  be::init_iterates(d_, 0);
  be::sync(d_);
  return OSQP_NO_ERROR;
}

int Engine::ls_solve(double *b, double tol_rel, double tol_abs, int *iters) {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  if (!b) return OSQP_DATA_VALIDATION_ERROR;
  be::activate(d_);
  std::vector<double> nq(n), v(m);
  for (int j = 0; j < n; j++) nq[j] = -b[j];
  for (int i = 0; i < m; i++) v[i] = ls_rho_[i] * b[n + i];
  be::h2d(d_, d_.q, nq.data(), sizeof(double) * n);
  be::h2d(d_, d_.v, v.data(), sizeof(double) * m);
  be::zero(d_, d_.x, sizeof(double) * n);
  be::set_pcg_tol(d_, tol_rel, tol_abs);
  be::kb_rhs(d_);
  const int cap = std::min(settings.cg_max_iter, kMaxCg);
  int flags[F_COUNT] = {0};
  int done = 0;
  for (int i0 = 0; i0 < cap && !done;) {           // budget: what the previous solve needed + 2, then doubling
    const int bud = std::min(cap - i0, std::max(4, i0 == 0 ? cg_budget_ + 2 : i0));
    for (int i = i0; i < i0 + bud; i++) { be::k1(d_, i); be::k2(d_, i); be::kv(d_, i); }
    i0 += bud;
    be::k1(d_, i0 < cap ? i0 : cap);               // the stopping test of the last update (its SpMV is wasted only if the cap was hit)
    be::fetch_flags(d_, flags);
    done = flags[F_DONE];
    if (!done && i0 >= cap) break;
    if (!done) { be::k2(d_, i0); be::kv(d_, i0); i0++; }
  }
  cg_budget_ = done ? flags[F_ITERS] : cap;
  if (iters) *iters = cg_budget_;
  be::init_iterates(d_, 0);                        // z~ = A x~ ; t0 for the next solve's start residual
  be::d2h(d_, b, d_.xs, sizeof(double) * n);
  if (m > 0) be::d2h(d_, b + n, d_.zt, sizeof(double) * m);
  return OSQP_NO_ERROR;
}

// ------------------------------------------------------------------------------------------------ batch path, direct solve
// Symbolic preparation of the banded-Cholesky linear solve of the batch kernel (batch_hip.hip): the pattern of
// K = P + sigma I + A' diag(rho) A, a reverse Cuthill-McKee ordering of it, the band slot of every P entry, and for
// every band slot the list of products A_ia A_ib that rho_i multiplies.  The reference's builtin algebra factorises the
// KKT matrix with QDLDL after an AMD ordering; for QPs small enough to live in one workgroup's LDS the
// reduced matrix K (n x n, SPD) under a BANDWIDTH-reducing ordering is the better fit: no indirect addressing in the
// factor, fixed trip counts.
void Engine::free_batch_direct() {
  void *ptrs[] = {bd_.perm, bd_.bp_slot, bd_.ke_slot, bd_.ke_ptr, bd_.kp_row, bd_.kp_a, bd_.kp_b, bd_.tri, bd_.kp_val};
  for (void *p : ptrs) if (p) be::dfree(d_, p);
  bd_ = BatchDirect();
}

void Engine::prepare_batch_direct() {
  if (bd_.tried) return;
  bd_.tried = true;
  const int nzA = (int)Arj_.size(), nzB = (int)Bj_.size();
  // adjacency of K (excluding the diagonal)
  double pairs = 0;
  for (int i = 0; i < m; i++) { const double len = Arp_[i + 1] - Arp_[i]; pairs += len * (len + 1) / 2; }
  if (pairs > 4e6 || n > 4096) { bd_.bw_symbolic = -2; return; }   // dense rows: K would be (nearly) dense -- PCG path
  std::vector<std::vector<int>> adj(n);
  for (int j = 0; j < n; j++)
    for (int k = Brp_[j]; k < Brp_[j + 1]; k++) { const int c = Bj_[k]; if (c < n && c != j) adj[j].push_back(c); }
  for (int i = 0; i < m; i++)
    for (int a = Arp_[i]; a < Arp_[i + 1]; a++)
      for (int b = Arp_[i]; b < Arp_[i + 1]; b++) if (a != b) adj[Arj_[a]].push_back(Arj_[b]);
  for (auto &v : adj) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
  // reverse Cuthill-McKee, component by component, each started from a pseudo-peripheral node
  std::vector<int> order; order.reserve(n);
  std::vector<char> seen(n, 0);
  std::vector<int> level(n, -1), frontier, next;
  auto bfs_far = [&](int start, int &ecc) {                  // farthest node of minimum degree from start (within its component)
    std::vector<int> touched;
    frontier.assign(1, start); level[start] = 0; touched.push_back(start);
    int last = start; ecc = 0;
    while (!frontier.empty()) {
      next.clear();
      int best = frontier[0];
      for (int v : frontier) if (adj[v].size() < adj[best].size()) best = v;
      last = best; ecc = level[best];
      for (int v : frontier) for (int w : adj[v]) if (level[w] < 0) { level[w] = level[v] + 1; next.push_back(w); touched.push_back(w); }
      frontier.swap(next);
    }
    for (int v : touched) level[v] = -1;
    return last;
  };
  for (int s0 = 0; s0 < n; s0++) {
    if (seen[s0]) continue;
    int start = s0, ecc = -1;
    for (int rounds = 0; rounds < 8; rounds++) {            // pseudo-peripheral node (George-Liu)
      int e2; const int far = bfs_far(start, e2);
      if (e2 <= ecc) break;
      ecc = e2; start = far;
    }
    size_t head = order.size();
    order.push_back(start); seen[start] = 1;
    while (head < order.size()) {
      const int v = order[head++];
      std::vector<int> nb;
      for (int w : adj[v]) if (!seen[w]) { seen[w] = 1; nb.push_back(w); }
      std::sort(nb.begin(), nb.end(), [&](int a, int b) { return adj[a].size() != adj[b].size() ? adj[a].size() < adj[b].size() : a < b; });
      order.insert(order.end(), nb.begin(), nb.end());
    }
  }
  std::reverse(order.begin(), order.end());
  std::vector<int> iperm(n);
  for (int k = 0; k < n; k++) iperm[order[k]] = k;
  int bw = 0;
  for (int j = 0; j < n; j++) for (int c : adj[j]) bw = std::max(bw, std::abs(iperm[j] - iperm[c]));
  bd_.bw_symbolic = bw;
  const int W = bw + kBatchNB;                              // column stride of the padded band (batch_hip.hip)
  if (bw > kBatchDirectMaxBw || !be::batch_direct_lds_bytes(n, m, std::max(nzA, nzB), bw)) return;
  // band slot (column-major band: slot = col * W + (row - col), row >= col, permuted indices) of the P + sigma I entries of B
  std::vector<int> bp_slot(nzB, -1);
  for (int j = 0; j < n; j++)
    for (int k = Brp_[j]; k < Brp_[j + 1]; k++) {
      const int c = Bj_[k];
      if (c >= n) continue;
      const int pr = iperm[j], pc = iperm[c];
      if (pr >= pc) bp_slot[k] = pc * W + (pr - pc);
    }
  // products of A' rho A, grouped by slot
  struct Prod { int slot, row, a, b; };
  std::vector<Prod> prods; prods.reserve((size_t)pairs);
  for (int i = 0; i < m; i++)
    for (int a = Arp_[i]; a < Arp_[i + 1]; a++)
      for (int b = a; b < Arp_[i + 1]; b++) {
        const int pa = iperm[Arj_[a]], pb = iperm[Arj_[b]];
        const int r = std::max(pa, pb), c = std::min(pa, pb);
        prods.push_back({c * W + (r - c), i, a, b});
      }
  std::stable_sort(prods.begin(), prods.end(), [](const Prod &x, const Prod &y) { return x.slot < y.slot; });
  std::vector<int> ke_slot, ke_ptr, kp_row(prods.size()), kp_a(prods.size()), kp_b(prods.size());
  for (size_t p = 0; p < prods.size(); p++) {
    if (p == 0 || prods[p].slot != prods[p - 1].slot) { ke_slot.push_back(prods[p].slot); ke_ptr.push_back((int)p); }
    kp_row[p] = prods[p].row; kp_a[p] = prods[p].a; kp_b[p] = prods[p].b;
  }
  ke_ptr.push_back((int)prods.size());
  std::vector<int> tri;
  for (int a = 1; a <= bw; a++) for (int b = a; b <= bw; b++) tri.push_back(a | (b << 8));
  auto up = [&](const std::vector<int> &h) { int *p = dev_vec<int>(d_, h.size()); if (!h.empty()) be::h2d(d_, p, h.data(), sizeof(int) * h.size()); return p; };
  bd_.perm = up(order); bd_.bp_slot = up(bp_slot); bd_.ke_slot = up(ke_slot); bd_.ke_ptr = up(ke_ptr);
  bd_.kp_row = up(kp_row); bd_.kp_a = up(kp_a); bd_.kp_b = up(kp_b); bd_.tri = up(tri);
  bd_.kp_val = dev_vec<double>(d_, prods.size());
  bd_.bw = bw; bd_.nents = (int)ke_slot.size(); bd_.nprod = (int)prods.size(); bd_.ntri = (int)tri.size();
  bd_.ok = true;
}

void Engine::fill_batch_params(BatchParams &p, int nbatch, int warm) {
  p.n = n; p.m = m; p.nbatch = nbatch; p.A = d_.A; p.B = d_.B; p.D = d_.D; p.Dinv = d_.Dinv; p.E = d_.E; p.Einv = d_.Einv;
  p.c = c_; p.cinv = cinv_; p.sigma = settings.sigma; p.alpha = settings.alpha; p.rho0 = clamp_rho(settings.rho); p.eq_factor = eq_factor_mixed_;
  p.eps_abs = settings.eps_abs; p.eps_rel = settings.eps_rel; p.eps_pinf = settings.eps_prim_inf; p.eps_dinf = settings.eps_dual_inf;
  p.cg_frac = settings.cg_tol_fraction; p.rho_tol = settings.adaptive_rho_tolerance;
  p.max_iter = settings.max_iter; p.check = settings.check_termination; p.rho_interval = settings.adaptive_rho ? auto_rho_interval() : 0;
  p.cg_max = settings.cg_max_iter; p.unscaled = settings.scaling && !settings.scaled_termination; p.scaling = settings.scaling;
  p.precond = settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER; p.rho_is_vec = settings.rho_is_vec; p.warm = warm;
  p.polish = settings.polishing; p.refine = settings.polish_refine_iter; p.delta = settings.delta;      // (honoured by the direct variants)
  p.variant = pol_.batch_variant;
}

void Engine::attach_batch_direct(BatchParams &p) {
  if (!bd_.ok) return;
  p.eq_factor_direct = eq_factor_set_ ? eq_factor_mixed_ : 1e3;
  p.bw = bd_.bw; p.nents = bd_.nents; p.ntri = bd_.ntri; p.perm = bd_.perm; p.bp_slot = bd_.bp_slot; p.ke_slot = bd_.ke_slot;
  p.ke_ptr = bd_.ke_ptr; p.kp_row = bd_.kp_row; p.kp_val = bd_.kp_val; p.tri = bd_.tri;
}

// ------------------------------------------------------------------------------------------------ small problems
// A QP small enough for the batch kernel's DIRECT variant (iterates, matrices and the banded LDL' factor of the reduced
// KKT matrix in one workgroup's LDS) is solved by ONE launch of that kernel with a batch of one: the whole ADMM loop runs
// on the device with exact linear solves and the reference's rho rule, i.e. the algorithm of the reference's direct path
// (same iteration counts as the oracle), instead of thousands of graph-replayed multi-kernel iterations with inexact
// inner solves -- on small LPs / rank-deficient QPs the latter can need 10x more ADMM iterations (fuzz).
// With `polishing`, a SOLVED problem is polished in the same launch (reduced KKT system on the active set, factorised in LDS,
// polish_refine_iter refinement steps: the reference's algorithm, _osqp.py:1710-1828).  Not taken with verbose
// output (per-iteration printing lives in the host-driven loop), with a time limit, or when OSQP_HIP_SMALL_DIRECT=0.
bool Engine::small_direct_applicable() {
  if (!pol_.small_direct || !be::device_assembly() || settings.verbose || settings.time_limit < 1e9 || reordered_) return false;
  if (settings.check_dualgap) return false;            // the one-launch kernel has no duality-gap test: the host-driven loop honours the setting
  if (!be::batch_lds_bytes(n, m)) return false;
  prepare_batch_direct();
  if (!bd_.ok) return false;
  BatchParams p{};
  fill_batch_params(p, 1, 0);
  attach_batch_direct(p);
  return be::batch_direct_selected(p);
}

int Engine::solve_small_direct(double t0) {
  const int warm = settings.warm_starting ? 1 : 0;
  std::vector<double> x(n, 0.0), y(std::max(m, 1), 0.0);       // (m = 0: batch_solve still wants a non-null y)
  if (warm) {                                                   // continue from the device iterates (x, y; z = A x as in warm_start)
    be::d2h(d_, x.data(), d_.x, sizeof(double) * n);
    if (m > 0) be::d2h(d_, y.data(), d_.y, sizeof(double) * m);
    for (int j = 0; j < n; j++) x[j] *= D_[j];
    for (int i = 0; i < m; i++) y[i] *= cinv_ * E_[i];
  }
  double rec[kBatchRec] = {0};
  // (the handle's own scaled z goes in and out by device pointer: a continued solve keeps its z iterate, _osqp.py:1197-1204)
  const int err = batch_solve(1, nullptr, nullptr, nullptr, x.data(), y.data(), rec, warm, m > 0 ? d_.z : nullptr);
  if (err) return err;
  const int st = (int)rec[0];
  set_status(st);
  info.iter = (int)rec[1]; info.obj_val = rec[2]; info.prim_res = rec[3]; info.dual_res = rec[4];
  info.rho_updates = (int)rec[6]; info.rho_estimate = rec[10];                 // (_osqp.py:1275)
  info.status_polish = (int)rec[8]; info.polish_time = rec[9];      // polished inside the kernel (reduced KKT on the factor in LDS)
  if (rec[5] != rho_bar_) {                                     // adaptive rho moved: keep the handle's state in step (_osqp.py:923-930)
    rho_bar_ = clamp_rho(rec[5]); settings.rho = rho_bar_;
    be::set_rho(d_, rho_bar_);
    be::precond(d_, settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER);
  }
  const bool pinf = st == OSQP_PRIMAL_INFEASIBLE || st == OSQP_PRIMAL_INFEASIBLE_INACCURATE;
  const bool dinf = st == OSQP_DUAL_INFEASIBLE || st == OSQP_DUAL_INFEASIBLE_INACCURATE;
  std::fill(sol_pc_.begin(), sol_pc_.end(), kNaN); std::fill(sol_dc_.begin(), sol_dc_.end(), kNaN);
  const bool finite_xy = std::isfinite(rec[2]) && st != OSQP_NON_CVX;
  if (!pinf && !dinf) {
    std::copy(x.begin(), x.end(), sol_x_.begin()); std::copy(y.begin(), y.begin() + m, sol_y_.begin());   // (solution.x/y point into these)
    if (finite_xy) {
      const int keep = settings.warm_starting;
      warm_start(x.data(), m > 0 ? y.data() : nullptr, /*keep_z=*/true);   // device x, y follow; z is the kernel's own (a later solve continues from them)
      settings.warm_starting = keep;
      // the v1 gap fields (update_gap_info) from the unscaled data on the host: a few hundred entries
      ensure_host_vectors();
      std::vector<double> px(n, 0.0), ax(m, 0.0), aty(n, 0.0);
      for (int j = 0; j < n; j++)
        for (int k = P_.p[j]; k < P_.p[j + 1]; k++) { const int i = P_.i[k]; px[i] += P_.x[k] * x[j]; if (i != j) px[j] += P_.x[k] * x[i]; }
      for (int j = 0; j < n; j++)
        for (int k = A_.p[j]; k < A_.p[j + 1]; k++) { ax[A_.i[k]] += A_.x[k] * x[j]; aty[j] += A_.x[k] * y[A_.i[k]]; }
      double xpx = 0, sup = 0, nax = 0, nz = 0, npx = 0, naty = 0, nq = 0;
      for (int j = 0; j < n; j++) { xpx += x[j] * px[j]; npx = std::max(npx, std::fabs(px[j])); naty = std::max(naty, std::fabs(aty[j])); nq = std::max(nq, std::fabs(q0_[j])); }
      for (int i = 0; i < m; i++) {
        if (y[i] > 0 && u0_[i] < OSQP_INFTY * kMinScaling) sup += u0_[i] * y[i];
        else if (y[i] < 0 && l0_[i] > -OSQP_INFTY * kMinScaling) sup += l0_[i] * y[i];
        nax = std::max(nax, std::fabs(ax[i])); nz = std::max(nz, std::fabs(std::min(std::max(ax[i], l0_[i]), u0_[i])));
      }
      info.dual_obj_val = -0.5 * xpx - sup;
      info.duality_gap = info.obj_val - info.dual_obj_val;
      const double tiny = 1e-10, gn = std::max(std::fabs(info.obj_val), std::fabs(info.dual_obj_val));
      info.rel_kkt_error = std::max(std::max(m == 0 ? 0.0 : info.prim_res / (std::max(nax, nz) + tiny), info.dual_res / (std::max(std::max(npx, naty), nq) + tiny)),
                                    std::fabs(info.duality_gap) / (gn + tiny));
    } else {
      cold_start();                                             // NaN iterates (non-convex problem) are no warm start
      info.dual_obj_val = info.duality_gap = info.rel_kkt_error = kNaN;
    }
  } else {
    std::fill(sol_x_.begin(), sol_x_.end(), kNaN); std::fill(sol_y_.begin(), sol_y_.end(), kNaN);
    if (pinf) std::copy(y.begin(), y.begin() + m, sol_pc_.begin()); else std::copy(x.begin(), x.end(), sol_dc_.begin());                    // the kernel returns the certificate in place of y / x
    cold_start();
  }
  stats_.pcg_iters_total = stats_.pcg_iters_max = stats_.pcg_unconverged = 0;
  stats_.kernel_launches = 1; stats_.graph_launches = 0;
  be::sync(d_);
  info.solve_time = std::max(now_s() - t0 - info.polish_time, 0.0);
  info.run_time = (first_run_ ? info.setup_time : info.update_time) + info.solve_time + info.polish_time;
  first_run_ = false; clear_update_time_ = true;
  return OSQP_NO_ERROR;
}

int Engine::batch_solve(int nbatch, const double *q, const double *l, const double *u, double *x, double *y, double *rec, int warm, double *zs_dev) {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  if (nbatch <= 0 || !x || !y || !rec) return OSQP_DATA_VALIDATION_ERROR;
  prepare_batch_direct();                                     // (symbolic part runs on every backend: tests read the bandwidth)
  if (!be::batch_lds_bytes(n, m) || reordered_) return OSQP_FUNC_NOT_IMPLEMENTED;      // (a reordered handle is a large single QP: the batch kernel is for QPs that fit one workgroup)
  be::activate(d_);
  be::ext_wait(d_);                                 // the scratch block may still be read by a kernel on a caller's stream
  const bool timing = pol_.batch_timing != 0;
  double tph[5]; tph[0] = now_s();
  const size_t N = (size_t)nbatch * n, M = (size_t)nbatch * m;
  if ((l || u) && !(l && u)) ensure_host_vectors();
  for (int b = 0; b < nbatch && (l || u); b++)                                                       // _osqp.py:1348-1349
    for (int i = 0; i < m; i++) {
      const double li = l ? l[(size_t)b * m + i] : l0_[i], ui = u ? u[(size_t)b * m + i] : u0_[i];
      if (!(li <= ui)) return OSQP_DATA_VALIDATION_ERROR;
    }
  tph[1] = now_s();
  // one device scratch block, kept for the next call: [q | l | u | x | y | rec | q0 | l0 | u0]
  const size_t need = 2 * N + 3 * M + (size_t)nbatch * kBatchRec + n + 2 * (size_t)m;
  if (need > bbuf_cap_) { if (bbuf_) be::dfree(d_, bbuf_); bbuf_ = dev_vec<double>(d_, need); bbuf_cap_ = need; }
  double *dq = bbuf_, *dl = dq + N, *du = dl + M, *dx = du + M, *dy = dx + N, *drec = dy + M, *dq0 = drec + (size_t)nbatch * kBatchRec, *dl0 = dq0 + n, *du0 = dl0 + m;
  const bool devv = be::device_vec_updates();         // then the solver's own q, l, u are resident (unscaled): no upload for NULL arguments
  if (q) be::h2d(d_, dq, q, sizeof(double) * N); else if (devv) dq0 = d_.qraw; else be::h2d(d_, dq0, q0_.data(), sizeof(double) * n);
  if (l) be::h2d(d_, dl, l, sizeof(double) * M); else if (devv) dl0 = d_.lraw; else be::h2d(d_, dl0, l0_.data(), sizeof(double) * m);
  if (u) be::h2d(d_, du, u, sizeof(double) * M); else if (devv) du0 = d_.uraw; else be::h2d(d_, du0, u0_.data(), sizeof(double) * m);
  if (warm) { be::h2d(d_, dx, x, sizeof(double) * N); be::h2d(d_, dy, y, sizeof(double) * M); }
  be::sync(d_); tph[2] = now_s();
  BatchParams p{};
  fill_batch_params(p, nbatch, warm);
  p.q = q ? dq : nullptr; p.l = l ? dl : nullptr; p.u = u ? du : nullptr; p.q0 = dq0; p.l0 = dl0; p.u0 = du0; p.x = dx; p.y = dy; p.rec = drec;
  p.zs = zs_dev;
  // Launch order: the problems that took most iterations in the PREVIOUS call of the same size go first (parametric batches -- MPC
  // steps, training epochs -- repeat their hard problems; with index order the last round of workgroups waits for stragglers:
  // 4096 MPC QPs 13.3 -> 11 ms).  Scheduling only: every problem is solved by its own workgroup exactly as before.
  const bool reorder = pol_.batch_reorder != 0;
  if (reorder && nbatch > 1 && (int)batch_order_.size() == nbatch) {
    if ((size_t)nbatch > batch_order_cap_) {
      if (d_batch_order_) be::dfree(d_, d_batch_order_);
      if (d_batch_iters_) { be::dfree(d_, d_batch_iters_); d_batch_iters_ = nullptr; d_batch_iters_n_ = 0; }
      d_batch_order_ = dev_vec<int>(d_, nbatch); batch_order_cap_ = nbatch;
    }
    be::h2d(d_, d_batch_order_, batch_order_.data(), sizeof(int) * nbatch);
    p.order = d_batch_order_;
  }
  d_batch_iters_n_ = 0;                            // (the device-pointer path's history does not describe this call)
  prepare_batch_direct();
  if (bd_.ok) {
    be::batch_products(d_, bd_.nprod, bd_.kp_a, bd_.kp_b, bd_.kp_val);             // A's values may have changed since the last call
    attach_batch_direct(p);
  }
  int err = be::batch_solve(d_, p);
  tph[3] = now_s();
  if (!err) {
    be::d2h(d_, x, dx, sizeof(double) * N); be::d2h(d_, y, dy, sizeof(double) * M); be::d2h(d_, rec, drec, sizeof(double) * kBatchRec * nbatch);
    if (reorder && nbatch > 1) {
      batch_order_.resize(nbatch);
      for (int b = 0; b < nbatch; b++) batch_order_[b] = b;
      std::stable_sort(batch_order_.begin(), batch_order_.end(), [&](int a, int b) { return rec[(size_t)a * kBatchRec + 1] > rec[(size_t)b * kBatchRec + 1]; });
    }
  }
  tph[4] = now_s();
  stats_.gpu_solve_ms = 1e3 * (tph[3] - tph[2]);
  if (timing) std::fprintf(stderr, "osqp_hip batch: validate %.2f ms, H2D %.2f ms, kernel %.2f ms, D2H %.2f ms\n", 1e3 * (tph[1] - tph[0]), 1e3 * (tph[2] - tph[1]), 1e3 * (tph[3] - tph[2]), 1e3 * (tph[4] - tph[3]));
  return err;
}


// Device-resident variant: q, l, u, x, y, rec are device pointers on this solver's device; the kernel is
// enqueued on the caller's stream and not waited for (stream == nullptr: the solver's stream, synchronous).
int Engine::batch_solve_device(int nbatch, const double *q, const double *l, const double *u, double *x, double *y, double *rec, int warm, void *stream) {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  // nbatch == 0: the applicability query of a rank whose share of a sharded batch is empty -- the answer depends on (n, m) alone, so every
  // rank of a job reaches the same decision before its first collective (osqp_amd/sharded.py)
  if (nbatch == 0) return (be::batch_lds_bytes(n, m) && !reordered_) ? OSQP_NO_ERROR : OSQP_FUNC_NOT_IMPLEMENTED;
  if (nbatch < 0 || !x || !y || !rec) return OSQP_DATA_VALIDATION_ERROR;
  if (!be::batch_lds_bytes(n, m) || reordered_) return OSQP_FUNC_NOT_IMPLEMENTED;
  be::activate(d_);
  be::ext_wait(d_);                                 // the previous device-pointer call: its kernel reads the shared vectors and kp_val
  // shared vectors (for the arguments given as NULL): the solver's own resident unscaled q, l, u
  double *dq0 = d_.qraw, *dl0 = d_.lraw, *du0 = d_.uraw;
  BatchParams p{};
  fill_batch_params(p, nbatch, warm);
  p.q = q; p.l = l; p.u = u; p.q0 = dq0; p.l0 = dl0; p.u0 = du0; p.x = x; p.y = y; p.rec = rec;
  // launch order as in batch_solve, entirely on the device: the kernel leaves every problem's iteration count, a rank kernel turns
  // the previous call's counts into this call's order (both on the caller's stream: ordered with the batch kernels themselves)
  const bool reorder = pol_.batch_reorder != 0;
  if (reorder && nbatch > 1) {
    if ((size_t)nbatch > batch_order_cap_) {           // (both buffers have the same capacity)
      if (d_batch_order_) be::dfree(d_, d_batch_order_);
      if (d_batch_iters_) be::dfree(d_, d_batch_iters_);
      d_batch_order_ = dev_vec<int>(d_, nbatch); d_batch_iters_ = nullptr; batch_order_cap_ = nbatch; d_batch_iters_n_ = 0;
    }
    if (!d_batch_iters_) { d_batch_iters_ = dev_vec<int>(d_, batch_order_cap_); d_batch_iters_n_ = 0; }
    be::sync(d_);                                    // (allocations / zero fills ran on the solver's stream)
    if (d_batch_iters_n_ == nbatch) { be::batch_order(d_, nbatch, d_batch_iters_, d_batch_order_, stream); p.order = d_batch_order_; }
    p.iters_out = d_batch_iters_;
    d_batch_iters_n_ = nbatch;
    batch_order_.clear();                            // (the host path's order does not describe this call)
  }
  prepare_batch_direct();
  if (bd_.ok) {
    be::batch_products(d_, bd_.nprod, bd_.kp_a, bd_.kp_b, bd_.kp_val);
    attach_batch_direct(p);
  }
  be::sync(d_);                                   // the uploads and the product refresh ran on the solver's stream
  const int err = be::batch_solve(d_, p, stream);
  if (!err) be::ext_record(d_, stream);           // later calls that overwrite or free what this kernel reads wait for it (ext_wait)
  return err;
}

int Engine::get_stats(OSQPHipStats *out) {
  if (!out) return OSQP_DATA_VALIDATION_ERROR;
  *out = stats_; out->pcg_fused = (d_.f1.on && use_slots_) ? 2.0 : (be::pcg_fused(d_) ? 1.0 : 0.0); out->batch_direct_bw = bd_.bw_symbolic;
  out->f1_replicas = d_.f1.on ? d_.f1.D : 0;
  out->woodbury_rows = d_.wb.on ? d_.wb.r : 0; out->woodbury_direct = (d_.wb.on && d_.wb.exact) ? ((d_.wb.x.on) ? 2 : 1) : 0;
  out->windowed_blocks = d_.A.nwin + d_.B.nwin; out->row_blocks = d_.A.nblk + d_.B.nblk;
  out->reordered = reordered_ ? 1.0 : 0.0; out->reorder_ms = reorder_ms_;
  // which preconditioner the PCG of this handle runs with RIGHT NOW (the setting cg_precond = diagonal selects the Jacobi family; the
  // Woodbury correction for dense rows is the engine's addition: OSQPHipPolicy::woodbury / woodbury_large switch it off)
  out->preconditioner = settings.cg_precond != OSQP_DIAGONAL_PRECONDITIONER ? OSQP_HIP_PRECOND_NONE
                        : !d_.wb.on ? OSQP_HIP_PRECOND_JACOBI : (d_.wb.large ? OSQP_HIP_PRECOND_JACOBI_WOODBURY_DENSE : OSQP_HIP_PRECOND_JACOBI_WOODBURY);
  return OSQP_NO_ERROR;
}
int Engine::time_kernel(int which, int reps, double *ms) {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  be::activate(d_);
  if (which < 0 || which > 16 || reps <= 0 || !ms) return OSQP_DATA_VALIDATION_ERROR;
  *ms = be::time_kernel(d_, which, reps);
  return OSQP_NO_ERROR;
}
int Engine::trace_read(unsigned long long *out, int count) {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  if (!out || count <= 0) return OSQP_DATA_VALIDATION_ERROR;
  return be::ktrace_read(d_, out, count) ? OSQP_NO_ERROR : OSQP_FUNC_NOT_IMPLEMENTED;
}
int Engine::test_spmv(int which, const double *in, double *out) {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  be::activate(d_);
  const int nin = which == 0 ? n : n + m, nout = which == 0 ? m : n;
  double *din = dev_vec<double>(d_, nin), *dout = dev_vec<double>(d_, nout);
  std::vector<double> hin(in, in + nin), hout(nout);
  if (reordered_) {                                   // (vectors of the caller's numbering, like everything else at the API)
    for (int j = 0; j < n; j++) hin[j] = in[pc_[j]];
    if (which != 0) for (int i = 0; i < m; i++) hin[n + i] = in[n + pr_[i]];
  }
  be::h2d(d_, din, hin.data(), sizeof(double) * nin);
  be::test_spmv(d_, which, din, dout);
  be::d2h(d_, hout.data(), dout, sizeof(double) * nout);
  for (int k = 0; k < nout; k++) out[reordered_ ? (which == 0 ? pr_[k] : pc_[k]) : k] = hout[k];
  be::dfree(d_, din); be::dfree(d_, dout);
  return OSQP_NO_ERROR;
}
int Engine::get_reordering(int *perm_cols, int *perm_rows) const {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  if (!perm_cols || (m > 0 && !perm_rows)) return OSQP_DATA_VALIDATION_ERROR;
  for (int j = 0; j < n; j++) perm_cols[j] = reordered_ ? pc_[j] : j;
  for (int i = 0; i < m; i++) perm_rows[i] = reordered_ ? pr_[i] : i;
  return OSQP_NO_ERROR;
}
int Engine::get_scaling(double *D, double *E, double *c) {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  for (int j = 0; j < n; j++) D[reordered_ ? pc_[j] : j] = D_[j];
  for (int i = 0; i < m; i++) E[reordered_ ? pr_[i] : i] = E_[i];
  *c = c_;
  return OSQP_NO_ERROR;
}

}  // namespace osqp_hip
