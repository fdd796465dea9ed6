// engine.cpp -- see engine.hpp.  Formulas cite /root/reference/src/osqppurepy/_osqp.py ("_osqp.py:LINE"), the only
// in-tree statement of the algorithm the reference's C core executes.


#include "engine_internal.hpp"

#include <atomic>
#include <cstdarg>

namespace osqp_hip {

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
  num("OSQP_HIP_BATCH_WAVE", p.batch_wave);
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
  num("OSQP_HIP_WOODBURY_FUSED", p.woodbury_fused); real("OSQP_HIP_WOODBURY_DIRECT_TOL", p.woodbury_direct_tol);
  on("OSQP_HIP_WOODBURY", p.woodbury); on("OSQP_HIP_WOODBURY_DIRECT", p.woodbury_direct); on("OSQP_HIP_WOODBURY_LARGE", p.woodbury_large);
  num("OSQP_HIP_REORDER", p.reorder); on("OSQP_HIP_WOODBURY_CACHE", p.woodbury_cache); on("OSQP_HIP_KFORM", p.kform); on("OSQP_HIP_WOODBURY_DUAL", p.woodbury_dual); on("OSQP_HIP_WOODBURY_VENDOR", p.woodbury_vendor);
  on("OSQP_HIP_GRAPH", p.graph); on("OSQP_HIP_SLOTS", p.slots); on("OSQP_HIP_PCG_FUSED", p.pcg_fused); num("OSQP_HIP_F1", p.f1); on("OSQP_HIP_WINDOW", p.window);
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
  p->polish_delta_floor = 1e-3; p->polish_pcg_tol = 1e-15; p->woodbury = 1; p->woodbury_direct = 1; p->woodbury_large = 1; p->woodbury_cache = 1;
  p->slot_poll = 1; p->poll_low = 6; p->poll_first = 0.8; p->poll_frac = 0.75; p->poll_wait = 0.7;
  p->finish_pairs = 12; p->poll_sleep_us = 30;
  p->reorder = 1; p->kform = 0; p->woodbury_dual = 1; p->woodbury_vendor = 0; p->woodbury_fused = 1; p->woodbury_direct_tol = 1e-6; p->debug_fail_refactor = 0; p->batch_wave = 0;
}
void Engine::set_default_policy(const OSQPHipPolicy *p) {
  g_default_policy_set = p != nullptr;
  if (p) g_default_policy = *p;
}
int Engine::get_policy(OSQPHipPolicy *p) const { if (!p) return OSQP_DATA_VALIDATION_ERROR; *p = pol_; return OSQP_NO_ERROR; }
int Engine::set_policy(const OSQPHipPolicy *p) {
  if (!p) return OSQP_DATA_VALIDATION_ERROR;
  if (!(p->extrap >= 0 && p->extrap <= 2) || p->rho_window < 0 || !(p->rho_window_tol > 0) || !(p->rho_tol_exp > 0 && p->rho_tol_exp <= 1) ||
      !(p->budget_sigma >= 0) || p->finish_pairs < 1 || p->batch_variant < 0 || p->batch_variant > 5 || p->batch_wave < -1 || p->batch_wave > 1 ||
      !(p->rho_eq_factor == 0 || p->rho_eq_factor >= 1) || p->reorder < 0 || p->reorder > 2 || !(p->woodbury_direct_tol > 0 && p->woodbury_direct_tol < 1) || !(p->polish_delta_floor > 0) || !(p->polish_pcg_tol > 0 && p->polish_pcg_tol < 1))
    return OSQP_SETTINGS_VALIDATION_ERROR;
  const OSQPHipPolicy old = pol_;
  pol_ = *p; pol_explicit_ = true;
  // [setup] fields keep the value the handle was built with
  pol_.slots = old.slots; pol_.pcg_fused = old.pcg_fused; pol_.f1 = old.f1; pol_.window = old.window; pol_.woodbury = old.woodbury; pol_.woodbury_direct = old.woodbury_direct; pol_.woodbury_large = old.woodbury_large; pol_.reorder = old.reorder; pol_.woodbury_cache = old.woodbury_cache; pol_.woodbury_fused = old.woodbury_fused; pol_.woodbury_direct_tol = old.woodbury_direct_tol; pol_.kform = old.kform; pol_.woodbury_dual = old.woodbury_dual; pol_.woodbury_vendor = old.woodbury_vendor;
  if (pol_.graph != old.graph) { use_graph_ = pol_.graph != 0; if (dev_ready_) { be::activate(d_); be::sync(d_); drop_graphs(); } }
  if (dev_ready_) d_.theta = pol_.extrap;
  if (dev_ready_ && d_.wb.dbg && pol_.debug_fail_refactor != old.debug_fail_refactor) { be::activate(d_); const int v = pol_.debug_fail_refactor; be::h2d(d_, d_.wb.dbg, &v, sizeof(int)); }
  if (dev_ready_ && pol_.rho_eq_factor >= 1.0 && pol_.rho_eq_factor != old.rho_eq_factor) return set_rho_eq_factor(pol_.rho_eq_factor);
  return OSQP_NO_ERROR;
}

// ---- verbose output.  The reference prints through c_print = PySys_WriteStdout under the GIL (/root/reference/cmake/printing.h:2-7); here every
// piece of text goes to the handle's print function (include/osqp_hip.h osqp_hip_set_print; the Python layer installs one that writes to
// sys.stdout), or to stdout when there is none.
namespace {
std::atomic<osqp_hip_print_fn> g_print_fn{nullptr};
std::atomic<void *> g_print_user{nullptr};
}
void Engine::set_default_print(osqp_hip_print_fn fn, void *user) { g_print_user.store(user); g_print_fn.store(fn); }
void Engine::say(const char *fmt, ...) const {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (print_fn_) print_fn_(buf, print_user_);
  else { std::fputs(buf, stdout); std::fflush(stdout); }
}
// one line of the iteration table (_osqp.py:960-978; the time column is the run time so far, :964-967)
void Engine::print_summary_line(int iter, double obj, double pri, double dua, double rho, double t0) const {
  say("%4d  %11.4e   %8.2e   %8.2e   %8.2e  %8.2es\n", iter, obj, pri, dua, rho, (first_run_ ? info.setup_time : info.update_time) + (now_s() - t0));
}
// the checks the state block has logged since the last call (policy.h Ctl::log: every kCtlPrintInterval iterations, _osqp.py:1230-1231)
void Engine::print_log(const Ctl &c, double t0) {
  if (!settings.verbose) return;
  if (c.nlog - log_printed_ > kCtlLog) log_printed_ = c.nlog - kCtlLog;      // (the host fell a whole ring behind: the oldest lines are gone)
  for (; log_printed_ < c.nlog; log_printed_++) {
    const double *e = c.log[log_printed_ % kCtlLog];
    print_summary_line((int)e[0], e[1], e[2], e[3], e[4], t0);
  }
}

Engine::Engine() {
  print_fn_ = g_print_fn.load(); print_user_ = g_print_user.load();
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
  // (... and the one POINTER of Dev that moves during a handle's life: the inverse the device-factorised Woodbury form applies, backend.h DevWb::cache_buf)
  const double sig[7] = {d_.theta, d_.alpha, d_.sigma, d_.rho_eq_factor, d_.rho_eq_mixed, (double)d_.eq_from_cnt, (double)reinterpret_cast<uintptr_t>(d_.wb.Sinv)};
  if (std::memcmp(sig, graph_sig_, sizeof(sig)) == 0) return;
  if (!graphs_.empty() || !sgraphs_.empty()) { be::sync(d_); drop_graphs(); }
  std::memcpy(graph_sig_, sig, sizeof(sig));
}

void Engine::free_all() {
  if (!dev_ready_) return;
  try { be::activate(d_); be::ext_wait(d_); be::sync(d_); } catch (const DeviceError &) {}       // runs in the destructor: release what we can
  drop_graphs();
  if (bbuf_) { be::dfree(d_, bbuf_); bbuf_ = nullptr; bbuf_cap_ = 0; }
  if (bmat_) { be::dfree(d_, bmat_); bmat_ = nullptr; bmat_cap_ = 0; }
  if (d_batch_order_) { be::dfree(d_, d_batch_order_); d_batch_order_ = nullptr; batch_order_cap_ = 0; }
  if (d_batch_iters_) { be::dfree(d_, d_batch_iters_); d_batch_iters_ = nullptr; d_batch_iters_n_ = 0; }
  batch_order_.clear();
  if (ckpt_) { be::dfree(d_, ckpt_); ckpt_ = nullptr; }
  free_batch_direct(); free_batch_spectral(); free_batch_wave();
  if (d_.f1.va) d_.Minv = d_.xs = d_.p = d_.r = d_.s = nullptr;      // (these point into the F1 arena, freed as one block below)
  if (d_.wb.cache_buf[0]) d_.wb.Sinv = d_.wb.cache_buf[0];      // (Sinv may point at one of the cached inverses: buffer 0 is this list's, the others are the backend's)
  void *ptrs[] = {d_.A.rowptr, d_.A.col, d_.A.blkdesc, d_.A.val, d_.B.rowptr, d_.B.col, d_.B.blkdesc, d_.B.val, d_.Bdiag, d_.A.runinfo, d_.B.runinfo, d_.A.blkwin, d_.B.blkwin, d_.A.lcol, d_.B.lcol, d_.qraw, d_.lraw, d_.uraw, d_.cnt,
                  d_.q, d_.l, d_.u, d_.D, d_.Dinv, d_.E, d_.Einv, d_.rho, d_.rho_inv, d_.ctype, d_.x, d_.z, d_.y, d_.dx,
                  d_.dy, d_.xs, d_.xg, d_.xsp, d_.ztg, d_.zt, d_.t0, d_.v, d_.r, d_.uu, d_.p, d_.s, d_.w, d_.t, d_.Minv, d_.uu2, d_.ms, d_.part, d_.res,
                  d_.scal, d_.flags, d_.slot, d_.Praw, d_.Araw, d_.cs, d_.Pi, d_.Pj, d_.Pm1, d_.Pm2, d_.Ai, d_.Aj, d_.AmA, d_.AmB,
                  d_.wb.AL.rowptr, d_.wb.AL.col, d_.wb.AL.blkdesc, d_.wb.AL.runinfo, d_.wb.AL.val, d_.wb.ALT.rowptr, d_.wb.ALT.col, d_.wb.ALT.blkdesc, d_.wb.ALT.runinfo, d_.wb.ALT.val,
                  d_.wb.al_src, d_.wb.alt_src, d_.wb.islong, d_.wb.rows, d_.wb.WT, d_.wb.S, d_.wb.Sinv, d_.wb.g, d_.wb.h, d_.wb.Dinv0, d_.wb.colmap, d_.wb.W, d_.wb.pv, d_.wb.info, d_.wb.dbg, d_.wb.x.tile, d_.wb.x.tile2, d_.wb.x.partG, d_.wb.x.partZ, d_.wb.x.ls0, d_.wb.x.ls1, d_.wb.x.lz0, d_.wb.x.lz1, d_.wb.x.sinvp, d_.wb.x.sc_ptr, d_.wb.x.sc_row, d_.wb.x.sc_src, d_.wb.x.sc_val, d_.wb.x.bjj,
                  d_.ctl, d_.f1.blk, d_.f1.stream, d_.f1.cptr, d_.f1.prp, d_.f1.pcol, d_.f1.psrc, d_.f1.pval, d_.f1.va, d_.f1.fcol, d_.f1.fq, d_.f1.sp_ptr, d_.f1.spk, d_.f1.spill, d_pc_, d_pr_,
                  d_.wb.gjwork, d_.wb.cc, d_.wb.sig, d_.wb.lidx, d_.wb.Ad, d_.wb.ud, d_.wb.ccd, d_.wb.gp, d_.wb.bq_ptr, d_.wb.bq_idx, d_.wb.bq_col, d_.wb.Bd.blkdesc, d_.wb.Bn.blkdesc, d_.wb.As.blkdesc, d_.wb.kind, d_.wb.dcol, d_.wb.srow, d_.wb.ssrc, d_.wb.sval, d_.wb.sg_ptr, d_.wb.sg_col, d_.wb.wv, d_.wb.den, d_.wb.beta, d_.wb.wbeta, d_.wb.rt, d_.wb.uz,
                  d_.kf.K.rowptr, d_.kf.K.col, d_.kf.K.blkdesc, d_.kf.K.runinfo, d_.kf.K.val, d_.kf.tptr, d_.kf.trow, d_.kf.ta, d_.kf.tb, d_.kf.rec};
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
  const std::vector<int> before = ctype_;
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
  if (before != ctype_) for (int k = 0; k < DevWb::kCache; k++) d_.wb.cache_rho[k] = -1.0;      // other classes, another rho vector under the same rho_bar: cached Woodbury inverses are dead
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
      if (be::wbf_active(d_)) { be::wbf_iteration(d_); continue; }      // column-space direct mode, fused: seven launches, the dense block streamed twice (backend.h DevWb::fused)
      be::kb_rhs(d_);
      if (wb) be::wb_apply(d_, 0, d_.wb.exact);
      if (wb && d_.wb.exact) { be::ka(d_, budget); continue; }      // M^-1 r_0 is the solve (x~ formed by the last kernel of M^-1)
      for (int i = 0; i < budget; i++) { be::k1(d_, i); be::k2(d_, i); if (!fused || i == budget - 1) { be::kv(d_, i); if (wb) be::wb_apply(d_, (i + 1) & 1); } }
      be::ka(d_, budget);
    }
  };
  stats_.kernel_launches += xy ? ((d_.wb.x.one && !d_.wb.x.slots) ? niter + 1.0 : 2.0 * niter + 1) : (double)niter * ((wb && d_.wb.exact) ? (d_.wb.dual ? 7 : 5) : (fused ? 3 + 2 * budget : 2 + 3 * budget + (wb ? 3 * (budget + 1) : 0)));
  if (!(use_graph_ && be::graphs_supported())) { enqueue(niter); return; }
  // one executable graph per (ADMM iterations, PCG budget); graphs are kept below kMaxGraphNodes kernel nodes (a
  // check_termination = 0 solve would otherwise capture max_iter * (2 + 3*budget) nodes in one graph)
  constexpr int kMaxGraphNodes = 8192;
  const int per = std::max(1, kMaxGraphNodes / (2 + 3 * budget + (wb ? 3 * (budget + 1) : 0)));
  for (int left = niter; left > 0;) {
    const int cnt = std::min(left, per);
    auto key = std::make_pair(cnt, budget | ((wb && d_.wb.exact) ? (1 << 24) : 0) | (xy ? (1 << 25) : 0) | (be::wbf_active(d_) ? (1 << 26) : 0));      // (the direct mode is another launch sequence: it may come and go with rho in the large-rank form)
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

// osqp_solve: the solve proper between a hipEvent pair on the solver's stream ("hipEvent around solve"; OSQPHipStats::gpu_solve_ms)
int Engine::solve() {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  be::activate(d_);
  be::ev_mark(d_, 0);
  const int err = solve_impl();
  be::ev_mark(d_, 1);
  stats_.gpu_solve_ms = be::ev_ms(d_);
  return err;
}

int Engine::solve_impl() {
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
  stats_.woodbury_factorisations = 0; stats_.woodbury_factor_ms = 0; stats_.woodbury_cache_hits = 0;
  double res[R_COUNT];
  admm_core(t0, res);
  info.rho_estimate = rho_estimate(res);                                                 // :1275
  info.solve_time = now_s() - t0;
  if (settings.polishing && info.status_val == OSQP_SOLVED) polish();                   // :1278-1279
  store_solution();
  be::sync(d_);
  info.run_time = (first_run_ ? info.setup_time : info.update_time) + info.solve_time + info.polish_time;   // :1284-1289
  first_run_ = false; clear_update_time_ = true;
  if (settings.verbose) print_footer();
  return OSQP_NO_ERROR;
}

// _osqp.py:1079-1096
void Engine::print_footer() const {
  say("\nstatus:               %s\n", info.status);
  if (settings.polishing && info.status_val == OSQP_SOLVED) {
    if (info.status_polish == 1) say("solution polish:      successful\n");
    else if (info.status_polish == -1) say("solution polish:      unsuccessful\n");
  }
  say("number of iterations: %d\n", info.iter);
  if (info.status_val == OSQP_SOLVED || info.status_val == OSQP_SOLVED_INACCURATE) {
    say("optimal objective:    %.4f\n", info.obj_val);
    say("run time:             %.2es\n", info.run_time);
  }
  say("optimal rho estimate: %.2e\n\n", info.rho_estimate);
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
  // (a rho_bar this handle has factorised for before is a look-up + a numerical check, not a factorisation: backend.h DevWb::cache_buf)
  struct Tally { Engine *e; double t0; int hits0; ~Tally() { if (t0 > 0) { if (e->d_.wb.cache_hits > hits0) e->stats_.woodbury_cache_hits += 1; else e->stats_.woodbury_factorisations += 1; e->stats_.woodbury_factor_ms += 1e3 * (now_s() - t0); } } } tally{this, tf, d_.wb.cache_hits};
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
  rho_at_last_check_ = rho_bar_;
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
  sync_graph_scalars();                                // (captured strings freeze Dev's scalars: also checked here, where strings are replayed outside admm_core -- polish, ls_solve)
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
  // Feeding from HISTORY: the previous solve of this handle recorded how many slot launches each of its chunks consumed (Ctl::hist).  While this
  // solve follows the same course -- a parametric re-solve, the steps of a benchmark -- every chunk gets exactly that string with its boundary
  // group right behind it, two chunks ahead of the device: no idle launches, and nothing depends on when the host gets to poll.  The first chunk
  // that outruns its string ends the mode (the polled top-ups below take over from whatever is queued); a chunk that needs less only idles.
  bool use_hist = pol_.slot_poll != 0 && !feed_hist_.empty();
  int fed = snap.boundaries;                         // chunks [.., fed) have their strings and groups enqueued
  long fed_end_seq = 0;                              // slot launches enqueued up to the end of chunk fed - 1
  auto feed_from_history = [&](int upto) {
    while (use_hist && fed < upto) {
      if (fed >= (int)feed_hist_.size()) break;      // every recorded chunk is queued: the solve ends there if it follows the recorded course (the poll loop below waits for that,
                                                     //  or for the device to begin a chunk beyond the record) -- topping up here would queue launches and groups behind the last boundary
      if (feed_hist_[fed] <= 0) { use_hist = false; break; }
      const int np = (feed_hist_[fed] + 1) / 2;
      run_slots(0, np, 0); launched += np;
      run_group(diagonal);
      fed += 1; fed_end_seq = 2 * launched;
    }
  };
  if (pol_.slot_log) { std::fprintf(stderr, "feed: history of %zu chunks:", feed_hist_.size()); for (int h : feed_hist_) std::fprintf(stderr, " %d", h); std::fprintf(stderr, "\n"); }
  feed_from_history(snap.boundaries + 2);
  // chunk in flight as of the last poll, and the progress counters at the first poll that saw it (rate estimate)
  if (!use_hist && launched == 0) {
    const int cnt = snap.ch_next - snap.iter;
    const double full = pairs_for(cnt, pred_for(snap, snap.ch_kind, snap.ch_tight));
    const int np = one_string(std::max(2, (int)std::floor(pol_.poll_first * full)), (int)std::floor(0.95 * full));
    run_slots(0, np, 0); launched += np; }
  for (;;) {
    be::ctl_poll(d_, &snap, &seq, &done);
    print_log(snap, t0);
    if (snap.status != CTL_RUNNING) break;
    if (now_s() - t0 > settings.time_limit) { timed_out = true; break; }
    if (use_hist) {
      // (the device is in chunk snap.boundaries; all of it and of the next is queued.  Everything up to the end of chunk fed - 1 consumed and
      //  the device still inside it: the course differs from the recorded one)
      if ((snap.boundaries < fed && seq >= fed_end_seq && !snap.chunk_done) || (fed >= (int)feed_hist_.size() && snap.boundaries >= fed)) {      // (chunk_done: the chunk HAS finished with its string, only its boundary group has not run yet)      // (... or past the last recorded chunk and still running)
        use_hist = false;
        if (pol_.slot_log) std::fprintf(stderr, "feed: history mode ends: device in chunk %d (iter %d, %d of its iterations done), chunks fed %d, launches executed %d of %ld fed\n", snap.boundaries, snap.iter, done, fed, seq, fed_end_seq);
      }
      else {
        feed_from_history(snap.boundaries + 2);
        if (use_hist) { std::this_thread::sleep_for(std::chrono::microseconds(std::max(pol_.poll_sleep_us, 100))); continue; }
      }
    }
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
  feed_hist_.assign(c.hist, c.hist + std::min(std::max(c.boundaries, 0), (int)kCtlHist));      // (what the next solve of this handle is fed from)
  if (pol_.slot_log) { std::fprintf(stderr, "feed: this solve's chunks consumed:"); for (int h : feed_hist_) std::fprintf(stderr, " %d", h); std::fprintf(stderr, "  (boundaries %d, launches enqueued %ld)\n", c.boundaries, 2 * launched); }
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
  log_printed_ = 0;
  if (settings.verbose) say("iter   objective    pri res    dua res    rho       time\n");      // _osqp.py:609-613
  int flags[F_COUNT] = {0};
  for (;;) {
    int st;
    // (evaluated at every boundary: the Woodbury direct mode may leave the slot form in the middle of a solve -- NEED_REFACTOR below)
    // (the Woodbury direct mode takes the slot form only when asked to, device_driven = 2: its two launches per ADMM iteration are so short
    //  that a boundary group of sixteen launches costs what the host round trip it replaces costs, and the fixed captured strings of the
    //  host-synchronous loop carry no idle launches -- 34 against 41 ms on the portfolio QP)
    d_.wb.x.slots = (pol_.device_driven >= 2) ? 1 : 0;
    const bool slots = use_slots_ && be::slots_supported(d_);
    // (`verbose` changes nothing about how a solve runs: the printed lines come from the state block's log, policy.h Ctl::log)
    const bool device_driven = slots && pol_.device_driven && be::ctl_supported(d_);
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
        print_log(c, t0);
      }
      (void)tight; (void)rho_was;
      if (st == CTL_RUNNING && c.rho_flag) apply_rho(c.rho_bar);
    } else {
      st = run_device_driven(t0, res, flags);
      info_from_ctl(t0);
      print_log(c, t0);
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
  if (settings.verbose && info.iter % kCtlPrintInterval != 0) print_summary_line(info.iter, info.obj_val, info.prim_res, info.dual_res, rho_at_last_check_, t0);      // _osqp.py:1259-1261
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
  if (settings.verbose) say("plsh  %11.4e   %8.2e   %8.2e   --------  %8.2es\n", pol_obj, pol_pri, pol_dua,
                            (first_run_ ? info.setup_time : info.update_time) + info.solve_time + info.polish_time);      // _osqp.py:980-996
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
    // (in place: the buffers solution.x / y / *_inf_cert point to never move during a handle's life)
    std::vector<double> o;
    auto back = [&o](std::vector<double> &v, const std::vector<int> &perm) { o.resize(v.size()); for (size_t k = 0; k < v.size(); k++) o[perm[k]] = v[k]; std::copy(o.begin(), o.end(), v.begin()); };
    back(sol_x_, pc_); back(sol_dc_, pc_); back(sol_y_, pr_); back(sol_pc_, pr_);
  }
}


}  // namespace osqp_hip
