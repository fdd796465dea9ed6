// engine_api.cpp -- the rest of the C-API surface of the engine: cold / warm start, data and settings updates, the LinSysSolver slot, the batch
// and small-problem paths (one-workgroup-per-QP kernels), statistics and probes.  See engine.hpp.
#include "engine_internal.hpp"

namespace osqp_hip {

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
  mat_epoch_ += 1;
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
  d_.fused = 0; d_.f1.on = 0; d_.kf.on = 0; d_.wb.on = 0;
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

// ------------------------------------------------------------------------------------------------ batch path, one wave per problem
// engine.hpp BatchWave / backend.h BatchParams::wv_*.  Rows of A sorted by length (descending, stable): sorted position p -> lane p % 64, slot p / 64; a group
// of 64 positions takes as many ELL steps as its longest (= first) row has entries.  A' (B's entries with column >= n) in the natural order of the variables.
constexpr int kBatchWaveMin = 2048, kBatchWaveMinOrdered = 1280;      // smallest batches that take the wave-per-problem kernel by default (without / with a launch order)
void Engine::free_batch_wave() {
  void *ptrs[] = {bwv_.Aidx, bwv_.Acol, bwv_.Tidx, bwv_.Tcol, bwv_.row, bwv_.queue};
  for (void *p : ptrs) if (p) be::dfree(d_, p);
  bwv_ = BatchWave();
}
void Engine::prepare_batch_wave() {
  if (bwv_.tried) return;
  bwv_.tried = true;
  if (n < 1 || n > kBatchSpecN || m < 1 || m > 256 || reordered_) return;
  std::vector<int> pos(m);
  for (int i = 0; i < m; i++) pos[i] = i;
  std::stable_sort(pos.begin(), pos.end(), [&](int a, int b) { return Arp_[a + 1] - Arp_[a] > Arp_[b + 1] - Arp_[b]; });
  int aend[4] = {0, 0, 0, 0}, tend[2] = {0, 0}, acc = 0;
  for (int g = 0; g < 4; g++) { if (g * 64 < m) { const int r = pos[g * 64]; acc += Arp_[r + 1] - Arp_[r]; } aend[g] = acc; }
  if (acc > kBatchWaveSA) return;
  std::vector<int> Aidx((size_t)std::max(acc, 1) * 64, -1), Acol((size_t)std::max(acc, 1) * 64, 0), row(256, -1);
  for (int p = 0; p < m; p++) {
    const int r = pos[p], lane = p % 64, g = p / 64, s0 = g ? aend[g - 1] : 0;
    row[p] = r;
    for (int k = Arp_[r]; k < Arp_[r + 1]; k++) { Aidx[(size_t)(s0 + k - Arp_[r]) * 64 + lane] = k; Acol[(size_t)(s0 + k - Arp_[r]) * 64 + lane] = Arj_[k]; }
  }
  int tacc = 0;
  std::vector<int> tcnt(n, 0);
  for (int j = 0; j < n; j++) for (int k = Brp_[j]; k < Brp_[j + 1]; k++) tcnt[j] += Bj_[k] >= n;
  for (int g = 0; g < 2; g++) { int mx = 0; for (int j = g * 64; j < std::min(n, g * 64 + 64); j++) mx = std::max(mx, tcnt[j]); tacc += mx; tend[g] = tacc; }
  if (tacc > kBatchWaveST) return;
  std::vector<int> Tidx((size_t)std::max(tacc, 1) * 64, -1), Tcol((size_t)std::max(tacc, 1) * 64, 0);
  for (int j = 0; j < n; j++) {
    const int lane = j % 64, g = j / 64, s0 = g ? tend[g - 1] : 0;
    int e = 0;
    for (int k = Brp_[j]; k < Brp_[j + 1]; k++) if (Bj_[k] >= n) { Tidx[(size_t)(s0 + e) * 64 + lane] = k; Tcol[(size_t)(s0 + e) * 64 + lane] = Bj_[k] - n; e++; }
  }
  if (!be::batch_wave_lds_bytes(n, m, acc + tacc)) return;
  auto up = [&](const std::vector<int> &v) { int *dptr = dev_vec<int>(d_, v.size()); be::h2d(d_, dptr, v.data(), sizeof(int) * v.size()); return dptr; };
  bwv_.Aidx = up(Aidx); bwv_.Acol = up(Acol); bwv_.Tidx = up(Tidx); bwv_.Tcol = up(Tcol); bwv_.row = up(row);
  bwv_.queue = dev_vec<int>(d_, 1);
  for (int g = 0; g < 4; g++) bwv_.aend[g] = aend[g];
  bwv_.tend[0] = tend[0]; bwv_.tend[1] = tend[1];
  bwv_.ok = true;
}
// (behind attach_batch_direct: the form needs the spectral decomposition)
void Engine::attach_batch_wave(BatchParams &p, int nbatch) {
  // OSQPHipPolicy::batch_wave; automatic = where it measured faster (tools/batch_size_sweep.py): large batches, smaller ones only with a launch order
  if (!p.sp_V || pol_.batch_wave < 0 || p.mat_on || p.polish) return;
  if (pol_.batch_wave == 0 && nbatch < (p.order ? kBatchWaveMinOrdered : kBatchWaveMin)) return;
  prepare_batch_wave();
  if (!bwv_.ok) return;
  p.wv_on = 1;
  for (int g = 0; g < 4; g++) p.wv_aend[g] = bwv_.aend[g];
  p.wv_tend[0] = bwv_.tend[0]; p.wv_tend[1] = bwv_.tend[1];
  // the longest-expected problems to the workgroup kernel (be::batch_solve applies it when there is a launch order): wv_cus CUs are left free for it
  // and take wv_split problems, several each, one after the other
  static const int env_split = std::getenv("OSQP_HIP_WAVE_SPLIT") ? std::atoi(std::getenv("OSQP_HIP_WAVE_SPLIT")) : -1;
  static const int env_cus = std::getenv("OSQP_HIP_WAVE_CUS") ? std::atoi(std::getenv("OSQP_HIP_WAVE_CUS")) : -1;
  // (tools/batch_size_sweep.py under OSQP_HIP_WAVE_SPLIT / OSQP_HIP_WAVE_CUS, MPC batch: 2048 QPs 2.21 ms with 32 / 32, 2.84 with 64 / 16; 4096 QPs 3.26 ms
  //  with 48 / 16, 3.41 with 32 / 32; 8192 QPs 5.66 against 6.18 -- a large batch is bound by the wave kernel's throughput and wants its CUs back)
  p.wv_cus = env_cus >= 0 ? env_cus : (nbatch < 3072 ? 32 : 16);
  p.wv_split = env_split >= 0 ? env_split : (nbatch < 3072 ? 32 : 48);
  p.wv_Aidx = bwv_.Aidx; p.wv_Acol = bwv_.Acol; p.wv_Tidx = bwv_.Tidx; p.wv_Tcol = bwv_.Tcol; p.wv_row = bwv_.row; p.wv_queue = bwv_.queue;
}

// ------------------------------------------------------------------------------------------------ batch path, spectral form of the direct solve
// engine.hpp BatchSpectral.  Dense work on the host, n <= kBatchSpecN: the SCALED matrices (c D P D + sigma I, E A D: what the kernels hold) are
// rebuilt from the host copies, K_ref and M1 assembled, K_ref = L L' (Cholesky), C = L^-1 M1 L^-T, C = Q Lambda Q' (cyclic Jacobi sweeps: C is
// symmetric PSD with eigenvalues in [0, 1 / rho_ref]), V = L^-T Q.  Checked before use: || V' K_ref V - I ||_max and || V' M1 V - Lambda ||_max.
constexpr int kBatchSpectralMin = 32;      // smallest batch that has the spectral form prepared for it (Engine::attach_batch_direct)
void Engine::free_batch_spectral() {
  void *ptrs[] = {bs_.V, bs_.lam, bs_.d_ctype, bs_.K0};
  for (void *p : ptrs) if (p) be::dfree(d_, p);
  bs_ = BatchSpectral();
}
bool Engine::prepare_batch_spectral(double rho_ref, double eqf, bool allow_build) {
  const int N = kBatchSpecN;
  if (n > N || m == 0 || !be::device_assembly() || reordered_) { bs_.ok = false; return false; }
  // the reference classes: those of the solver's own bounds, classified as the kernel classifies a problem's (batch_hip.hip, _osqp.py:505-518)
  std::vector<int> ct(m);
  for (int i = 0; i < m; i++) {
    const double li = E_[i] * std::max(l0_[i], -OSQP_INFTY), ui = E_[i] * std::min(u0_[i], OSQP_INFTY);
    int ty = (li < -OSQP_INFTY * 1e-4 && ui > OSQP_INFTY * 1e-4) ? -1 : ((ui - li < 1e-4) ? 1 : 0);
    if (!settings.rho_is_vec) ty = 0;
    ct[i] = ty;
  }
  if ((bs_.ok || bs_.failed) && bs_.mat_epoch == mat_epoch_ && bs_.eqf == eqf && bs_.sigma == settings.sigma && bs_.rho_is_vec == settings.rho_is_vec && bs_.ctype == ct) return bs_.ok;
  if (!allow_build) return false;                          // (what exists was built for other matrices / classes; this caller does not pay for a new one)
  const double rref = bs_.ok ? bs_.rho_ref : rho_ref;       // (any reference works: K(rho) = K_ref + (rho - rho_ref) M1; the first call's rho stays)
  free_batch_spectral();
  // a rejected attempt (pivot <= 0, sweeps not converged, check missed) is remembered under the same key: every later call would redo O(n^3 x sweeps)
  // host work for the same verdict.  Cleared by whatever changes the key (matrix update, other classes, sigma).
  auto reject = [&]() { bs_.ok = false; bs_.failed = true; bs_.ctype = ct; bs_.eqf = eqf; bs_.sigma = settings.sigma; bs_.rho_is_vec = settings.rho_is_vec; bs_.mat_epoch = mat_epoch_; };
  std::vector<double> K((size_t)n * n, 0.0), M1((size_t)n * n, 0.0);
  for (int j = 0; j < n; j++) {
    for (int k = P_.p[j]; k < P_.p[j + 1]; k++) {
      const int i = P_.i[k];
      const double v = c_ * D_[i] * P_.x[k] * D_[j];
      K[(size_t)i * n + j] += v; if (i != j) K[(size_t)j * n + i] += v;
    }
    K[(size_t)j * n + j] += settings.sigma;
  }
  { // A' W A by rows of A (CSC -> per-row lists)
    std::vector<std::vector<std::pair<int, double>>> rows(m);
    for (int j = 0; j < n; j++) for (int k = A_.p[j]; k < A_.p[j + 1]; k++) rows[A_.i[k]].push_back({j, E_[A_.i[k]] * A_.x[k] * D_[j]});
    for (int i = 0; i < m; i++) {
      const int ty = ct[i];
      for (auto &a : rows[i]) for (auto &b : rows[i]) {
        const double v = a.second * b.second;
        if (ty == -1) K[(size_t)a.first * n + b.first] += 1e-6 * v;                 // loose rows keep rho_i = 1e-6 whatever rho_bar is (_osqp.py:520)
        else M1[(size_t)a.first * n + b.first] += (ty == 1 ? eqf : 1.0) * v;
      }
    }
  }
  std::vector<double> Kref(K);
  for (size_t e = 0; e < Kref.size(); e++) Kref[e] += rref * M1[e];
  // Cholesky Kref = L L' (lower, in place in Lm)
  std::vector<double> Lm(Kref);
  for (int c = 0; c < n; c++) {
    double dg = Lm[(size_t)c * n + c];
    for (int k = 0; k < c; k++) dg -= Lm[(size_t)c * n + k] * Lm[(size_t)c * n + k];
    if (!(dg > 0.0)) { reject(); return false; }
    dg = std::sqrt(dg); Lm[(size_t)c * n + c] = dg;
    for (int r = c + 1; r < n; r++) {
      double v = Lm[(size_t)r * n + c];
      for (int k = 0; k < c; k++) v -= Lm[(size_t)r * n + k] * Lm[(size_t)c * n + k];
      Lm[(size_t)r * n + c] = v / dg;
    }
  }
  auto fwd = [&](std::vector<double> &X) {                    // X <- L^-1 X  (X: n x n row-major, column by column)
    for (int col = 0; col < n; col++) for (int r = 0; r < n; r++) {
      double v = X[(size_t)r * n + col];
      for (int k = 0; k < r; k++) v -= Lm[(size_t)r * n + k] * X[(size_t)k * n + col];
      X[(size_t)r * n + col] = v / Lm[(size_t)r * n + r];
    }
  };
  auto transpose = [&](std::vector<double> &X) { for (int a = 0; a < n; a++) for (int b = a + 1; b < n; b++) std::swap(X[(size_t)a * n + b], X[(size_t)b * n + a]); };
  std::vector<double> C(M1);
  fwd(C); transpose(C); fwd(C);                              // L^-1 M1 L^-T  (symmetric)
  for (int a = 0; a < n; a++) for (int b = a + 1; b < n; b++) { const double v = 0.5 * (C[(size_t)a * n + b] + C[(size_t)b * n + a]); C[(size_t)a * n + b] = C[(size_t)b * n + a] = v; }
  // cyclic Jacobi: C = Q diag(lam) Q'
  std::vector<double> Q((size_t)n * n, 0.0);
  for (int a = 0; a < n; a++) Q[(size_t)a * n + a] = 1.0;
  bool jacobi_converged = false;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0.0, dsum = 0.0;
    for (int a = 0; a < n; a++) { dsum += C[(size_t)a * n + a] * C[(size_t)a * n + a]; for (int b = a + 1; b < n; b++) off += C[(size_t)a * n + b] * C[(size_t)a * n + b]; }
    if (off <= 1e-32 * (dsum + 1e-300)) { jacobi_converged = true; break; }
    for (int p_ = 0; p_ < n - 1; p_++) for (int q_ = p_ + 1; q_ < n; q_++) {
      const double apq = C[(size_t)p_ * n + q_];
      if (apq == 0.0) continue;
      const double theta = (C[(size_t)q_ * n + q_] - C[(size_t)p_ * n + p_]) / (2.0 * apq);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0)), cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
      for (int k = 0; k < n; k++) { const double ckp = C[(size_t)k * n + p_], ckq = C[(size_t)k * n + q_]; C[(size_t)k * n + p_] = cs * ckp - sn * ckq; C[(size_t)k * n + q_] = sn * ckp + cs * ckq; }
      for (int k = 0; k < n; k++) { const double cpk = C[(size_t)p_ * n + k], cqk = C[(size_t)q_ * n + k]; C[(size_t)p_ * n + k] = cs * cpk - sn * cqk; C[(size_t)q_ * n + k] = sn * cpk + cs * cqk; }
      for (int k = 0; k < n; k++) { const double qkp = Q[(size_t)k * n + p_], qkq = Q[(size_t)k * n + q_]; Q[(size_t)k * n + p_] = cs * qkp - sn * qkq; Q[(size_t)k * n + q_] = sn * qkp + cs * qkq; }
    }
  }
  // V = L^-T Q  (back substitution per column)
  std::vector<double> V(Q);
  for (int col = 0; col < n; col++) for (int r = n - 1; r >= 0; r--) {
    double v = V[(size_t)r * n + col];
    for (int k = r + 1; k < n; k++) v -= Lm[(size_t)k * n + r] * V[(size_t)k * n + col];
    V[(size_t)r * n + col] = v / Lm[(size_t)r * n + r];
  }
  std::vector<double> lam(n);
  for (int k = 0; k < n; k++) lam[k] = std::max(C[(size_t)k * n + k], 0.0);
  // check, EVERY entry (two n^3 products: cheap next to the sweeps): V' Kref V = I, V' M1 V = Lambda -- a V that is wrong anywhere would serve as K^-1 for a whole batch
  double err = jacobi_converged ? 0.0 : 1.0;
  { std::vector<double> T((size_t)n * n);
    for (int pass = 0; pass < 2 && err < 1e-9; pass++) {
      const std::vector<double> &X = pass ? M1 : Kref;
      for (int i = 0; i < n; i++) for (int b = 0; b < n; b++) { double t = 0; for (int j = 0; j < n; j++) t += X[(size_t)i * n + j] * V[(size_t)j * n + b]; T[(size_t)i * n + b] = t; }      // T = X V
      for (int a = 0; a < n; a++) for (int b = 0; b < n; b++) {
        double s_ = 0; for (int i = 0; i < n; i++) s_ += V[(size_t)i * n + a] * T[(size_t)i * n + b];
        err = std::max(err, pass ? std::fabs(s_ - (a == b ? lam[a] : 0.0)) / (1.0 + std::max(lam[a], lam[b])) : std::fabs(s_ - (a == b ? 1.0 : 0.0)));
      }
    } }
  if (!(err < 1e-9)) { reject(); return false; }
  std::vector<double> Vp((size_t)N * N, 0.0), lp(N, 0.0);
  for (int k = 0; k < n; k++) { lp[k] = lam[k]; for (int j = 0; j < n; j++) Vp[(size_t)k * N + j] = V[(size_t)j * n + k]; }      // column-major, zero-padded
  bs_.V = dev_vec<double>(d_, Vp.size()); be::h2d(d_, bs_.V, Vp.data(), sizeof(double) * Vp.size());
  bs_.lam = dev_vec<double>(d_, N); be::h2d(d_, bs_.lam, lp.data(), sizeof(double) * N);
  bs_.d_ctype = dev_vec<int>(d_, m); be::h2d(d_, bs_.d_ctype, ct.data(), sizeof(int) * m);
  bs_.Vh = Vp; bs_.lamh = lp; bs_.k0_ok = false;
  bs_.ctype = ct; bs_.rho_ref = rref; bs_.eqf = eqf; bs_.sigma = settings.sigma; bs_.rho_is_vec = settings.rho_is_vec; bs_.mat_epoch = mat_epoch_;
  bs_.ok = true;
  return true;
}

// K^-1(rho0) = V diag(1 / (1 + (rho0 - rho_ref) lambda)) V' in the register layout batch_hip.hip's threads hold it in (the result layout of the f64 matrix
// instruction), formed once for the whole batch: a problem loads its first K^-1 instead of building it.
void Engine::prepare_batch_k0(double rho0) {
  if (!bs_.ok) return;
  if (bs_.k0_ok && bs_.k0_rho == rho0) return;
  const int N = kBatchSpecN, T = 256;
  const double dl = rho0 - bs_.rho_ref;
  std::vector<double> dk(N), K0((size_t)64 * T);
  for (int k = 0; k < N; k++) dk[k] = 1.0 / (1.0 + dl * bs_.lamh[k]);
  const double *V = bs_.Vh.data();                            // V[k * N + j] = V(j, k)
  // (layout: batch_hip.hip kacc -- thread tid = 64 w + l holds, for tile t = tr * 8 + tc and register r, element (32 w + 16 tr + l / 16 + 4 r, 16 tc + l % 16))
  for (int tid = 0; tid < T; tid++) {
    const int wv = tid >> 6, l = tid & 63, lj = l & 15, lk = l >> 4;
    for (int t = 0; t < 16; t++) for (int r = 0; r < 4; r++) {
      const int i = 32 * wv + 16 * (t / 8) + lk + 4 * r, j = 16 * (t % 8) + lj;
      double acc = 0.0;
      for (int k = 0; k < n; k++) acc = std::fma(V[(size_t)k * N + i] * dk[k], V[(size_t)k * N + j], acc);
      K0[(size_t)(t * 4 + r) * T + tid] = acc;
    }
  }
  if (!bs_.K0) bs_.K0 = dev_vec<double>(d_, K0.size());
  be::h2d(d_, bs_.K0, K0.data(), sizeof(double) * K0.size());
  bs_.k0_rho = rho0; bs_.k0_ok = true;
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

void Engine::attach_batch_direct(BatchParams &p, bool spectral) {
  if (!bd_.ok) return;
  p.eq_factor_direct = eq_factor_set_ ? eq_factor_mixed_ : 1e3;
  // the spectral form of the same solve, where it applies (never with polish: that factorises another matrix in the band) -- and where the caller is a
  // batch that pays for the host-side decomposition (dense Cholesky + Jacobi sweeps: tens to hundreds of ms for a kernel that runs about one): a
  // single osqp_solve of a small QP and small batches keep the banded kernel unless a batch has prepared the form for this key before
  if (pol_.batch_variant == 0 && !settings.polishing && n <= kBatchSpecN) {
    if (raw_stale_) ensure_host_vectors();             // (the solver's own bounds define the reference classes)
    int n_ineq = 0;
    for (int i = 0; i < m && settings.rho_is_vec; i++) {
      const double li = E_[i] * std::max(l0_[i], -OSQP_INFTY), ui = E_[i] * std::min(u0_[i], OSQP_INFTY);
      n_ineq += !((li < -OSQP_INFTY * 1e-4 && ui > OSQP_INFTY * 1e-4) || (ui - li < 1e-4));
    }
    if (!settings.rho_is_vec) n_ineq = m;
    if (prepare_batch_spectral(p.rho0, n_ineq == 0 ? 1e3 : p.eq_factor_direct, spectral)) {
      p.sp_V = bs_.V; p.sp_lam = bs_.lam; p.sp_ctype = bs_.d_ctype; p.sp_rho_ref = bs_.rho_ref; p.sp_eqf = bs_.eqf;
      prepare_batch_k0(p.rho0);
      if (bs_.k0_ok) { p.sp_K0 = bs_.K0; p.sp_K0_rho = bs_.k0_rho; }
    }
  }
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
// polish_refine_iter refinement steps: the reference's algorithm, _osqp.py:1710-1828).  Not taken with a time limit or when
// OSQP_HIP_SMALL_DIRECT=0.  `verbose` does not change the path: the whole loop is one launch, so the table holds its last line only.
bool Engine::small_direct_applicable() {
  if (!pol_.small_direct || !be::device_assembly() || settings.time_limit < 1e9 || reordered_) return false;
  if (settings.check_dualgap) return false;            // the one-launch kernel has no duality-gap test: the host-driven loop honours the setting
  if (!be::batch_lds_bytes(n, m)) return false;
  prepare_batch_direct();
  if (!bd_.ok) return false;
  BatchParams p{};
  fill_batch_params(p, 1, 0);
  attach_batch_direct(p, false);
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
  if (settings.verbose) {                          // (one launch ran the whole loop: the table holds its last line, _osqp.py:1259-1261)
    say("iter   objective    pri res    dua res    rho       time\n");
    print_summary_line(info.iter, info.obj_val, info.prim_res, info.dual_res, rho_bar_, t0);
    if (info.status_polish) say("plsh  %11.4e   %8.2e   %8.2e   --------  %8.2es\n", info.obj_val, info.prim_res, info.dual_res, info.run_time);
    print_footer();
  }
  first_run_ = false; clear_update_time_ = true;
  return OSQP_NO_ERROR;
}

// Per-problem matrices (BatchParams::mat_on): one scratch block [Aval | Bval | D | Dinv | E | Einv | c | products], filled by be::batch_prepare on `stream`
// (assembly + the reference's equilibration per problem).  The spectral form (one V for the whole batch) is switched off for such a call.
int Engine::attach_batch_matrices(BatchParams &p, const double *Px_dev, const double *Ax_dev, void *stream) {
  if (!be::device_assembly()) return OSQP_FUNC_NOT_IMPLEMENTED;
  const size_t nb = (size_t)p.nbatch, nzA = (size_t)d_.A.nnz, nzB = (size_t)d_.B.nnz, np = bd_.ok ? (size_t)bd_.nprod : 0;
  const size_t need = nb * (nzA + nzB + 2 * (size_t)n + 2 * (size_t)m + 1 + np);
  if (need > bmat_cap_) { be::ext_wait(d_); if (bmat_) be::dfree(d_, bmat_); bmat_ = dev_vec<double>(d_, need); bmat_cap_ = need; be::sync(d_); }
  p.Aval_b = bmat_; p.Bval_b = p.Aval_b + nb * nzA; p.D_b = p.Bval_b + nb * nzB; p.Dinv_b = p.D_b + nb * n; p.E_b = p.Dinv_b + nb * n; p.Einv_b = p.E_b + nb * m;
  p.c_b = p.Einv_b + nb * m; p.kp_val_b = np ? p.c_b + nb : nullptr; p.nprod = (int)np; p.kp_a = bd_.kp_a; p.kp_b = bd_.kp_b;
  p.mat_on = 1; p.sp_V = nullptr; p.sp_K0 = nullptr;
  return be::batch_prepare(d_, p, Px_dev, Ax_dev, settings.scaling, stream);
}

int Engine::batch_solve(int nbatch, const double *q, const double *l, const double *u, double *x, double *y, double *rec, int warm, double *zs_dev, const double *Px, const double *Ax) {
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
  const size_t NP = Px ? (size_t)nbatch * P_.nnz() : 0, NA = Ax ? (size_t)nbatch * A_.nnz() : 0;
  const size_t need = 2 * N + 3 * M + (size_t)nbatch * kBatchRec + n + 2 * (size_t)m + NP + NA;
  if (need > bbuf_cap_) { if (bbuf_) be::dfree(d_, bbuf_); bbuf_ = dev_vec<double>(d_, need); bbuf_cap_ = need; }
  double *dq = bbuf_, *dl = dq + N, *du = dl + M, *dx = du + M, *dy = dx + N, *drec = dy + M, *dq0 = drec + (size_t)nbatch * kBatchRec, *dl0 = dq0 + n, *du0 = dl0 + m;
  double *dPx = du0 + m, *dAx = dPx + NP;
  if (Px) be::h2d(d_, dPx, Px, sizeof(double) * NP);
  if (Ax) be::h2d(d_, dAx, Ax, sizeof(double) * NA);
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
    attach_batch_direct(p, nbatch >= kBatchSpectralMin && !Px && !Ax);
    if (!Px && !Ax) attach_batch_wave(p, nbatch);
    batch_wave_last_ = p.wv_on ? ((p.order && p.wv_cus > 0 && nbatch >= 8 * p.wv_split) ? p.wv_split : 0) : -1;
  }
  if (Px || Ax) { const int e2 = attach_batch_matrices(p, Px ? dPx : nullptr, Ax ? dAx : nullptr, nullptr); if (e2) return e2; }
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
int Engine::batch_solve_device(int nbatch, const double *q, const double *l, const double *u, double *x, double *y, double *rec, int warm, void *stream, const double *Px, const double *Ax) {
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
    attach_batch_direct(p, nbatch >= kBatchSpectralMin && !Px && !Ax);
    if (!Px && !Ax) attach_batch_wave(p, nbatch);
    batch_wave_last_ = p.wv_on ? ((p.order && p.wv_cus > 0 && nbatch >= 8 * p.wv_split) ? p.wv_split : 0) : -1;
  }
  be::sync(d_);                                   // the uploads and the product refresh ran on the solver's stream
  if (Px || Ax) { const int e2 = attach_batch_matrices(p, Px, Ax, stream); if (e2) return e2; }      // (on the caller's stream, in front of the solve launch)
  const int err = be::batch_solve(d_, p, stream);
  if (!err) be::ext_record(d_, stream);           // later calls that overwrite or free what this kernel reads wait for it (ext_wait)
  return err;
}

int Engine::get_stats(OSQPHipStats *out) {
  if (!out) return OSQP_DATA_VALIDATION_ERROR;
  *out = stats_; out->pcg_fused = (d_.f1.on && use_slots_) ? 2.0 : ((d_.kf.on && use_slots_) ? 3.0 : (be::pcg_fused(d_) ? 1.0 : 0.0)); out->batch_direct_bw = bd_.bw_symbolic;
  out->f1_replicas = d_.f1.on ? d_.f1.D : 0;
  out->kform_nnz = d_.kf.on ? (double)d_.kf.K.nnz : 0.0;
  out->batch_wave_split = (double)batch_wave_last_;
  out->woodbury_dual_cols = (d_.wb.on && d_.wb.dual) ? (double)d_.wb.cd : 0.0;
  out->woodbury_fused_iteration = be::wbf_active(d_) ? (d_.wb.dense ? (d_.wb.thin ? 3.0 : 2.0) : 1.0) : 0.0;
  out->woodbury_one_launch = (be::wbx_active(d_) && d_.wb.x.one && !d_.wb.x.slots) ? 1.0 : 0.0;
  out->woodbury_rows = d_.wb.on ? d_.wb.r : 0; out->woodbury_direct = (d_.wb.on && d_.wb.exact) ? ((d_.wb.x.on) ? 2 : 1) : 0;
  out->windowed_blocks = d_.A.nwin + d_.B.nwin; out->row_blocks = d_.A.nblk + d_.B.nblk;
  out->reordered = reordered_ ? 1.0 : 0.0; out->reorder_ms = reorder_ms_;
  out->f1_far_columns = d_.f1.on && d_.f1.mix ? (double)d_.f1.nsp : 0.0;
  // which preconditioner the PCG of this handle runs with RIGHT NOW (the setting cg_precond = diagonal selects the Jacobi family; the
  // Woodbury correction for dense rows is the engine's addition: OSQPHipPolicy::woodbury / woodbury_large switch it off)
  out->preconditioner = settings.cg_precond != OSQP_DIAGONAL_PRECONDITIONER ? OSQP_HIP_PRECOND_NONE
                        : !d_.wb.on ? OSQP_HIP_PRECOND_JACOBI : (d_.wb.large ? OSQP_HIP_PRECOND_JACOBI_WOODBURY_DENSE : OSQP_HIP_PRECOND_JACOBI_WOODBURY);
  return OSQP_NO_ERROR;
}
int Engine::time_kernel(int which, int reps, double *ms) {
  if (!dev_ready_) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  be::activate(d_);
  if (which < 0 || which > 23 || reps <= 0 || !ms) return OSQP_DATA_VALIDATION_ERROR;
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
