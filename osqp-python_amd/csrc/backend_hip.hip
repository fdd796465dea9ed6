// backend_hip.hip -- gfx950 (MI355X / CDNA4) implementation of the device-op interface in backend.h.
//
// Everything on the ADMM / PCG hot path is a hand-written HIP kernel; there is no rocSPARSE / hipBLAS call and no
// CPU path.  The path is sparse and bandwidth-bound (~0.17 flop/byte), so no MFMA: the design goals are
//   * coalesced streaming of the CSR arrays -- "CSR-stream": a workgroup stages a contiguous chunk of <= kChunk products
//     val * x[col] in LDS with unit-stride global loads, then one (or two) lanes per row sum their LDS segment.  Banded
//     ("windowed") row blocks fetch the window of the input vector they touch with coalesced loads, keep it in LDS and gather
//     from LDS through 16-bit local indices (10 bytes per entry, no dependent global gather); rows longer than kLongRow get a
//     workgroup of their own;
//   * two kernels per PCG iteration (Chronopoulos-Gear single-reduction PCG with the vector update fused into the SpMV-A
//     kernel and  t = rho .* (A u)  kept by a recurrence), every dot product / norm fused into the kernel that produces its
//     operands, and no reduction result needed before a kernel's row EPILOGUE (late hooks);
//   * wave reductions with DPP moves, never __shfl (which compiles to an LDS round trip per 32-bit half and step);
//   * no host round trip inside a chunk of ADMM iterations, and no launch wasted on a converged PCG: which phase a launch runs
//     (KB / K2F on a B slot; K1 / K1F / KA on an A slot) is decided on the device from a phase record ("slot kernels");
//   * deterministic reductions: every kernel runs with exactly kGrid workgroups; a workgroup writes ONE partial per
//     reduced quantity (slot[blockIdx.x]) and the consumer kernel sums the kGrid partials in a fixed order
//     (no atomics), so iteration counts are reproducible run to run;
//   * launch batching with hipGraph (five captured strings of slot launches serve every chunk, see engine.cpp).
#include "hip_common.h"

namespace osqp_hip {
namespace be {

namespace {

// ---------------------------------------------------------------------------------------------- residual kernels
struct EKr1 : NoPrefetch {
  const double *z, *y, *dy, *l, *u, *E, *Einv;
  double pu = 0, au = 0, zu = 0, ps = 0, as = 0, zs = 0, du = 0, ds = 0, lhs = 0, sup = 0;
  __device__ __forceinline__ void operator()(int i, const double (&s)[1]) {
    const double ax = s[0], zi = z[i], pr = ax - zi, ei = Einv[i], dyi = dy[i], yi = y[i], li = l[i], ui = u[i];
    pu = nanmax(pu, fabs(ei * pr)); au = nanmax(au, fabs(ei * ax)); zu = nanmax(zu, fabs(ei * zi));
    ps = nanmax(ps, fabs(pr)); as = nanmax(as, fabs(ax)); zs = nanmax(zs, fabs(zi));
    du = nanmax(du, fabs(E[i] * dyi)); ds = nanmax(ds, fabs(dyi));
    lhs += ui * fmax(dyi, 0.0) + li * fmin(dyi, 0.0);                        // _osqp.py:811-813
    if (yi > 0.0 && ui < OSQP_INFTY * 1e-4) sup += ui * yi;
    else if (yi < 0.0 && li > -OSQP_INFTY * 1e-4) sup += li * yi;
  }
};
// cond != 0 (boundary group of a device-driven solve): run only when the chunk has finished at a termination check / adaptation point
__device__ __forceinline__ bool ctl_res_due(const Dev &d) { return d.ctl->chunk_done && d.ctl->ch_at_check && d.ctl->status == CTL_RUNNING; }
__global__ __launch_bounds__(kBlock) void k_res_m(Dev d, int cond) {
  __shared__ StreamLdsW<1, double> lds;
  if (cond && !ctl_res_due(d)) return;
  GVec g{d.x};
  EKr1 e{{}, d.z, d.y, d.dy, d.l, d.u, d.E, d.Einv};
  process_rows<1>(d.A, g, e, lds);
  __syncthreads();
  double *red = lds.red;
  put_partial(d.part, SL_RES0 + R_PRI_U, block_max(e.pu, red)); put_partial(d.part, SL_RES0 + R_AX_U, block_max(e.au, red));
  put_partial(d.part, SL_RES0 + R_Z_U, block_max(e.zu, red)); put_partial(d.part, SL_RES0 + R_PRI_S, block_max(e.ps, red));
  put_partial(d.part, SL_RES0 + R_AX_S, block_max(e.as, red)); put_partial(d.part, SL_RES0 + R_Z_S, block_max(e.zs, red));
  put_partial(d.part, SL_RES0 + R_DY_U, block_max(e.du, red)); put_partial(d.part, SL_RES0 + R_DY_S, block_max(e.ds, red));
  put_partial(d.part, SL_RES0 + R_PINF_LHS, block_sum(e.lhs, red)); put_partial(d.part, SL_RES0 + R_SUPP, block_sum(e.sup, red));
}

struct GTwo {        // P part -> sum 0 (with pn), A' part -> sum 1 (with pm)
  const double *pn, *pm; int n;
  __device__ __forceinline__ void operator()(int c, double a, double (&pr)[2]) const {
    if (c < n) { pr[0] = a * pn[c]; pr[1] = 0.0; } else { pr[0] = 0.0; pr[1] = a * pm[c - n]; }
  }
};
struct EKr2 : NoPrefetch {
  const double *x, *q, *dx, *D, *Dinv; double sigma;
  double du = 0, pu = 0, au = 0, ds = 0, ps = 0, as = 0, xu = 0, xs = 0, xpx = 0, qx = 0, qdx = 0, qns = 0, qnu = 0;
  __device__ __forceinline__ void operator()(int j, const double (&s)[2]) {
    const double xj = x[j], px = s[0] - sigma * xj, aty = s[1], qj = q[j], dr = px + qj + aty, di = Dinv[j], dxj = dx[j];
    qns = nanmax(qns, fabs(qj)); qnu = nanmax(qnu, fabs(di * qj));                          // _osqp.py:766-794 (the q terms)
    du = nanmax(du, fabs(di * dr)); pu = nanmax(pu, fabs(di * px)); au = nanmax(au, fabs(di * aty));
    ds = nanmax(ds, fabs(dr)); ps = nanmax(ps, fabs(px)); as = nanmax(as, fabs(aty));
    xu = nanmax(xu, fabs(D[j] * dxj)); xs = nanmax(xs, fabs(dxj));
    xpx += xj * px; qx += qj * xj; qdx += qj * dxj;
  }
};
__global__ __launch_bounds__(kBlock) void k_res_n(Dev d, int cond) {
  __shared__ StreamLds<2> lds;
  if (cond && !ctl_res_due(d)) return;
  GTwo g{d.x, d.y, d.n};
  EKr2 e{{}, d.x, d.q, d.dx, d.D, d.Dinv, d.sigma};
  process_rows<2>(d.B, g, e, lds);
  __syncthreads();
  double *red = lds.red;
  put_partial(d.part, SL_RES0 + R_DUA_U, block_max(e.du, red)); put_partial(d.part, SL_RES0 + R_PX_U, block_max(e.pu, red));
  put_partial(d.part, SL_RES0 + R_ATY_U, block_max(e.au, red)); put_partial(d.part, SL_RES0 + R_DUA_S, block_max(e.ds, red));
  put_partial(d.part, SL_RES0 + R_PX_S, block_max(e.ps, red)); put_partial(d.part, SL_RES0 + R_ATY_S, block_max(e.as, red));
  put_partial(d.part, SL_RES0 + R_DX_U, block_max(e.xu, red)); put_partial(d.part, SL_RES0 + R_DX_S, block_max(e.xs, red));
  put_partial(d.part, SL_RES0 + R_XPX, block_sum(e.xpx, red)); put_partial(d.part, SL_RES0 + R_QX, block_sum(e.qx, red));
  put_partial(d.part, SL_RES0 + R_QDX, block_sum(e.qdx, red));
  put_partial(d.part, SL_RES0 + R_QN_S, block_max(e.qns, red)); put_partial(d.part, SL_RES0 + R_QN_U, block_max(e.qnu, red));
}

__device__ __forceinline__ bool res_is_sum(int q) {
  return q == R_PINF_LHS || q == R_SUPP || q == R_XPX || q == R_QX || q == R_QDX || q == R_ADX_VIOL;
}
// final reduction of the per-workgroup partials: workgroup b handles quantity q0 + b
__device__ __forceinline__ bool ctl_stage2_due(const Dev &d, int bits) { return d.ctl->status == CTL_RUNNING && (d.ctl->stage2 & bits) != 0; }
__global__ __launch_bounds__(kBlock) void k_res_final(Dev d, int q0, int cond) {        // cond 1: first stage of a boundary group, 2: its second stage
  __shared__ double sred[2 * kWaves];
  if (cond == 1 && !ctl_res_due(d)) return;
  if (cond == 2 && !ctl_stage2_due(d, NEED_PINF | NEED_DINF)) return;
  const int q = q0 + blockIdx.x;
  const double *slot = d.part + (SL_RES0 + q) * kGrid;
  const double v = res_is_sum(q) ? partial_sum(slot, sred) : partial_max(slot, sred);
  if (threadIdx.x == 0) d.res[q] = v;
}

// second-stage infeasibility tests (rare) -------------------------------------------------------
struct GAtOnly { const double *pm; int n; __device__ __forceinline__ void operator()(int c, double a, double (&pr)[1]) const { pr[0] = c >= n ? a * pm[c - n] : 0.0; } };
struct GPOnly { const double *pn; int n; __device__ __forceinline__ void operator()(int c, double a, double (&pr)[1]) const { pr[0] = c < n ? a * pn[c] : 0.0; } };
struct EAbs2 : NoPrefetch {
  const double *scale, *sub; double sigma; double mu = 0, ms = 0;
  __device__ __forceinline__ void operator()(int j, const double (&s)[1]) {
    const double v = s[0] - (sub ? sigma * sub[j] : 0.0);
    mu = nanmax(mu, fabs(scale[j] * v)); ms = nanmax(ms, fabs(v));
  }
};
__global__ __launch_bounds__(kBlock) void k_inf_primal(Dev d, int cond) {            // || Dinv A' dy ||_inf  (_osqp.py:815-818)
  __shared__ StreamLds<1> lds;
  if (cond && !ctl_stage2_due(d, NEED_PINF)) return;
  GAtOnly g{d.dy, d.n};
  EAbs2 e{{}, d.Dinv, nullptr, 0.0};
  process_rows<1>(d.B, g, e, lds);
  __syncthreads();
  put_partial(d.part, SL_RES0 + R_ATDY_U, block_max(e.mu, lds.red)); put_partial(d.part, SL_RES0 + R_ATDY_S, block_max(e.ms, lds.red));
}
__global__ __launch_bounds__(kBlock) void k_inf_dual_p(Dev d, int cond) {            // || Dinv P dx ||_inf   (_osqp.py:846-853)
  __shared__ StreamLds<1> lds;
  if (cond && !ctl_stage2_due(d, NEED_DINF)) return;
  GPOnly g{d.dx, d.n};
  EAbs2 e{{}, d.Dinv, d.dx, d.sigma};
  process_rows<1>(d.B, g, e, lds);
  __syncthreads();
  put_partial(d.part, SL_RES0 + R_PDX_U, block_max(e.mu, lds.red)); put_partial(d.part, SL_RES0 + R_PDX_S, block_max(e.ms, lds.red));
}
struct EViol : NoPrefetch {
  const double *l, *u, *Einv; double thr; int unscaled; double viol = 0;
  __device__ __forceinline__ void operator()(int i, const double (&s)[1]) {    // _osqp.py:861-872
    const double a = unscaled ? Einv[i] * s[0] : s[0];
    if ((u[i] < OSQP_INFTY * 1e-4 && a > thr) || (l[i] > -OSQP_INFTY * 1e-4 && a < -thr)) viol += 1.0;
  }
};
__global__ __launch_bounds__(kBlock) void k_inf_dual_a(Dev d, double thr, int unscaled, int cond) {
  __shared__ StreamLdsW<1, double> lds;
  if (cond) { if (!ctl_stage2_due(d, NEED_DINF)) return; thr = d.ctl->inf_thr_d; unscaled = d.ctl->inf_unscaled; }
  GVec g{d.dx};
  EViol e{{}, d.l, d.u, d.Einv, thr, unscaled};
  process_rows<1>(d.A, g, e, lds);
  __syncthreads();
  put_partial(d.part, SL_RES0 + R_ADX_VIOL, block_sum(e.viol, lds.red));
}

// ---------------------------------------------------------------------------------------------- rho / preconditioner / init
__global__ __launch_bounds__(kBlock) void k_set_rho(Dev d, double rho_bar, int cond) {
  if (cond) { if (!d.ctl->rho_flag) return; rho_bar = d.ctl->rho_bar; }        // boundary group: rho_bar as k_decide left it
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < d.m; i += stride) {
    const int t = d.ctype[i];
    const double eqf = d.eq_from_cnt ? (d.cnt[0] == 0 ? 1e3 : d.rho_eq_mixed) : d.rho_eq_factor;   // engine.cpp classify_constraints
    const double r = t == -1 ? 1e-6 : (t == 1 ? eqf * rho_bar : rho_bar);                 // _osqp.py:520-522, :1590-1594
    d.rho[i] = r; d.rho_inv[i] = 1.0 / r;
    d.v[i] = r * d.z[i] - d.y[i]; d.ztg[i] = d.zt[i]; d.t0[i] = r * d.zt[i];      // (the x~ sequence has a kink at a rho change: the next PCG starts from x~ itself)
  }
}
struct GPrec { const double *rho; int n; __device__ __forceinline__ void operator()(int c, double a, double (&pr)[1]) const { pr[0] = c >= n ? rho[c - n] * a * a : 0.0; } };
__global__ __launch_bounds__(kBlock) void k_precond(Dev d, int cond) {
  __shared__ StreamLds<1> lds;
  if (cond && !d.ctl->rho_flag) return;
  GPrec g{d.rho, d.n};
  EPrec e{{}, d.B.val, d.Bdiag, d.Minv};
  process_rows<1>(d.B, g, e, lds);
}
__global__ __launch_bounds__(kBlock) void k_fill(double *p, int n, double v) {
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) p[i] = v;
}
__global__ __launch_bounds__(kBlock) void k_init_n(Dev d) {
  const int stride = gridDim.x * kBlock;
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < d.n; j += stride) { d.xs[j] = d.x[j]; d.dx[j] = 0.0; }
}
__global__ __launch_bounds__(kBlock) void k_init_guess(Dev d, int cond) {      // no history: the next PCG starts from x~ itself
  if (cond && !d.ctl->rho_flag) return;
  const int stride = gridDim.x * kBlock;
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < d.n; j += stride) { const double v = d.xs[j]; d.xg[j] = v; d.xsp[j] = v; }
}
struct EInit : NoPrefetch {
  const double *rho, *y; double *z, *zt, *t0, *v, *dy; int full; double *ztg;
  __device__ __forceinline__ void operator()(int i, const double (&s)[1]) {
    const double a = s[0];
    if (full) { z[i] = a; dy[i] = 0.0; }
    zt[i] = a; ztg[i] = a; t0[i] = rho[i] * a; v[i] = rho[i] * z[i] - y[i];
  }
};
__global__ __launch_bounds__(kBlock) void k_init_m(Dev d, int full) {
  __shared__ StreamLdsW<1, double> lds;
  GVec g{d.xs};
  EInit e{{}, d.rho, d.y, d.z, d.zt, d.t0, d.v, d.dy, full, d.ztg};
  process_rows<1>(d.A, g, e, lds);
}
__global__ __launch_bounds__(kBlock) void k_normalcone(Dev d) {
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < d.m; i += stride) {
    const double t = d.z[i] + d.y[i], zn = fmin(fmax(t, d.l[i]), d.u[i]);
    d.z[i] = zn; d.y[i] = t - zn;
  }
}
__global__ void k_set_scal(double *scal, double rel, double ab) { scal[S_TOL_REL] = rel; scal[S_TOL_ABS] = ab; }
// vector updates on the device ----------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_scale_q(Dev d, double c) {                     // _osqp.py:1328
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < d.n; j += gridDim.x * kBlock) d.q[j] = c * d.D[j] * d.qraw[j];
}
__global__ __launch_bounds__(kBlock) void k_scale_bounds(Dev d, int rho_is_vec) {          // :1357-1358, :505-518 (on the scaled bounds)
  int ineq = 0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < d.m; i += gridDim.x * kBlock) {
    const double ls = d.E[i] * d.lraw[i], us = d.E[i] * d.uraw[i];
    d.l[i] = ls; d.u[i] = us;
    int t;
    if (ls < -OSQP_INFTY * 1e-4 && us > OSQP_INFTY * 1e-4) t = -1;
    else if (us - ls < 1e-4) t = 1;
    else t = 0;
    if (!rho_is_vec) t = 0;
    d.ctype[i] = t;
    ineq += (t == 0);
  }
  for (int o = 32; o > 0; o >>= 1) ineq += __shfl_down(ineq, o, 64);
  if ((threadIdx.x & 63) == 0 && ineq) atomicAdd(&d.cnt[0], ineq);                         // (integer: order-independent)
}
__global__ __launch_bounds__(kBlock) void k_count_bad(int m, const double *l, const double *u, int *cnt) {   // :1348-1349
  int bad = 0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < m; i += gridDim.x * kBlock) bad += !(l[i] <= u[i]);
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_down(bad, o, 64);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(&cnt[1], bad);
}
__global__ __launch_bounds__(kBlock) void k_gather(double *dst, const double *src, const int *idx, int cnt) {
  for (int k = blockIdx.x * kBlock + threadIdx.x; k < cnt; k += gridDim.x * kBlock) dst[k] = src[idx[k]];
}
__global__ __launch_bounds__(kBlock) void k_scale_warm(Dev d, const double *xin, const double *yin, double c) {   // :1493-1545
  const int stride = gridDim.x * kBlock;
  if (xin) for (int j = blockIdx.x * kBlock + threadIdx.x; j < d.n; j += stride) d.x[j] = d.Dinv[j] * xin[j];
  if (yin) for (int i = blockIdx.x * kBlock + threadIdx.x; i < d.m; i += stride) d.y[i] = c * d.Einv[i] * yin[i];
}



// ---------------------------------------------------------------------------------------------- assembly and scaling
// (setup / update_data_mat only: plain grid-stride kernels, one thread per entry or per row)
__device__ __forceinline__ double limit_scaling_dev(double v) { return v < 1e-4 ? 1.0 : (v > 1e4 ? 1e4 : v); }     // _osqp.py:363-387
__global__ __launch_bounds__(kBlock) void k_asm_diag(Dev d, double sigma) {
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < d.n; j += gridDim.x * kBlock) d.B.val[d.Bdiag[j]] = sigma;
}
__global__ __launch_bounds__(kBlock) void k_asm_scatter(Dev d, int scaled, double c, double sigma) {
  const int stride = gridDim.x * kBlock;
  for (int k = blockIdx.x * kBlock + threadIdx.x; k < d.nzA; k += stride) {
    double v = d.Araw[k];
    if (scaled) v *= d.E[d.Ai[k]] * d.D[d.Aj[k]];
    d.A.val[d.AmA[k]] = v; d.B.val[d.AmB[k]] = v;
  }
  for (int k = blockIdx.x * kBlock + threadIdx.x; k < d.nzP; k += stride) {
    const int i = d.Pi[k], j = d.Pj[k];
    double v = d.Praw[k];
    if (scaled) v *= c * d.D[i] * d.D[j];
    if (i == j) atomicAdd(&d.B.val[d.Pm1[k]], v);      // onto the sigma k_asm_diag stored (repeated (j, j) entries of a valid CSC sum up)
    else { d.B.val[d.Pm1[k]] = v; d.B.val[d.Pm2[k]] = v; }
  }
}
// out[r] = max |val| over the entries of row r with column < climit
// Long rows (more than kLongRow entries: their own row block, DevCsr::blkdesc carries a negative end row) are walked by a whole workgroup,
// the others by one thread each.  (Until round 4 every row was one thread's loop: the 5 000- and 10 000-entry rows of the lasso / portfolio
// matrices made each of the ~70 equilibration launches of a setup take 1-12 ms.)
template <class F>
__device__ __forceinline__ void for_long_rows(const DevCsr &M, F &&f) {      // f(row, first entry, end entry), workgroup-uniform
  const int4 *desc = reinterpret_cast<const int4 *>(M.blkdesc);
  for (int b = blockIdx.x; b < M.nblk; b += gridDim.x) { const int4 ds = desc[b]; if (ds.y < 0) f(ds.x, ds.z, ds.w); }
}
__global__ __launch_bounds__(kBlock) void k_rowmax(DevCsr M, int climit, double *out) {
  __shared__ double sred[2 * kWaves];
  for (int r = blockIdx.x * kBlock + threadIdx.x; r < M.nrows; r += gridDim.x * kBlock) {
    const int k0 = M.rowptr[r], k1 = M.rowptr[r + 1];
    if (k1 - k0 > kLongRow) continue;
    double mx = 0.0;
    for (int k = k0; k < k1; k++) if (M.col[k] < climit) mx = fmax(mx, fabs(M.val[k]));
    out[r] = mx;
  }
  for_long_rows(M, [&](int r, int k0, int k1) {
    double mx = 0.0;
    for (int k = k0 + threadIdx.x; k < k1; k += kBlock) if (M.col[k] < climit) mx = fmax(mx, fabs(M.val[k]));
    mx = block_max(mx, sred);
    if (threadIdx.x == 0) out[r] = mx;
  });
}
__global__ __launch_bounds__(kBlock) void k_ruiz_delta(double *v, int cnt) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < cnt; i += gridDim.x * kBlock) v[i] = 1.0 / sqrt(limit_scaling_dev(v[i]));
}
// A <- diag(et) A diag(dt) ; E *= et
__global__ __launch_bounds__(kBlock) void k_ruiz_scale_A(Dev d, const double *dt, const double *et) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < d.m; i += gridDim.x * kBlock) {
    const double ei = et[i];
    const int k0 = d.A.rowptr[i], k1 = d.A.rowptr[i + 1];
    if (k1 - k0 <= kLongRow) for (int k = k0; k < k1; k++) d.A.val[k] *= ei * dt[d.A.col[k]];
    d.E[i] *= ei;
  }
  for_long_rows(d.A, [&](int i, int k0, int k1) {
    const double ei = et[i];
    for (int k = k0 + threadIdx.x; k < k1; k += kBlock) d.A.val[k] *= ei * dt[d.A.col[k]];
  });
}
// B = [P | A'] <- [diag(dt) P diag(dt) | diag(dt) A' diag(et)] ; q *= dt ; D *= dt
__global__ __launch_bounds__(kBlock) void k_ruiz_scale_B(Dev d, const double *dt, const double *et) {
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < d.n; j += gridDim.x * kBlock) {
    const double dj = dt[j];
    const int k0 = d.B.rowptr[j], k1 = d.B.rowptr[j + 1];
    if (k1 - k0 <= kLongRow)
      for (int k = k0; k < k1; k++) {
        const int c = d.B.col[k];
        d.B.val[k] *= c < d.n ? dt[c] * dj : et[c - d.n] * dj;     // (same factor, same order of operands, as the entry's copy in A)
      }
    d.q[j] *= dj; d.D[j] *= dj;
  }
  for_long_rows(d.B, [&](int j, int k0, int k1) {
    const double dj = dt[j];
    for (int k = k0 + threadIdx.x; k < k1; k += kBlock) { const int c = d.B.col[k]; d.B.val[k] *= c < d.n ? dt[c] * dj : et[c - d.n] * dj; }
  });
}
// cost normalisation, one workgroup: ct = 1 / limit(max(limit(||q||_inf), mean_j ||P_:j||_inf)) ; c *= ct     (_osqp.py:443-448)
__global__ __launch_bounds__(kBlock) void k_ruiz_cost(Dev d, const double *np) {
  __shared__ double sred[2 * kWaves];
  double sum = 0.0, nq = 0.0;
  for (int j = threadIdx.x; j < d.n; j += kBlock) { sum += np[j]; nq = fmax(nq, fabs(d.q[j])); }
  block_sum_max(sum, nq, sred);
  if (threadIdx.x == 0) {
    const double mean = sum / (double)(d.n > 0 ? d.n : 1);
    const double ct = 1.0 / limit_scaling_dev(fmax(limit_scaling_dev(nq), mean));
    d.cs[1] = ct; d.cs[0] *= ct;
  }
}
__global__ __launch_bounds__(kBlock) void k_ruiz_cost_apply(Dev d) {
  const double ct = d.cs[1];
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < d.n; j += gridDim.x * kBlock) {
    for (int k = d.B.rowptr[j]; k < d.B.rowptr[j + 1] && d.B.col[k] < d.n; k++) d.B.val[k] *= ct;      // (the P part comes first in a row of B = [P | A'])
    d.q[j] *= ct;
  }
}
__global__ __launch_bounds__(kBlock) void k_ruiz_finish(Dev d, double sigma) {
  const int stride = gridDim.x * kBlock;
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < d.n; j += stride) { d.Dinv[j] = 1.0 / d.D[j]; d.B.val[d.Bdiag[j]] += sigma; }
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < d.m; i += stride) d.Einv[i] = 1.0 / d.E[i];
}

// ---------------------------------------------------------------------------------------------- boundary kernels (device-driven solves)
// The chunk described by the state block starts: phase record A (read by the next slot launch: strings are enqueued in pairs, so
// the launch after a boundary group has parity 0), PCG tolerance, statistics.  seq carries on across the chunks of a solve.
__device__ __forceinline__ void ctl_begin_chunk(const Dev &d, const Ctl &c, int seq) {
  int *r = d.slot;
  r[SR_PHASE] = P_KB; r[SR_K] = 0; r[SR_ADMM] = 0; r[SR_TARGET] = c.ch_next - c.iter; r[SR_USED] = 0; r[SR_CONV] = 0;
  r[SR_CAP] = c.budget[c.ch_tight]; r[SR_SEQ] = seq;
  r[SR_WORDS + SR_SEQ] = seq - 1; r[SR_WORDS + SR_ADMM] = 0;      // (record A is the newer one)
  d.scal[S_TOL_REL] = c.tol_rel; d.scal[S_TOL_ABS] = ctl_chunk_tol_abs(c);
  for (int q = F_STAT_SUM; q < F_COUNT; q++) d.flags[q] = 0;
}
__global__ void k_ctl_begin(Dev d, int epoch) {
  Ctl &c = *d.ctl;
  c.chunk_done = 0; c.rho_flag = 0; c.stage2 = 0; c.status = CTL_RUNNING; c.seq_begin = 0;
  ctl_begin_chunk(d, c, 0);
  d.slot[2 * SR_WORDS] = epoch;
}
// The rules of policy.h at a finished chunk.  One wave: the state block, the residual block and the statistics are staged in LDS
// (coalesced), lane 0 runs the rules on the LDS copy, the block goes back coalesced.
__global__ __launch_bounds__(64) void k_decide(Dev d, int stage) {
  __shared__ Ctl c;
  __shared__ double res[R_COUNT];
  __shared__ int fl[F_COUNT];
  Ctl *g = d.ctl;
  if (g->status != CTL_RUNNING) return;                                // the solve is over (or handed to the host)
  if (stage == 1 ? !g->chunk_done : !g->stage2) {                      // the chunk has not finished (short string) / no second stage pending
    // (a group that finds its chunk unfinished must not repeat the rho update the PREVIOUS boundary asked for: k_set_rho / k_init_guess
    //  in the middle of a chunk would reset the PCG start history -- results would depend on how the host timed its strings)
    if (stage == 1 && threadIdx.x == 0) g->rho_flag = 0;
    return;
  }
  static_assert(sizeof(Ctl) % sizeof(int) == 0, "Ctl is copied word by word");
  constexpr int W = sizeof(Ctl) / sizeof(int);
  const int *gi = reinterpret_cast<const int *>(g);
  int *ci = reinterpret_cast<int *>(&c);
  for (int i = threadIdx.x; i < W; i += 64) ci[i] = gi[i];
  for (int i = threadIdx.x; i < R_COUNT; i += 64) res[i] = d.res[i];
  for (int i = threadIdx.x; i < F_COUNT; i += 64) fl[i] = d.flags[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    int st;
    if (stage == 1) {
      c.chunk_done = 0;
      if (c.boundaries >= 0 && c.boundaries < kCtlHist) c.hist[c.boundaries] = c.seq_end - c.seq_begin;      // launches the chunk consumed (Engine::run_device_driven feeds the next solve from it)
      for (int q = 0; q < F_COUNT; q++) c.last_flags[q] = fl[q];
      st = ctl_boundary(c, res, fl);
    } else st = ctl_boundary_stage2(c, res, c.last_flags);
    c.status = st;
    if (st == CTL_RUNNING && !c.stage2) { c.seq_begin = d.slot[SR_SEQ]; ctl_begin_chunk(d, c, d.slot[SR_SEQ]); }       // (record A: written by the last slot launch of the string)
  }
  __syncthreads();
  int *go = reinterpret_cast<int *>(g);
  for (int i = threadIdx.x; i < W; i += 64) go[i] = ci[i];
}


}  // namespace

// ---------------------------------------------------------------------------------------------- interface
const char *name() { return "hip-gfx950"; }

int init(Dev &d, int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    std::fprintf(stderr, "osqp_hip: no HIP device available -- this engine has no CPU fallback\n");
    return OSQP_ALGEBRA_LOAD_ERROR;
  }
  if (device >= count) return OSQP_SETTINGS_VALIDATION_ERROR;
  d.device = device;
  HIP_CHECK(hipSetDevice(device));
  hipStream_t s;
  HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  d.stream = s;
  Impl *p = new Impl();
  HIP_CHECK(hipEventCreate(&p->ev0)); HIP_CHECK(hipEventCreate(&p->ev1)); HIP_CHECK(hipEventCreate(&p->ev_s0)); HIP_CHECK(hipEventCreate(&p->ev_s1)); HIP_CHECK(hipEventCreateWithFlags(&p->ev_ext, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&p->ev_wait, hipEventDisableTiming));
  HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&p->pin_res), sizeof(double) * R_COUNT, hipHostMallocDefault));
  HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&p->pin_flags), sizeof(int) * (F_COUNT + 16), hipHostMallocDefault));
  HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&p->pin_poll), sizeof(int) * kSlotInts, hipHostMallocDefault));
  HIP_CHECK(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
  HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&p->pin_ctl), sizeof(Ctl), hipHostMallocDefault));
  HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&p->pin_ctl2), sizeof(Ctl), hipHostMallocDefault));
  d.impl = p;
  return OSQP_NO_ERROR;
}
void destroy(Dev &d) {
  if (!d.impl) return;
  (void)hipSetDevice(d.device);
  Impl &p = im(d);
  (void)hipEventDestroy(p.ev0); (void)hipEventDestroy(p.ev1); (void)hipEventDestroy(p.ev_s0); (void)hipEventDestroy(p.ev_s1); (void)hipEventDestroy(p.ev_ext); (void)hipEventDestroy(p.ev_wait); (void)hipHostFree(p.pin_res); (void)hipHostFree(p.pin_flags);
  (void)hipHostFree(p.pin_poll); if (p.side) (void)hipStreamDestroy(p.side);
  (void)hipHostFree(p.pin_ctl); (void)hipHostFree(p.pin_ctl2);
  if (p.blas) wb_release_blas(p.blas);                 // (woodbury_hip.hip: the rocBLAS handle of the device-factorised form)
  batch_release(d);
  for (int k = 1; k < DevWb::kCache; k++) if (d.wb.cache_buf[k]) { (void)hipFree(d.wb.cache_buf[k]); d.wb.cache_buf[k] = nullptr; }      // (buffer 0 is the engine's own Sinv allocation)
  dev_release(d);
  delete &p; d.impl = nullptr;
  if (d.stream) { (void)hipStreamDestroy(st(d)); d.stream = nullptr; }
}
void *alloc(Dev &d, size_t bytes) {
  HIP_CHECK(hipSetDevice(d.device));
  void *p = nullptr;
  HIP_CHECK(hipMalloc(&p, bytes));
  HIP_CHECK(hipMemsetAsync(p, 0, bytes, st(d)));
  return p;
}
void dfree(Dev &d, void *p) { (void)hipSetDevice(d.device); (void)hipFree(p); }     // best effort: runs in destructors
void h2d(Dev &d, void *dst, const void *src, size_t b) {
  if (!b) return;
  HIP_CHECK(hipSetDevice(d.device));
  HIP_CHECK(hipMemcpyAsync(dst, src, b, hipMemcpyHostToDevice, st(d)));
  HIP_CHECK(hipStreamSynchronize(st(d)));      // the host buffer may be a temporary
}
void d2h(Dev &d, void *dst, const void *src, size_t b) {
  if (!b) return;
  HIP_CHECK(hipSetDevice(d.device));
  HIP_CHECK(hipMemcpyAsync(dst, src, b, hipMemcpyDeviceToHost, st(d)));
  HIP_CHECK(hipStreamSynchronize(st(d)));
}
void zero(Dev &d, void *dst, size_t b) { if (!b) return; HIP_CHECK(hipSetDevice(d.device)); HIP_CHECK(hipMemsetAsync(dst, 0, b, st(d))); }
void sync(Dev &d) { HIP_CHECK(hipSetDevice(d.device)); HIP_CHECK(hipStreamSynchronize(st(d))); }
void ev_mark(Dev &d, int which) { HIP_CHECK(hipSetDevice(d.device)); Impl &p = im(d); HIP_CHECK(hipEventRecord(which ? p.ev_s1 : p.ev_s0, st(d))); }
double ev_ms(Dev &d) { Impl &p = im(d); float ms = 0.f; HIP_CHECK(hipEventSynchronize(p.ev_s1)); HIP_CHECK(hipEventElapsedTime(&ms, p.ev_s0, p.ev_s1)); return (double)ms; }
void activate(Dev &d) { HIP_CHECK(hipSetDevice(d.device)); }
void ext_record(Dev &d, void *stream) {
  if (!stream || !d.impl) return;
  HIP_CHECK(hipSetDevice(d.device));
  HIP_CHECK(hipEventRecord(im(d).ev_ext, static_cast<hipStream_t>(stream)));
  im(d).ext_pending = true;
}
void ext_wait(Dev &d) {
  if (!d.impl || !im(d).ext_pending) return;
  HIP_CHECK(hipSetDevice(d.device));
  HIP_CHECK(hipEventSynchronize(im(d).ev_ext));
  im(d).ext_pending = false;
}

bool ctl_supported(const Dev &d) { return d.ctl != nullptr && d.slot != nullptr; }
void ctl_upload(Dev &d, const Ctl &c) {
  HIP_CHECK(hipSetDevice(d.device));
  Impl &p = im(d);
  std::memcpy(p.pin_ctl, &c, sizeof(Ctl));
  HIP_CHECK(hipMemcpyAsync(d.ctl, p.pin_ctl, sizeof(Ctl), hipMemcpyHostToDevice, st(d)));
  HIP_CHECK(hipStreamSynchronize(st(d)));          // (the staging buffer is reused)
}
void ctl_begin(Dev &d) { HIP_CHECK(hipSetDevice(d.device)); dev_publish(d); hipLaunchKernelGGL(k_ctl_begin, dim3(1), dim3(1), 0, st(d), d, ++im(d).epoch); }
void ctl_group(Dev &d, int diagonal) {
  LAUNCH(k_res_m, d, d, 1);
  LAUNCH(k_res_n, d, d, 1);
  hipLaunchKernelGGL(k_res_final, dim3(R_QN_U + 1), dim3(kBlock), 0, st(d), d, 0, 1);
  hipLaunchKernelGGL(k_decide, dim3(1), dim3(64), 0, st(d), d, 1);
  // second stage of the infeasibility tests, when the first asked for it (_osqp.py:815-818, :846-872)
  LAUNCH(k_inf_primal, d, d, 1);
  LAUNCH(k_inf_dual_p, d, d, 1);
  LAUNCH(k_inf_dual_a, d, d, 0.0, 0, 1);
  hipLaunchKernelGGL(k_res_final, dim3(5), dim3(kBlock), 0, st(d), d, (int)R_ATDY_U, 2);
  hipLaunchKernelGGL(k_decide, dim3(1), dim3(64), 0, st(d), d, 2);
  LAUNCH(k_set_rho, d, d, 0.0, 1);
  LAUNCH(k_init_guess, d, d, 1);
  if (diagonal) LAUNCH(k_precond, d, d, 1);
  kf_values(d, 1);                                        // K form: the explicit reduced matrix follows rho (no-op without the form)
  if (wbx_slots(d)) wb_factor_device(d, 1);               // Woodbury direct mode in the slot form: D0, S, S^-1 (+ check), S^-1 A_L when rho changed
}
void ctl_poll(Dev &d, Ctl *out, int *seq, int *done) {
  HIP_CHECK(hipSetDevice(d.device));
  Impl &p = im(d);
  // (the slot records FIRST: the state block is then at least as new as the launch count -- a count that says "the chunk's string is consumed" comes with
  //  the chunk_done flag its last KA set, Engine::run_device_driven's test of a chunk that outran its string)
  HIP_CHECK(hipMemcpyAsync(p.pin_poll, d.slot, sizeof(int) * kSlotInts, hipMemcpyDeviceToHost, p.side));
  HIP_CHECK(hipMemcpyAsync(p.pin_ctl2, d.ctl, sizeof(Ctl), hipMemcpyDeviceToHost, p.side));
  HIP_CHECK(hipStreamSynchronize(p.side));
  std::memcpy(out, p.pin_ctl2, sizeof(Ctl));
  if (p.pin_poll[2 * SR_WORDS] != p.epoch) { *seq = 0; *done = 0; out->status = CTL_RUNNING; return; }      // k_ctl_begin has not run yet
  const int *ra = p.pin_poll, *rb = p.pin_poll + SR_WORDS;
  const int *nw = ra[SR_SEQ] >= rb[SR_SEQ] ? ra : rb;
  *seq = nw[SR_SEQ]; *done = nw[SR_ADMM];
}
void ctl_download(Dev &d, Ctl *out) {
  HIP_CHECK(hipSetDevice(d.device));
  Impl &p = im(d);
  HIP_CHECK(hipMemcpyAsync(p.pin_ctl2, d.ctl, sizeof(Ctl), hipMemcpyDeviceToHost, st(d)));
  HIP_CHECK(hipStreamSynchronize(st(d)));
  std::memcpy(out, p.pin_ctl2, sizeof(Ctl));
}

void residuals(Dev &d) {
  HIP_CHECK(hipSetDevice(d.device));
  LAUNCH(k_res_m, d, d, 0);
  LAUNCH(k_res_n, d, d, 0);
  hipLaunchKernelGGL(k_res_final, dim3(R_QN_U + 1), dim3(kBlock), 0, st(d), d, 0, 0);
}
void infeas_primal(Dev &d) {
  HIP_CHECK(hipSetDevice(d.device));
  LAUNCH(k_inf_primal, d, d, 0);
  hipLaunchKernelGGL(k_res_final, dim3(2), dim3(kBlock), 0, st(d), d, (int)R_ATDY_U, 0);
}
void infeas_dual(Dev &d, double thr, int unscaled) {
  HIP_CHECK(hipSetDevice(d.device));
  LAUNCH(k_inf_dual_p, d, d, 0);
  LAUNCH(k_inf_dual_a, d, d, thr, unscaled, 0);
  hipLaunchKernelGGL(k_res_final, dim3(3), dim3(kBlock), 0, st(d), d, (int)R_PDX_U, 0);
}
void fetch_res(Dev &d, double *h) {
  HIP_CHECK(hipSetDevice(d.device));
  HIP_CHECK(hipMemcpyAsync(im(d).pin_res, d.res, sizeof(double) * R_COUNT, hipMemcpyDeviceToHost, st(d)));
  HIP_CHECK(hipStreamSynchronize(st(d)));
  std::memcpy(h, im(d).pin_res, sizeof(double) * R_COUNT);
}
void fetch_flags(Dev &d, int *h) {
  HIP_CHECK(hipSetDevice(d.device));
  HIP_CHECK(hipMemcpyAsync(im(d).pin_flags, d.flags, sizeof(int) * F_COUNT, hipMemcpyDeviceToHost, st(d)));
  if (d.slot) HIP_CHECK(hipMemcpyAsync(im(d).pin_flags + F_COUNT, d.slot, sizeof(int) * 16, hipMemcpyDeviceToHost, st(d)));
  HIP_CHECK(hipMemsetAsync(d.flags + F_STAT_SUM, 0, sizeof(int) * (F_COUNT - F_STAT_SUM), st(d)));
  HIP_CHECK(hipStreamSynchronize(st(d)));
  std::memcpy(h, im(d).pin_flags, sizeof(int) * F_COUNT);
}
void fetch_res_flags(Dev &d, double *hr, int *hf) {
  HIP_CHECK(hipSetDevice(d.device));
  HIP_CHECK(hipMemcpyAsync(im(d).pin_res, d.res, sizeof(double) * R_COUNT, hipMemcpyDeviceToHost, st(d)));
  HIP_CHECK(hipMemcpyAsync(im(d).pin_flags, d.flags, sizeof(int) * F_COUNT, hipMemcpyDeviceToHost, st(d)));
  if (d.slot) HIP_CHECK(hipMemcpyAsync(im(d).pin_flags + F_COUNT, d.slot, sizeof(int) * 16, hipMemcpyDeviceToHost, st(d)));
  HIP_CHECK(hipMemsetAsync(d.flags + F_STAT_SUM, 0, sizeof(int) * (F_COUNT - F_STAT_SUM), st(d)));
  HIP_CHECK(hipStreamSynchronize(st(d)));
  std::memcpy(hr, im(d).pin_res, sizeof(double) * R_COUNT);
  std::memcpy(hf, im(d).pin_flags, sizeof(int) * F_COUNT);
}

void set_rho(Dev &d, double rho_bar) { HIP_CHECK(hipSetDevice(d.device)); d.wb.rho_key = rho_bar; LAUNCH(k_set_rho, d, d, rho_bar, 0); LAUNCH(k_init_guess, d, d, 0); }
void precond(Dev &d, int diagonal) {
  HIP_CHECK(hipSetDevice(d.device));
  if (diagonal) LAUNCH(k_precond, d, d, 0);
  else LAUNCH(k_fill, d, d.Minv, d.n, 1.0);
  kf_values(d, 0);                                        // K form: K.val from the current rho / matrix values (every caller of precond has just changed one of them)
  if (d.wb.on) wb_factor(d);
}
void set_pcg_tol(Dev &d, double rel, double ab) {
  HIP_CHECK(hipSetDevice(d.device));
  hipLaunchKernelGGL(k_set_scal, dim3(1), dim3(1), 0, st(d), d.scal, rel, ab);
}
void init_iterates(Dev &d, int full) {
  HIP_CHECK(hipSetDevice(d.device));
  if (full) LAUNCH(k_init_n, d, d);                 // full = 2: x~ = x like 1, but the z iterate in place is kept (as full = 0 does)
  LAUNCH(k_init_guess, d, d, 0);
  LAUNCH(k_init_m, d, d, full == 1 ? 1 : 0);
}

bool device_vec_updates() { return true; }
void copy_in(Dev &d, void *dst, const void *src, size_t bytes, int src_on_device) {
  if (!bytes) return;
  HIP_CHECK(hipSetDevice(d.device));
  HIP_CHECK(hipMemcpyAsync(dst, src, bytes, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st(d)));
  if (!src_on_device) HIP_CHECK(hipStreamSynchronize(st(d)));      // the caller may reuse its (pageable) buffer as soon as the call returns
}
void stream_wait(Dev &d, void *caller_stream) {
  if (!caller_stream || caller_stream == d.stream) return;
  HIP_CHECK(hipSetDevice(d.device));
  HIP_CHECK(hipEventRecord(im(d).ev_wait, static_cast<hipStream_t>(caller_stream)));      // (its own event: ev_ext may still mark a pending batch kernel)
  HIP_CHECK(hipStreamWaitEvent(st(d), im(d).ev_wait, 0));
}
void gather(Dev &d, double *dst, const double *src, const int *idx, int cnt) { if (cnt <= 0) return; HIP_CHECK(hipSetDevice(d.device)); LAUNCH(k_gather, d, dst, src, idx, cnt); }
void scale_q(Dev &d, double c) { HIP_CHECK(hipSetDevice(d.device)); LAUNCH(k_scale_q, d, d, c); }
void scale_bounds(Dev &d, int rho_is_vec) {
  HIP_CHECK(hipSetDevice(d.device));
  HIP_CHECK(hipMemsetAsync(d.cnt, 0, sizeof(int), st(d)));
  if (d.m > 0) LAUNCH(k_scale_bounds, d, d, rho_is_vec);
}
int count_bad_bounds(Dev &d, const double *l, const double *u) {
  if (d.m == 0) return 0;
  HIP_CHECK(hipSetDevice(d.device));
  HIP_CHECK(hipMemsetAsync(d.cnt + 1, 0, sizeof(int), st(d)));
  LAUNCH(k_count_bad, d, d.m, l, u, d.cnt);
  int bad = 0;
  HIP_CHECK(hipMemcpyAsync(&bad, d.cnt + 1, sizeof(int), hipMemcpyDeviceToHost, st(d)));
  HIP_CHECK(hipStreamSynchronize(st(d)));
  return bad;
}
void scale_warm(Dev &d, const double *x, const double *y, double c) { HIP_CHECK(hipSetDevice(d.device)); LAUNCH(k_scale_warm, d, d, x, y, c); }

void project_normalcone(Dev &d) { HIP_CHECK(hipSetDevice(d.device)); LAUNCH(k_normalcone, d, d); }


bool device_assembly() { return true; }
void assemble(Dev &d, int scaled, double c, int with_sigma) {
  HIP_CHECK(hipSetDevice(d.device));
  const double sg = with_sigma ? d.sigma : 0.0;
  LAUNCH(k_asm_diag, d, d, sg);
  LAUNCH(k_asm_scatter, d, d, scaled, c, sg);
}
double ruiz(Dev &d, int iters) {
  HIP_CHECK(hipSetDevice(d.device));
  const double one[2] = {1.0, 1.0};
  HIP_CHECK(hipMemcpyAsync(d.cs, one, sizeof(one), hipMemcpyHostToDevice, st(d)));
  LAUNCH(k_fill, d, d.D, d.n, 1.0);
  if (d.m > 0) LAUNCH(k_fill, d, d.E, d.m, 1.0);
  double *dt = d.w, *np = d.p, *et = d.t;            // PCG work vectors are free during setup
  for (int it = 0; it < iters; it++) {
    LAUNCH(k_rowmax, d, d.B, d.n + d.m, dt);          // KKT column j = row j of [P | A']      (_norm_KKT_cols :348-361)
    LAUNCH(k_ruiz_delta, d, dt, d.n);
    if (d.m > 0) { LAUNCH(k_rowmax, d, d.A, d.n, et); LAUNCH(k_ruiz_delta, d, et, d.m); LAUNCH(k_ruiz_scale_A, d, d, dt, et); }
    LAUNCH(k_ruiz_scale_B, d, d, dt, et);
    LAUNCH(k_rowmax, d, d.B, d.n, np);                // column norms of the scaled P
    hipLaunchKernelGGL(k_ruiz_cost, dim3(1), dim3(kBlock), 0, st(d), d, np);
    LAUNCH(k_ruiz_cost_apply, d, d);
  }
  LAUNCH(k_ruiz_finish, d, d, d.sigma);
  double cs[2];
  HIP_CHECK(hipMemcpyAsync(cs, d.cs, sizeof(cs), hipMemcpyDeviceToHost, st(d)));
  HIP_CHECK(hipStreamSynchronize(st(d)));
  return cs[0];
}

bool graphs_supported() { return true; }
void graph_begin(Dev &d) { HIP_CHECK(hipSetDevice(d.device)); HIP_CHECK(hipStreamBeginCapture(st(d), hipStreamCaptureModeThreadLocal)); }
void *graph_end(Dev &d) {
  hipGraph_t g = nullptr; hipGraphExec_t ex = nullptr;
  HIP_CHECK(hipStreamEndCapture(st(d), &g));
  HIP_CHECK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  HIP_CHECK(hipGraphDestroy(g));
  return ex;
}
void graph_launch(Dev &d, void *g) { HIP_CHECK(hipSetDevice(d.device)); HIP_CHECK(hipGraphLaunch(static_cast<hipGraphExec_t>(g), st(d))); }
void graph_free(Dev &d, void *g) { if (g) { (void)hipSetDevice(d.device); (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(g)); } }

void test_spmv(Dev &d, int which, const double *in, double *out) {
  HIP_CHECK(hipSetDevice(d.device));
  LAUNCH(k_test_spmv, d, which == 0 ? d.A : d.B, in, out);
}


}  // namespace be
}  // namespace osqp_hip
