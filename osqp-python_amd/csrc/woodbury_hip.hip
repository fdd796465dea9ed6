// woodbury_hip.hip -- Woodbury-corrected Jacobi preconditioner for the dense rows of A (backend.h DevWb): the small form
// (r <= 128 rows, S inverted on the host) and the device-factorised form (r <= 16384: one fp64 GEMM + Cholesky + inverse through rocBLAS / rocSOLVER,
// loaded on demand).  Split out of backend_hip.hip in round 4.
#include <rocblas/rocblas.h>          // types and prototypes only: the libraries are dlopen()ed (dense_libs)
#include <rocsolver/rocsolver.h>
#include <dlfcn.h>
#include <initializer_list>
#include <mutex>
#include "hip_common.h"

namespace osqp_hip {
namespace be {
// dense_hip.hip (declared here: hip_common.h is one of the two files the committed PMC summaries are stamped with)
void dense_gemm_sym(void *stream, int N, int K, double alpha, const double *A, long as_i, long as_k, const double *B, long bs_k, long bs_j, double *C, long ld);

namespace {

// ---------------------------------------------------------------------------------------------- Woodbury preconditioner (backend.h DevWb)
__global__ __launch_bounds__(kBlock) void k_wb_gather(Dev d) {
  const DevWb &w = d.wb;
  const int stride = gridDim.x * kBlock;
  for (int k = blockIdx.x * kBlock + threadIdx.x; k < w.AL.nnz; k += stride) w.AL.val[k] = d.A.val[w.al_src[k]];
  for (int k = blockIdx.x * kBlock + threadIdx.x; k < w.ALT.nnz; k += stride) w.ALT.val[k] = d.A.val[w.alt_src[k]];
  for (int a = 0; a < w.r; a++)                                            // dense transpose (pattern fixed: the other entries stay zero)
    for (int k = w.AL.rowptr[a] + blockIdx.x * kBlock + threadIdx.x; k < w.AL.rowptr[a + 1]; k += stride) w.WT[(size_t)w.AL.col[k] * w.r + a] = d.A.val[w.al_src[k]];
}
// D0 = B_jj + sum over the SHORT rows of rho_i A_ij^2
struct GPrecShort { const double *rho; const unsigned char *islong; int n; __device__ __forceinline__ void operator()(int c, double a, double (&pr)[1]) const { pr[0] = (c >= n && !islong[c - n]) ? rho[c - n] * a * a : 0.0; } };
__global__ __launch_bounds__(kBlock) void k_wb_diag(Dev d, int cond) {      // cond: inside a boundary group -- only when it updated rho
  __shared__ StreamLds<1> lds;
  if (cond && !d.ctl->rho_flag) return;
  GPrecShort g{d.rho, d.wb.islong, d.n};
  EPrec e{{}, d.B.val, d.Bdiag, d.wb.Dinv0};
  process_rows<1>(d.B, g, e, lds);
}
// S_ab = sum_j A_L[a,j] A_L[b,j] / D0_j + (a == b) / rho_a : workgroup a, thread (slice s, b); column j of A_L is contiguous in WT.
// The j loop is split over kWbSlices slices of the workgroup (j = s mod kWbSlices), four independent loads in flight per step, and the
// slices are summed in index order (deterministic).  (r03: one thread per (a, b) walked all n columns with one dependent load chain --
// 3.3 ms per rho update on the portfolio QP, a fifth of its solve.)
constexpr int kWbSlices = 8;
__global__ __launch_bounds__(kWbMaxRows * kWbSlices) void k_wb_S(Dev d, int cond) {
  const DevWb &w = d.wb;
  if (cond && !d.ctl->rho_flag) return;
  __shared__ double part[kWbSlices][kWbMaxRows];
  const int a = blockIdx.x, b = threadIdx.x & (kWbMaxRows - 1), s = threadIdx.x / kWbMaxRows, r = w.r, n = d.n;
  const int bb = b < r ? b : 0;
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
  int j = s;
  for (; j + 3 * kWbSlices < n; j += 4 * kWbSlices) {
    const size_t o0 = (size_t)j * r, o1 = (size_t)(j + kWbSlices) * r, o2 = (size_t)(j + 2 * kWbSlices) * r, o3 = (size_t)(j + 3 * kWbSlices) * r;
    const double a0 = w.WT[o0 + a], a1 = w.WT[o1 + a], a2 = w.WT[o2 + a], a3 = w.WT[o3 + a];      // (workgroup-uniform)
    const double b0 = w.WT[o0 + bb], b1 = w.WT[o1 + bb], b2 = w.WT[o2 + bb], b3 = w.WT[o3 + bb];
    const double d0 = w.Dinv0[j], d1 = w.Dinv0[j + kWbSlices], d2 = w.Dinv0[j + 2 * kWbSlices], d3 = w.Dinv0[j + 3 * kWbSlices];
    acc0 += a0 * d0 * b0; acc1 += a1 * d1 * b1; acc2 += a2 * d2 * b2; acc3 += a3 * d3 * b3;
  }
  for (; j < n; j += kWbSlices) acc0 += w.WT[(size_t)j * r + a] * w.Dinv0[j] * w.WT[(size_t)j * r + bb];
  part[s][b] = (acc0 + acc1) + (acc2 + acc3);
  __syncthreads();
  if (s == 0 && b < r) {
    double acc = 0.0;
    for (int q = 0; q < kWbSlices; q++) acc += part[q][b];
    if (a == b) acc += d.rho_inv[w.rows[a]];
    w.S[(size_t)a * r + b] = acc;
  }
}
// S^-1 on the device (small form, r <= kWbMaxRows): in-place Gauss-Jordan elimination of the SPD matrix in LDS by ONE workgroup (no pivoting:
// the pivots of an SPD matrix are positive), symmetrised, checked against S -- || S S^-1 - I ||_max must be at rounding level for the direct
// mode -- and written to w.Sinv.  w.info[0] = 1: accurate, 0: inaccurate (the direct mode must not be used), -1: a pivot was not positive.
// Until round 4 this was a host round trip (S down, Cholesky + triangular inverse on one core, S^-1 up: ~0.5 ms per rho update, and the reason
// the direct mode's chunk boundaries could not be decided on the device).  cond: inside a boundary group -- only when it updated rho; a failed
// check then hands the solve to the host (CTL_NEED_HOST / NEED_REFACTOR) and cancels the chunk the group has just begun.
constexpr int kInvT = 1024;
__global__ __launch_bounds__(kInvT) void k_wb_invert(Dev d, int cond) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const DevWb &w = d.wb;
  if (cond && !d.ctl->rho_flag) return;
  const int r = w.r, tid = threadIdx.x, rr = r * r;
  double *M = sm, *rowk = M + rr, *colk = rowk + r, *red = colk + r;      // red: kInvT / 64 doubles
  __shared__ double pmin_s;
  for (int e = tid; e < rr; e += kInvT) M[e] = w.S[e];
  if (tid == 0) pmin_s = 1e300;
  __syncthreads();
  for (int k = 0; k < r; k++) {
    const double p = M[k * r + k], pi = 1.0 / p;
    if (tid < r) { rowk[tid] = tid == k ? pi : M[k * r + tid] * pi; colk[tid] = M[tid * r + k]; }
    if (tid == 0 && !(p >= pmin_s)) pmin_s = p;                           // (a NaN pivot is kept)
    __syncthreads();
    for (int e = tid; e < rr; e += kInvT) {
      const int i = e / r, j = e - i * r;
      M[e] = i == k ? rowk[j] : (j == k ? -colk[i] * pi : M[e] - colk[i] * rowk[j]);
    }
    __syncthreads();
  }
  // symmetrise (the elimination keeps symmetry up to rounding; consumers read rows as columns), then || S S^-1 - I ||_max
  for (int e = tid; e < rr; e += kInvT) { const int i = e / r, j = e - i * r; if (i < j) { const double v = 0.5 * (M[e] + M[j * r + i]); M[e] = v; M[j * r + i] = v; } }
  __syncthreads();
  double err = 0.0;
  for (int e = tid; e < rr; e += kInvT) {
    const int a = e / r, b = e - a * r;
    double acc0 = a == b ? -1.0 : 0.0, acc1 = 0.0;
    int k = 0;
    for (; k + 1 < r; k += 2) { acc0 += w.S[a * r + k] * M[k * r + b]; acc1 += w.S[a * r + k + 1] * M[(k + 1) * r + b]; }
    if (k < r) acc0 += w.S[a * r + k] * M[k * r + b];
    err = nanmax(err, fabs(acc0 + acc1));
  }
  err = wave_max(err);
  if ((tid & 63) == kReduceLane) red[tid >> 6] = err;
  __syncthreads();
  for (int e = tid; e < rr; e += kInvT) w.Sinv[e] = M[e];
  if (tid == 0) {
    double e2 = 0.0;
    for (int q = 0; q < kInvT / 64; q++) e2 = nanmax(e2, red[q]);
    int ok = !(pmin_s > 0.0) ? -1 : ((e2 < 1e-9) ? 1 : 0);
    if (w.dbg && *w.dbg > 0) { *w.dbg -= 1; ok = 0; }          // test hook (OSQPHipPolicy::debug_fail_refactor)
    w.info[0] = ok;
    if (cond && ok != 1) {                                    // device-driven solve: the direct mode cannot continue at this rho -- the host takes over
      Ctl *c = d.ctl;
      c->status = CTL_NEED_HOST; c->need |= NEED_REFACTOR;
      d.slot[SR_PHASE] = P_KB; d.slot[SR_ADMM] = 0; d.slot[SR_TARGET] = 0;      // the chunk the group has just begun never starts: the slots behind it idle
    }
  }
}
struct GDr { const double *Dinv0, *r; __device__ __forceinline__ void operator()(int c, double a, double (&pr)[1]) const { pr[0] = a * Dinv0[c] * r[c]; } };
__global__ __launch_bounds__(kBlock) void k_wb_p1(Dev d) {                 // g = A_L (D0^-1 r)
  __shared__ StreamLds<1> lds;
  if (d.flags[F_DONE]) return;
  GDr g{d.wb.Dinv0, d.r};
  EStore e{{}, d.wb.g};
  process_rows<1>(d.wb.AL, g, e, lds);
}
__global__ __launch_bounds__(kWbMaxRows) void k_wb_p2(Dev d) {             // h = S^-1 g
  const DevWb &w = d.wb;
  if (d.flags[F_DONE]) return;
  __shared__ double sg[kWbMaxRows];
  const int a = threadIdx.x, r = w.r;
  if (a < r) sg[a] = w.g[a];
  __syncthreads();
  if (a >= r) return;
  double acc = 0.0;
  for (int b = 0; b < r; b++) acc += w.Sinv[(size_t)a * r + b] * sg[b];
  w.h[a] = acc;
}
struct EWb3 {
  const double *Dinv0, *r; double *uu; double *xs;       // xs != nullptr: the direct mode -- u = K^-1 r_0 is added to x~ right here
  double g = 0, rn = 0, pr = 0, pd = 0, px = 0;
  __device__ __forceinline__ void prefetch(int j) { pr = r[j]; pd = Dinv0[j]; if (xs) px = xs[j]; }
  __device__ __forceinline__ void operator()(int j, const double (&s)[1]) {
    const double u = pd * (pr - s[0]);
    uu[j] = u; g += pr * u; rn = nanmax(rn, fabs(pr));
    if (xs) xs[j] = px + u;
  }
};
__global__ __launch_bounds__(kBlock) void k_wb_direct(Dev d) {             // exact mode: x~ += u (u = K^-1 r_0); the PCG statistics see one iteration
  const int stride = gridDim.x * kBlock;
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < d.n; j += stride) d.xs[j] += d.uu[j];
  if (blockIdx.x == 0 && threadIdx.x == 0) { d.flags[F_DONE] = 1; d.flags[F_ITERS] = 1; }
}
// direct != 0 (exact mode, M = K): x~ = x_g + u in the same pass, and the PCG statistics see one iteration -- the flag is written by workgroup 0
// at its END and not read by this launch (a workgroup that starts late must not take it for the previous solve's)
__global__ __launch_bounds__(kBlock) void k_wb_p3(Dev d, int parity, int direct) {     // u = D0^-1 (r - A_L' h); partials gamma = <r, u>, ||r||_inf
  __shared__ StreamLds<1> lds;
  if (!direct && d.flags[F_DONE]) return;
  GVec g{d.wb.h};
  EWb3 e{d.wb.Dinv0, d.r, d.uu, direct ? d.xs : nullptr};
  process_rows<1>(d.wb.ALT, g, e, lds);
  __syncthreads();
  double G = e.g, RN = e.rn;
  block_sum_max(G, RN, lds.red);
  put_partial(d.part, SL_GAMMA0 + parity, G); put_partial(d.part, SL_RN0 + parity, RN);
  if (direct && blockIdx.x == 0 && threadIdx.x == 0) { d.flags[F_DONE] = 1; d.flags[F_ITERS] = 1; }
}

// ---- many long rows (DevWb::large): dense S on the device
// W[a][colmap[j]] = A_L[a, j] / sqrt(D0_j): workgroups stride over the long rows (the pattern is fixed, the other entries stay zero)
__global__ __launch_bounds__(kBlock) void k_wb_fillW(Dev d) {
  const DevWb &w = d.wb;
  for (int a = blockIdx.x; a < w.r; a += gridDim.x)
    for (int k = w.AL.rowptr[a] + threadIdx.x; k < w.AL.rowptr[a + 1]; k += kBlock) {
      const int j = w.AL.col[k];
      w.W[(size_t)a * w.ct + w.colmap[j]] = w.AL.val[k] * sqrt(w.Dinv0[j]);
    }
}
__global__ __launch_bounds__(kBlock) void k_wb_gather_large(Dev d) {
  const DevWb &w = d.wb;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x; k < (size_t)w.AL.nnz; k += stride) w.AL.val[k] = d.A.val[w.al_src[k]];
  for (size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x; k < (size_t)w.ALT.nnz; k += stride) w.ALT.val[k] = d.A.val[w.alt_src[k]];
}
__global__ __launch_bounds__(kBlock) void k_wb_adddiag(Dev d) {
  const DevWb &w = d.wb;
  for (int a = blockIdx.x * kBlock + threadIdx.x; a < w.r; a += gridDim.x * kBlock) w.S[(size_t)a * w.r + a] += d.rho_inv[w.rows[a]];
}
// the factorisation works on one triangle (entries M[c * r + q], q >= c): mirror it
__global__ __launch_bounds__(kBlock) void k_wb_symm(double *M, int r) {
  for (int c = blockIdx.x; c < r; c += gridDim.x)
    for (int q = c + 1 + threadIdx.x; q < r; q += kBlock) M[(size_t)q * r + c] = M[(size_t)c * r + q];
}
// out = M in  (M: r x r, symmetric, full storage): one workgroup per row, 8 r^2 bytes per launch -- HBM-bound
__global__ __launch_bounds__(kBlock) void k_wb_gemv(const double *M, const double *in, double *out, int r, const int *done) {
  __shared__ double red[2 * kWaves];
  if (done && *done) return;
  for (int a = blockIdx.x; a < r; a += gridDim.x) {
    const double *row = M + (size_t)a * r;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    int k = threadIdx.x;
    for (; k + 3 * kBlock < r; k += 4 * kBlock) {
      const double m0 = row[k], m1 = row[k + kBlock], m2 = row[k + 2 * kBlock], m3 = row[k + 3 * kBlock];
      acc0 += m0 * in[k]; acc1 += m1 * in[k + kBlock]; acc2 += m2 * in[k + 2 * kBlock]; acc3 += m3 * in[k + 3 * kBlock];
    }
    for (; k < r; k += kBlock) acc0 += row[k] * in[k];
    const double tot = block_sum((acc0 + acc1) + (acc2 + acc3), red);
    if (threadIdx.x == 0) out[a] = tot;
  }
}
// probe of the direct mode: v, rho .* (A v), comparison of M^-1 K v with v
__global__ __launch_bounds__(kBlock) void k_wb_probe_init(Dev d) {
  const DevWb &w = d.wb;
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < d.n; j += gridDim.x * kBlock) {
    const double v = 0.5 + (double)((((unsigned)j * 2654435761u) >> 8) & 0xffffu) / 65536.0;
    w.pv[j] = v; w.pv[(size_t)d.n + d.m + j] = v;
  }
}
__global__ __launch_bounds__(kBlock) void k_wb_probe_rho(Dev d) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < d.m; i += gridDim.x * kBlock) d.wb.pv[(size_t)d.n + i] *= d.rho[i];
}
__global__ __launch_bounds__(kBlock) void k_wb_maxdiff(const double *a, const double *b, int cnt, double *out) {      // one workgroup: out[0] = max |a - b|, out[1] = max |b|
  __shared__ double red[2 * kWaves];
  double e = 0.0, s = 0.0;
  for (int j = threadIdx.x; j < cnt; j += kBlock) { e = nanmax(e, fabs(a[j] - b[j])); s = nanmax(s, fabs(b[j])); }
  block_max2(e, s, red);
  if (threadIdx.x == 0) { out[0] = e; out[1] = s; }
}
// ---- column-space ("dual") form (backend.h DevWb::dual)
__global__ __launch_bounds__(kBlock) void k_wbd_gather(Dev d) {            // the singleton entries' values <- A.val
  const DevWb &w = d.wb;
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < d.n; j += gridDim.x * kBlock) if (w.kind[j] == 2) w.sval[j] = d.A.val[w.ssrc[j]];
}
__global__ __launch_bounds__(kBlock) void k_wbd_weights(Dev d) {           // sigma_a, 1 + rho_a sigma_a, w_a  (one thread per long row; its singletons in list order)
  const DevWb &w = d.wb;
  for (int a = blockIdx.x * kBlock + threadIdx.x; a < w.r; a += gridDim.x * kBlock) {
    double sg = 0.0;
    for (int k = w.sg_ptr[a]; k < w.sg_ptr[a + 1]; k++) { const int j = w.sg_col[k]; const double v = w.sval[j]; sg += v * v * w.Dinv0[j]; }
    const double rho = d.rho[w.rows[a]], dn = 1.0 + rho * sg;
    w.den[a] = dn; w.wv[a] = rho / dn; if (w.sig) w.sig[a] = sg;
  }
}
__global__ __launch_bounds__(kBlock) void k_wbd_fillW(Dev d) {             // W[a][colmap[j]] = sqrt(w_a) A_L[a, j] on the dense columns
  const DevWb &w = d.wb;
  for (int a = blockIdx.x; a < w.r; a += gridDim.x) {
    const double sw = sqrt(w.wv[a]);
    for (int k = w.AL.rowptr[a] + threadIdx.x; k < w.AL.rowptr[a + 1]; k += kBlock) {
      const int c = w.colmap[w.AL.col[k]];
      if (c >= 0) w.W[(size_t)a * w.cd + c] = w.AL.val[k] * sw;
    }
  }
}
__global__ __launch_bounds__(kBlock) void k_wbd_adddiag(Dev d) {           // T_kk += D0 of the k-th dense column
  const DevWb &w = d.wb;
  for (int k = blockIdx.x * kBlock + threadIdx.x; k < w.cd; k += gridDim.x * kBlock) w.S[(size_t)k * w.cd + k] += 1.0 / w.Dinv0[w.dcol[k]];
}
__global__ __launch_bounds__(kBlock) void k_wbd_beta(Dev d) {              // beta_a, w_a beta_a
  const DevWb &w = d.wb;
  if (d.flags[F_DONE]) return;
  for (int a = blockIdx.x * kBlock + threadIdx.x; a < w.r; a += gridDim.x * kBlock) {
    double b = 0.0;
    for (int k = w.sg_ptr[a]; k < w.sg_ptr[a + 1]; k++) { const int j = w.sg_col[k]; b += w.sval[j] * w.Dinv0[j] * d.r[j]; }
    w.beta[a] = b; w.wbeta[a] = w.wv[a] * b;
  }
}
struct EWbdG : NoPrefetch {                                                // g_C = r_C - A_d' (w .* beta): rows of the transpose = columns; only the dense ones have an equation in T
  const int *kind, *colmap; const double *r; double *gc;
  __device__ __forceinline__ void operator()(int j, const double (&s)[1]) { if (kind[j] == 1) gc[colmap[j]] = r[j] - s[0]; }
};
__global__ __launch_bounds__(kBlock) void k_wbd_g(Dev d) {
  __shared__ StreamLds<1> lds;
  if (d.flags[F_DONE]) return;
  GVec g{d.wb.wbeta};
  EWbdG e{{}, d.wb.kind, d.wb.colmap, d.r, d.wb.g};
  process_rows<1>(d.wb.ALT, g, e, lds);
}
// x_C = T^-1 g_C, written straight to its columns of uz (one workgroup per row of T^-1; 8 cd^2 bytes per launch)
__global__ __launch_bounds__(kBlock) void k_wbd_gemv(Dev d) {
  __shared__ double red[2 * kWaves];
  const DevWb &w = d.wb;
  if (d.flags[F_DONE]) return;
  const int cd = w.cd;
  if ((cd & 1) == 0) {                                   // two rows per workgroup against one read of g, 16-byte loads (one row, 8-byte loads: 44.9 us at cd = 5 000)
    typedef double d2 __attribute__((ext_vector_type(2)));
    for (int a0 = blockIdx.x * 2; a0 < cd; a0 += gridDim.x * 2) {
      const bool two = a0 + 1 < cd;
      const double *r0 = w.Sinv + (size_t)a0 * cd, *r1 = w.Sinv + (size_t)(two ? a0 + 1 : a0) * cd;
      double s0 = 0.0, s1 = 0.0;
      int k = threadIdx.x * 2;
      for (; k + 3 * 2 * kBlock < cd; k += 4 * 2 * kBlock) {
        d2 u[4], p[4], q[4];
#pragma unroll
        for (int t = 0; t < 4; t++) { u[t] = *reinterpret_cast<const d2 *>(w.g + k + t * 2 * kBlock); p[t] = *reinterpret_cast<const d2 *>(r0 + k + t * 2 * kBlock); q[t] = *reinterpret_cast<const d2 *>(r1 + k + t * 2 * kBlock); }      // (T^-1 with non-temporal loads: 186 -> 189 us per iteration -- it is the one operand the Infinity Cache can keep)
#pragma unroll
        for (int t = 0; t < 4; t++) { s0 = fma(p[t].x, u[t].x, s0); s0 = fma(p[t].y, u[t].y, s0); s1 = fma(q[t].x, u[t].x, s1); s1 = fma(q[t].y, u[t].y, s1); }
      }
      for (; k < cd; k += 2 * kBlock) {
        const d2 u = *reinterpret_cast<const d2 *>(w.g + k), p = *reinterpret_cast<const d2 *>(r0 + k), q = *reinterpret_cast<const d2 *>(r1 + k);
        s0 = fma(p.x, u.x, s0); s0 = fma(p.y, u.y, s0); s1 = fma(q.x, u.x, s1); s1 = fma(q.y, u.y, s1);
      }
      const double t0 = block_sum(s0, red), t1 = block_sum(s1, red);
      if (threadIdx.x == 0) { w.uz[w.dcol[a0]] = t0; if (w.ud) w.ud[a0] = t0; }
      if (threadIdx.x == 64 && two) { w.uz[w.dcol[a0 + 1]] = t1; if (w.ud) w.ud[a0 + 1] = t1; }
    }
    return;
  }
  for (int a = blockIdx.x; a < cd; a += gridDim.x) {
    const double *row = w.Sinv + (size_t)a * cd;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    int k = threadIdx.x;
    for (; k + 3 * kBlock < cd; k += 4 * kBlock) {
      const double m0 = row[k], m1 = row[k + kBlock], m2 = row[k + 2 * kBlock], m3 = row[k + 3 * kBlock];
      acc0 += m0 * w.g[k]; acc1 += m1 * w.g[k + kBlock]; acc2 += m2 * w.g[k + 2 * kBlock]; acc3 += m3 * w.g[k + 3 * kBlock];
    }
    for (; k < cd; k += kBlock) acc0 += row[k] * w.g[k];
    const double tot = block_sum((acc0 + acc1) + (acc2 + acc3), red);
    if (threadIdx.x == 0) { w.uz[w.dcol[a]] = tot; if (w.ud) w.ud[a] = tot; }
  }
}
struct EWbdT : NoPrefetch {                                                // rho_a t_a = rho_a (beta_a + A_d[a] x_C) / (1 + rho_a sigma_a) = w_a (beta_a + A_d[a] x_C)
  const double *beta, *wv; double *rt;
  __device__ __forceinline__ void operator()(int a, const double (&s)[1]) { rt[a] = wv[a] * (beta[a] + s[0]); }
};
__global__ __launch_bounds__(kBlock) void k_wbd_t(Dev d) {
  __shared__ StreamLds<1> lds;
  if (d.flags[F_DONE]) return;
  GVec g{d.wb.uz};                                       // (zero on every column that is not dense: the singleton entries of a row drop out of the product)
  EWbdT e{{}, d.wb.beta, d.wb.wv, d.wb.rt};
  process_rows<1>(d.wb.AL, g, e, lds);
}
// u = M^-1 r column by column; partials gamma = <r, u>, ||r||_inf; direct: x~ += u and the PCG statistics see one iteration (as k_wb_p3)
__global__ __launch_bounds__(kBlock) void k_wbd_fin(Dev d, int parity, int direct) {
  __shared__ double red[2 * kWaves];
  const DevWb &w = d.wb;
  if (!direct && d.flags[F_DONE]) return;
  double g = 0.0, rn = 0.0;
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < d.n; j += gridDim.x * kBlock) {
    const int kd = w.kind[j];
    const double rj = d.r[j];
    const double u = kd == 1 ? w.uz[j] : (kd == 2 ? (rj - w.sval[j] * w.rt[w.srow[j]]) * w.Dinv0[j] : rj * w.Dinv0[j]);
    d.uu[j] = u; g += rj * u; rn = nanmax(rn, fabs(rj));
    if (direct) d.xs[j] += u;
  }
  block_sum_max(g, rn, red);
  put_partial(d.part, SL_GAMMA0 + parity, g); put_partial(d.part, SL_RN0 + parity, rn);
  if (direct && blockIdx.x == 0 && threadIdx.x == 0) { d.flags[F_DONE] = 1; d.flags[F_ITERS] = 1; }
}
// ---- fused ADMM iteration of the column-space direct mode (backend.h DevWb::fused).  Formulas of the iteration: _osqp.py:644-703; x~ = x_g + K^-1 r_0.
struct GWbf {                                              // row j of B = [P + sigma I | A']: sum 0 = ((P + sigma I) x_g)_j, sum 1 = (A' c)_j with c given as a - b (- nothing) or as one vector
  const double *xg, *ca, *cb; int n;
  __device__ __forceinline__ void operator()(int c, double a, double (&pr)[2]) const {
    if (c < n) { pr[0] = a * xg[c]; pr[1] = 0.0; }
    else { pr[0] = 0.0; pr[1] = a * (cb ? ca[c - n] - cb[c - n] : ca[c - n]); }
  }
};
struct EWbfR {                                             // non-dense columns: r_0j = sigma x_j - q_j - ((P + sigma I) x_g)_j + (A' c)_j
  const int *kind; const double *x, *q; double *r; double sigma; double px = 0, pq = 0;
  __device__ __forceinline__ void prefetch(int j) { px = x[j]; pq = q[j]; }
  __device__ __forceinline__ void operator()(int j, const double (&s)[2]) { if (kind[j] != 1) r[j] = sigma * px - pq - s[0] + s[1]; }
};
__global__ __launch_bounds__(kBlock) void k_wbf_r(Dev d) {
  __shared__ StreamLds<2> lds;
  GWbf g{d.xg, d.v, d.t0, d.n};
  EWbfR e{d.wb.kind, d.x, d.q, d.r, d.sigma};
  process_rows<2>(d.wb.Bn, g, e, lds);
  if (blockIdx.x == 0 && threadIdx.x == 0) { d.flags[F_DONE] = 0; d.flags[F_ITERS] = 0; }
}
__global__ __launch_bounds__(kBlock) void k_wbf_beta(Dev d) {                // per constraint row: cc = v - t0 (- w_a beta_a on dense row a); beta_a from r_0 at the row's singleton columns
  const DevWb &w = d.wb;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < d.m; i += gridDim.x * kBlock) {
    double c = d.v[i] - d.t0[i];
    if (w.islong[i]) {
      const int a = w.lidx[i];
      double b = 0.0;
      for (int k = w.sg_ptr[a]; k < w.sg_ptr[a + 1]; k++) { const int j = w.sg_col[k]; b += w.sval[j] * w.Dinv0[j] * d.r[j]; }
      w.beta[a] = b; c -= w.wv[a] * b;
      if (w.ccd) w.ccd[a] = c;
    }
    w.cc[i] = c;
  }
}
// (one sum, single product buffer: 16 KB of LDS and <= 64 registers -- EIGHT workgroups per CU, launched as 2 kGrid workgroups: these two passes stream
//  the 400 MB dense block; with four workgroups per CU the bytes in flight per CU did not cover the memory latency: 115 us per pass)
struct GWbf1 { const double *xg, *cc; int n; __device__ __forceinline__ void operator()(int c, double a, double (&pr)[1]) const { pr[0] = c < n ? -(a * xg[c]) : a * cc[c - n]; } };
struct EWbfG1 {
  const int *kind, *colmap; const double *x, *q; double *gc; double sigma; double px = 0, pq = 0;
  __device__ __forceinline__ void prefetch(int j) { px = x[j]; pq = q[j]; }
  __device__ __forceinline__ void operator()(int j, const double (&s)[1]) { if (kind[j] == 1) gc[colmap[j]] = sigma * px - pq + s[0]; }
};
constexpr int kWbfGrid = 2 * kGrid;
__global__ __launch_bounds__(kBlock) void k_wbf_g(Dev d) {
  __shared__ StreamLds<1, 1> lds;
  GWbf1 g{d.xg, d.wb.cc, d.n};
  EWbfG1 e{d.wb.kind, d.wb.colmap, d.x, d.q, d.wb.g, d.sigma};
  process_rows<1>(d.wb.Bd, g, e, lds);
}
// the z / y update of one constraint row from z~_i (_osqp.py:682-703), the next PCG start's A x_g by linearity and what the next right-hand side reads (as pcg_hip.hip EKa)
struct WbfRowUpd {
  const double *l, *u, *rho, *rho_inv; double *z, *y, *zt, *t0, *v, *dy, *ztg; double alpha, theta;
  __device__ __forceinline__ void operator()(int i, double ztil) const {
    const double prho = rho[i], pz = z[i], py = y[i], pzt = zt[i];
    const double zr = alpha * ztil + (1.0 - alpha) * pz;
    const double zn = fmin(fmax(zr + rho_inv[i] * py, l[i]), u[i]);
    const double dyi = prho * (zr - zn), yn = py + dyi;
    const double zg = ztil + theta * (ztil - pzt);
    y[i] = yn; dy[i] = dyi; z[i] = zn; zt[i] = ztil; v[i] = prho * zn - yn; ztg[i] = zg; t0[i] = prho * zg;
  }
};
struct EWbfT : NoPrefetch {                                // dense row a: rho_a t_a, then z~ = A x_g + A u with A u = A_d[a] u_C + beta_a - rho_a t_a sigma_a
  const double *beta, *wv, *sig; double *rt; const int *rows; WbfRowUpd up;
  __device__ __forceinline__ void operator()(int a, const double (&s)[1]) {
    const double t = wv[a] * (beta[a] + s[0]);
    rt[a] = t;
    const int i = rows[a];
    up(i, up.ztg[i] + (s[0] + beta[a] - t * sig[a]));
  }
};
__global__ __launch_bounds__(kBlock) void k_wbf_t(Dev d) {
  __shared__ StreamLds<1, 1> lds;
  GVec g{d.wb.uz};
  EWbfT e{{}, d.wb.beta, d.wb.wv, d.wb.sig, d.wb.rt, d.wb.rows, WbfRowUpd{d.l, d.u, d.rho, d.rho_inv, d.z, d.y, d.zt, d.t0, d.v, d.dy, d.ztg, d.alpha, d.theta}};
  process_rows<1>(d.wb.AL, g, e, lds);
}
// ---- the same two passes with the dense block held dense (backend.h DevWb::dense).  Ad [r][cd] row-major, cd even: every load is 16 bytes.
typedef double wb_d2 __attribute__((ext_vector_type(2)));
// The dense block is read once per pass and is larger than every cache (0.4 GB): its loads are NON-TEMPORAL, so that the two passes do not evict what the
// iteration reads again -- x_C, the partial sums, and T^-1 (0.2 GB, which then stays in the Infinity Cache from one iteration to the next): fused lasso
// iteration 198 -> 183 us (0.63 -> 0.69 of the HBM peak), 37.3 -> 34.7 ms per cold solve.
__device__ __forceinline__ wb_d2 wb_ld2(const double *p) { return __builtin_nontemporal_load(reinterpret_cast<const wb_d2 *>(p)); }
__global__ __launch_bounds__(kBlock) void k_wbd_fillAd(Dev d) {            // Ad[a][colmap[j]] = A_L[a, j] on the dense columns
  const DevWb &w = d.wb;
  for (int a = blockIdx.x; a < w.r; a += gridDim.x)
    for (int k = w.AL.rowptr[a] + threadIdx.x; k < w.AL.rowptr[a + 1]; k += kBlock) { const int c = w.colmap[w.AL.col[k]]; if (c >= 0) w.Ad[(size_t)a * w.cd + c] = w.AL.val[k]; }
}
// partial column sums  gp[rb][c] = sum over the rows a of block rb of Ad[a][c] cc_a:  workgroup (column tile of 2 kBlock columns, row block); a thread owns
// two adjacent columns and walks the block's rows, eight 16-byte loads in flight; the row's weight is a uniform (scalar) load.  Rows in ascending order:
// the sums do not depend on the launch geometry of anything else.
constexpr int kWbGdRows = 8;
__global__ __launch_bounds__(kBlock) void k_wbf_gd(Dev d) {
  const DevWb &w = d.wb;
  const int cd = w.cd, c0 = (blockIdx.x * kBlock + threadIdx.x) * 2, rb = blockIdx.y;
  const int per = (w.r + w.grb - 1) / w.grb, a0 = rb * per, a1 = min(w.r, a0 + per);
  if (c0 >= cd) return;
  const double *col = w.Ad + c0;
  double s0 = 0.0, s1 = 0.0;
  int a = a0;
  for (; a + kWbGdRows <= a1; a += kWbGdRows) {
    wb_d2 v[kWbGdRows];
#pragma unroll
    for (int k = 0; k < kWbGdRows; k++) v[k] = wb_ld2(col + (size_t)(a + k) * cd);
#pragma unroll
    for (int k = 0; k < kWbGdRows; k++) { const double c = w.ccd[a + k]; s0 = fma(v[k].x, c, s0); s1 = fma(v[k].y, c, s1); }
  }
  for (; a < a1; a++) { const wb_d2 v = wb_ld2(col + (size_t)a * cd); const double c = w.ccd[a]; s0 = fma(v.x, c, s0); s1 = fma(v.y, c, s1); }
  *reinterpret_cast<wb_d2 *>(w.gp + (size_t)rb * cd + c0) = wb_d2{s0, s1};
}
// g_C[c] = sigma x_j - q_j + (the entries of B's row j outside the dense rows: -(P + sigma I) x_g and the short rows' A' cc) + the row blocks' partial sums.
// 16 columns per workgroup, sixteen threads per column (row blocks rb = q, q + 16, ..; thread 0 of a column also takes the small list), summed in a fixed
// order.  (64 columns x 4 threads: 79 workgroups for 5 000 columns, 13.8 us.)
__global__ __launch_bounds__(kBlock) void k_wbf_gr(Dev d) {
  static_assert(kBlock == 256, "sixteen threads per column, 16 columns");
  __shared__ double part[16][17];
  const DevWb &w = d.wb;
  const int cx = threadIdx.x & 15, q = threadIdx.x >> 4, c = blockIdx.x * 16 + cx;
  double s = 0.0;
  if (c < w.cd) {
    for (int rb = q; rb < w.grb; rb += 16) s += w.gp[(size_t)rb * w.cd + c];
    if (q == 0) {
      const int j = w.dcol[c];
      double b = d.sigma * d.x[j] - d.q[j];
      for (int k = w.bq_ptr[c]; k < w.bq_ptr[c + 1]; k++) { const int col = w.bq_col[k]; const double a = d.B.val[w.bq_idx[k]]; b += col < d.n ? -(a * d.xg[col]) : a * w.cc[col - d.n]; }
      s += b;
    }
  }
  part[q][cx] = s;
  __syncthreads();
  if (q == 0 && c < w.cd) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) t += part[k][cx];
    w.g[c] = t;
  }
}
// rho_a t_a and the z / y update of the dense rows (EWbfT) from  s_a = Ad[a] . x_C:  two rows per workgroup against one read of x_C
__global__ __launch_bounds__(kBlock) void k_wbf_td(Dev d) {
  __shared__ double red[2 * kWaves];
  const DevWb &w = d.wb;
  const int cd = w.cd;
  EWbfT e{{}, w.beta, w.wv, w.sig, w.rt, w.rows, WbfRowUpd{d.l, d.u, d.rho, d.rho_inv, d.z, d.y, d.zt, d.t0, d.v, d.dy, d.ztg, d.alpha, d.theta}};
  for (int a0 = blockIdx.x * 2; a0 < w.r; a0 += gridDim.x * 2) {
    const bool two = a0 + 1 < w.r;
    const double *r0 = w.Ad + (size_t)a0 * cd, *r1 = w.Ad + (size_t)(two ? a0 + 1 : a0) * cd;
    double s0 = 0.0, s1 = 0.0;
    int k = threadIdx.x * 2;
    for (; k + 3 * 2 * kBlock < cd; k += 4 * 2 * kBlock) {
      wb_d2 u[4], p[4], q[4];
#pragma unroll
      for (int t = 0; t < 4; t++) { u[t] = *reinterpret_cast<const wb_d2 *>(w.ud + k + t * 2 * kBlock); p[t] = wb_ld2(r0 + k + t * 2 * kBlock); q[t] = wb_ld2(r1 + k + t * 2 * kBlock); }
#pragma unroll
      for (int t = 0; t < 4; t++) { s0 = fma(p[t].x, u[t].x, s0); s0 = fma(p[t].y, u[t].y, s0); s1 = fma(q[t].x, u[t].x, s1); s1 = fma(q[t].y, u[t].y, s1); }
    }
    for (; k < cd; k += 2 * kBlock) {
      const wb_d2 u = *reinterpret_cast<const wb_d2 *>(w.ud + k), p = wb_ld2(r0 + k), q = wb_ld2(r1 + k);
      s0 = fma(p.x, u.x, s0); s0 = fma(p.y, u.y, s0); s1 = fma(q.x, u.x, s1); s1 = fma(q.y, u.y, s1);
    }
    const double t0 = block_sum(s0, red), t1 = block_sum(s1, red);
    if (threadIdx.x == 0) { const double one[1] = {t0}; e(a0, one); }
    if (threadIdx.x == 64 && two) { const double one[1] = {t1}; e(a0 + 1, one); }
  }
}
__global__ __launch_bounds__(kBlock) void k_wbf_x(Dev d) {                   // u column by column, x~ = x_g + u, the x update (_osqp.py:660-668), the next PCG start
  const DevWb &w = d.wb;
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < d.n; j += gridDim.x * kBlock) {
    const int kd = w.kind[j];
    const double u = kd == 1 ? w.uz[j] : (kd == 2 ? (d.r[j] - w.sval[j] * w.rt[w.srow[j]]) * w.Dinv0[j] : d.r[j] * w.Dinv0[j]);
    const double xt = d.xg[j] + u, xo = d.x[j], xn = d.alpha * xt + (1.0 - d.alpha) * xo;
    d.uu[j] = u; d.xs[j] = xt; d.dx[j] = xn - xo; d.x[j] = xn;
    d.xg[j] = xt + d.theta * (xt - d.xsp[j]); d.xsp[j] = xt;
  }
}
struct EWbfS : NoPrefetch { WbfRowUpd up; const unsigned char *islong; __device__ __forceinline__ void operator()(int i, const double (&s)[1]) { if (!islong[i]) up(i, s[0]); } };
__global__ __launch_bounds__(kBlock) void k_wbf_s(Dev d) {                   // the short rows: z~ = A x~ and their z / y update; the PCG statistics see one iteration
  __shared__ StreamLds<1> lds;
  GVec g{d.xs};
  EWbfS e{{}, WbfRowUpd{d.l, d.u, d.rho, d.rho_inv, d.z, d.y, d.zt, d.t0, d.v, d.dy, d.ztg, d.alpha, d.theta}, d.wb.islong};
  process_rows<1>(d.wb.As, g, e, lds);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    d.flags[F_DONE] = 1; d.flags[F_ITERS] = 1;
    d.flags[F_STAT_SUM] += 1; d.flags[F_STAT_SUMSQ] += 1; d.flags[F_STAT_N] += 1;
    if (d.flags[F_STAT_MAX] < 1) d.flags[F_STAT_MAX] = 1;
  }
}
// ---- thin forms (DevWb::thin: every row of B of a column that is not dense and every short row of A has at most 64 entries): one thread per row, no LDS
// staging -- the process_rows launches these replace cost their fixed ~12 us each for a few entries per row.  k_wbf_rb = k_wbf_r + k_wbf_beta in one
// launch: a singleton column belongs to exactly one dense row, whose thread forms r_0 there itself.
__device__ __forceinline__ double wbf_rcol(const Dev &d, int j) {
  double s0 = 0.0, s1 = 0.0;
  for (int k = d.B.rowptr[j]; k < d.B.rowptr[j + 1]; k++) { const int c = d.B.col[k]; const double a = d.B.val[k]; if (c < d.n) s0 = fma(a, d.xg[c], s0); else s1 = fma(a, d.v[c - d.n] - d.t0[c - d.n], s1); }
  return d.sigma * d.x[j] - d.q[j] - s0 + s1;
}
__global__ __launch_bounds__(kBlock) void k_wbf_rb(Dev d) {
  const DevWb &w = d.wb;
  const int top = d.n > d.m ? d.n : d.m;
  for (int t = blockIdx.x * kBlock + threadIdx.x; t < top; t += gridDim.x * kBlock) {
    if (t < d.n && w.kind[t] == 0) d.r[t] = wbf_rcol(d, t);
    if (t < d.m) {
      double c = d.v[t] - d.t0[t];
      if (w.islong[t]) {
        const int a = w.lidx[t];
        double b = 0.0;
        for (int k = w.sg_ptr[a]; k < w.sg_ptr[a + 1]; k++) { const int j = w.sg_col[k]; const double rj = wbf_rcol(d, j); d.r[j] = rj; b += w.sval[j] * w.Dinv0[j] * rj; }
        w.beta[a] = b; c -= w.wv[a] * b;
        if (w.ccd) w.ccd[a] = c;
      }
      w.cc[t] = c;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { d.flags[F_DONE] = 0; d.flags[F_ITERS] = 0; }
}
__global__ __launch_bounds__(kBlock) void k_wbf_s2(Dev d) {
  const DevWb &w = d.wb;
  WbfRowUpd up{d.l, d.u, d.rho, d.rho_inv, d.z, d.y, d.zt, d.t0, d.v, d.dy, d.ztg, d.alpha, d.theta};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < d.m; i += gridDim.x * kBlock) {
    if (w.islong[i]) continue;
    double s = 0.0;
    for (int k = d.A.rowptr[i]; k < d.A.rowptr[i + 1]; k++) s = fma(d.A.val[k], d.xs[d.A.col[k]], s);
    up(i, s);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    d.flags[F_DONE] = 1; d.flags[F_ITERS] = 1;
    d.flags[F_STAT_SUM] += 1; d.flags[F_STAT_SUMSQ] += 1; d.flags[F_STAT_N] += 1;
    if (d.flags[F_STAT_MAX] < 1) d.flags[F_STAT_MAX] = 1;
  }
}
__global__ void k_wb_seq(double *g, int r) { for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < r; a += gridDim.x * blockDim.x) g[a] = 1.0 + 0.25 * (a % 7); }

}  // namespace

// ---- dense solver libraries, loaded on first use (rocBLAS: fp64 GEMM; rocSOLVER: Cholesky factorisation and inverse).  Nothing of the
// engine links against them: where they are missing the large-rank mode is off and such problems keep the Jacobi preconditioner.
namespace {
struct DenseLibs {
  bool tried = false, ok = false;
  void *hblas = nullptr, *hsolver = nullptr;
  rocblas_status (*create)(rocblas_handle *) = nullptr;
  rocblas_status (*destroy)(rocblas_handle) = nullptr;
  rocblas_status (*set_stream)(rocblas_handle, hipStream_t) = nullptr;
  rocblas_status (*dgemm)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int, rocblas_int, const double *, const double *, rocblas_int,
                          const double *, rocblas_int, const double *, double *, rocblas_int) = nullptr;
  rocblas_status (*dpotrf)(rocblas_handle, const rocblas_fill, const rocblas_int, double *, const rocblas_int, rocblas_int *) = nullptr;
  rocblas_status (*dpotri)(rocblas_handle, const rocblas_fill, const rocblas_int, double *, const rocblas_int, rocblas_int *) = nullptr;
};
DenseLibs &dense_libs() {
  static DenseLibs L;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (L.tried) return L;
  L.tried = true;
  auto open_any = [](std::initializer_list<const char *> names) -> void * { for (const char *nm : names) if (void *h = dlopen(nm, RTLD_NOW | RTLD_LOCAL)) return h; return nullptr; };
  L.hblas = open_any({"librocblas.so.5", "librocblas.so", "/opt/rocm/lib/librocblas.so"});
  L.hsolver = open_any({"librocsolver.so.0", "librocsolver.so", "/opt/rocm/lib/librocsolver.so"});
  if (!L.hblas || !L.hsolver) return L;
  auto sym = [](void *h, const char *nm) { return dlsym(h, nm); };
  L.create = reinterpret_cast<decltype(L.create)>(sym(L.hblas, "rocblas_create_handle"));
  L.destroy = reinterpret_cast<decltype(L.destroy)>(sym(L.hblas, "rocblas_destroy_handle"));
  L.set_stream = reinterpret_cast<decltype(L.set_stream)>(sym(L.hblas, "rocblas_set_stream"));
  L.dgemm = reinterpret_cast<decltype(L.dgemm)>(sym(L.hblas, "rocblas_dgemm"));
  L.dpotrf = reinterpret_cast<decltype(L.dpotrf)>(sym(L.hsolver, "rocsolver_dpotrf"));
  L.dpotri = reinterpret_cast<decltype(L.dpotri)>(sym(L.hsolver, "rocsolver_dpotri"));
  L.ok = L.create && L.destroy && L.set_stream && L.dgemm && L.dpotrf && L.dpotri;
  return L;
}
}  // namespace

namespace {

}  // namespace

void wb_release_blas(void *handle) { if (handle && dense_libs().ok) (void)dense_libs().destroy(static_cast<rocblas_handle>(handle)); }
bool wb_supported() { return true; }
void wb_refresh(Dev &d) {
  if (!d.wb.on) return;
  HIP_CHECK(hipSetDevice(d.device));
  if (d.wb.large) LAUNCH(k_wb_gather_large, d, d); else LAUNCH(k_wb_gather, d, d);
  if (d.wb.dual) LAUNCH(k_wbd_gather, d, d);
  if (d.wb.dense) LAUNCH(k_wbd_fillAd, d, d);
  for (int k = 0; k < DevWb::kCache; k++) d.wb.cache_rho[k] = -1.0;      // new matrix values: no inverse computed for the old ones may be looked up (the probe would reject it; this saves the probe)
}
void wb_direct(Dev &d) { LAUNCH(k_wb_direct, d, d); }
void wb_apply(Dev &d, int parity, int direct) {
  if (d.wb.dual) {                                          // column-space form: five launches, 8 (2 nnz(A_L) + cd^2) bytes
    hipLaunchKernelGGL(k_wbd_beta, dim3((d.wb.r + kBlock - 1) / kBlock), dim3(kBlock), 0, st(d), d);
    LAUNCH(k_wbd_g, d, d);
    hipLaunchKernelGGL(k_wbd_gemv, dim3(std::min(d.wb.cd, 8 * kGrid)), dim3(kBlock), 0, st(d), d);
    LAUNCH(k_wbd_t, d, d);
    LAUNCH(k_wbd_fin, d, d, parity, direct);
    return;
  }
  LAUNCH(k_wb_p1, d, d);
  if (d.wb.large) hipLaunchKernelGGL(k_wb_gemv, dim3(std::min(d.wb.r, 8 * kGrid)), dim3(kBlock), 0, st(d), d.wb.Sinv, d.wb.g, d.wb.h, d.wb.r, d.flags + F_DONE);
  else hipLaunchKernelGGL(k_wb_p2, dim3(1), dim3(kWbMaxRows), 0, st(d), d);
  LAUNCH(k_wb_p3, d, d, parity, direct);
}

bool wb_large_supported() { return true; }
void wbf_iteration(Dev &d) {
  const int top = d.n > d.m ? d.n : d.m;
  if (d.wb.thin) hipLaunchKernelGGL(k_wbf_rb, dim3(std::min((top + kBlock - 1) / kBlock, 4 * kGrid)), dim3(kBlock), 0, st(d), d);
  else {
    LAUNCH(k_wbf_r, d, d);
    hipLaunchKernelGGL(k_wbf_beta, dim3((d.m + kBlock - 1) / kBlock), dim3(kBlock), 0, st(d), d);
  }
  if (d.wb.dense) {
    hipLaunchKernelGGL(k_wbf_gd, dim3((d.wb.cd / 2 + kBlock - 1) / kBlock, d.wb.grb), dim3(kBlock), 0, st(d), d);
    hipLaunchKernelGGL(k_wbf_gr, dim3((d.wb.cd + 15) / 16), dim3(kBlock), 0, st(d), d);
  } else hipLaunchKernelGGL(k_wbf_g, dim3(kWbfGrid), dim3(kBlock), 0, st(d), d);
  hipLaunchKernelGGL(k_wbd_gemv, dim3(std::min(d.wb.cd, 8 * kGrid)), dim3(kBlock), 0, st(d), d);
  if (d.wb.dense) hipLaunchKernelGGL(k_wbf_td, dim3(std::min((d.wb.r + 1) / 2, 2 * kWbfGrid)), dim3(kBlock), 0, st(d), d);
  else hipLaunchKernelGGL(k_wbf_t, dim3(kWbfGrid), dim3(kBlock), 0, st(d), d);
  hipLaunchKernelGGL(k_wbf_x, dim3(std::min((d.n + kBlock - 1) / kBlock, kGrid)), dim3(kBlock), 0, st(d), d);
  if (d.wb.thin) hipLaunchKernelGGL(k_wbf_s2, dim3(std::min((d.m + kBlock - 1) / kBlock, 4 * kGrid)), dim3(kBlock), 0, st(d), d);
  else LAUNCH(k_wbf_s, d, d);
}      // (own kernels: dense_hip.hip; the vendor route needs the libraries, checked where it is asked for)

// D0, W, S = W W' + 1 / rho_L, S^-1 -- all on the device (r up to kWbLargeMax); then the two numerical checks
static void wb_factor_large(Dev &d) {
  DevWb &w = d.wb;
  Impl &p = im(d);
  rocblas_handle h = nullptr;
  static DenseLibs none;
  DenseLibs &L = w.vendor ? dense_libs() : none;             // (the libraries are touched only on the vendor route)
  if (w.vendor) {
    if (!L.ok) throw DeviceError("osqp_hip: the dense solver libraries are not available");
    if (!p.blas) {
      rocblas_handle hh = nullptr;
      if (L.create(&hh) != rocblas_status_success) throw DeviceError("osqp_hip: rocblas_create_handle failed");
      p.blas = hh;
    }
    h = static_cast<rocblas_handle>(p.blas);
    if (L.set_stream(h, st(d)) != rocblas_status_success) throw DeviceError("osqp_hip: rocblas_set_stream failed");
  }
  const int ct = w.ct;
  const int r = w.dual ? w.cd : w.r;                         // order of the dense system: S (row space, r x r) or T (column space, cd x cd)
  const bool log = w.log != 0;
  double tlap[6] = {0, 0, 0, 0, 0, 0};
  auto lap = [&](int k) { if (log) { HIP_CHECK(hipStreamSynchronize(st(d))); tlap[k] = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); } };
  lap(0);
  LAUNCH(k_wb_diag, d, d, 0);
  // (column-space form: the row weights w_a = rho_a / (1 + rho_a sigma_a) belong to the APPLICATION of M^-1 as much as to T -- they follow rho here, ahead of the
  //  look-up of a cached inverse, whose probe applies M^-1 with them)
  if (w.dual) hipLaunchKernelGGL(k_wbd_weights, dim3((w.r + kBlock - 1) / kBlock), dim3(kBlock), 0, st(d), d);
  double *out = w.pv + (size_t)d.n + d.m + d.n;                 // [4 + r] scratch behind the probe vectors
  // M^-1 (K v) = v ?   K v = B [v; rho .* (A v)]   -- the numerical test of "M is K" with the CURRENT matrices, rho vector and w.Sinv
  auto probe = [&]() {
    LAUNCH(k_wb_probe_init, d, d);
    LAUNCH(k_test_spmv, d, d.A, w.pv, w.pv + d.n);
    LAUNCH(k_wb_probe_rho, d, d);
    LAUNCH(k_test_spmv, d, d.B, w.pv, d.r);
    HIP_CHECK(hipMemsetAsync(d.flags + F_DONE, 0, sizeof(int), st(d)));
    wb_apply(d, 0);
    hipLaunchKernelGGL(k_wb_maxdiff, dim3(1), dim3(kBlock), 0, st(d), d.uu, w.pv + (size_t)d.n + d.m, d.n, out + 2);
  };
  // ---- an inverse computed for this rho_bar before?  (backend.h DevWb::cache_buf)  Only with the probe on: it is what validates the entry
  if (!w.cache_buf[0]) { w.cache_buf[0] = w.Sinv; w.cache_used = 0; w.cache_next = 0; }
  if (w.probe && w.cache_on) {
    for (int k = 0; k < w.cache_used; k++) {
      if (w.cache_rho[k] != w.rho_key) continue;
      w.Sinv = w.cache_buf[k];
      probe();
      double chk[2] = {1, 0};
      HIP_CHECK(hipMemcpyAsync(chk, out + 2, sizeof(chk), hipMemcpyDeviceToHost, st(d)));
      HIP_CHECK(hipStreamSynchronize(st(d)));
      if (chk[0] <= w.exact_tol * chk[1]) {
        w.exact = 1; w.cache_hits += 1;
        if (log) std::fprintf(stderr, "osqp_hip woodbury: rho %.6e: cached inverse, |M^-1 K v - v| %.2e / %.2e\n", w.rho_key, chk[0], chk[1]);
        return;
      }
      w.cache_rho[k] = -1.0;                                     // (the matrices or the rho vector have changed under the same rho_bar: the entry is dead)
      break;
    }
  }
  // the buffer this factorisation writes: a free one, else the oldest
  { int slot = -1;
    for (int k = 0; k < w.cache_used; k++) if (w.cache_rho[k] < 0) { slot = k; break; }
    if (slot < 0 && w.cache_used < ((w.probe && w.cache_on) ? DevWb::kCache : 1)) {      // (without the probe no look-up can ever hit: one buffer)
      slot = w.cache_used;
      if (!w.cache_buf[slot]) { void *b = nullptr; if (hipMalloc(&b, sizeof(double) * (size_t)r * r) != hipSuccess) { (void)hipGetLastError(); slot = -1; } else w.cache_buf[slot] = static_cast<double *>(b); }
      if (slot >= 0) w.cache_used += 1;
    }
    if (slot < 0) { slot = w.cache_next % std::max(1, w.cache_used); w.cache_next += 1; }
    w.Sinv = w.cache_buf[slot]; w.cache_rho[slot] = -1.0; w.cache_next = slot + 1;
  }
  const double one = 1.0, zero = 0.0;
  if (w.dual) {
    LAUNCH(k_wbd_fillW, d, d);
    lap(1);
    // W is r x cd row-major = cd x r column-major (ld cd): T = W_cm W_cm' (cd x cd, inner dimension r)
    if (!w.vendor) dense_gemm_sym(st(d), r, w.r, 1.0, w.W, 1, r, w.W, r, 1, w.S, r);             // T(i, j) = sum_a W[a][i] W[a][j]: one triangle on the matrix cores, mirrored
    else if (L.dgemm(h, rocblas_operation_none, rocblas_operation_transpose, r, r, w.r, &one, w.W, r, w.W, r, &zero, w.S, r) != rocblas_status_success)
      throw DeviceError("osqp_hip: rocblas_dgemm failed");
    LAUNCH(k_wbd_adddiag, d, d);
  } else {
  LAUNCH(k_wb_fillW, d, d);
  lap(1);
  // W is r x ct row-major = ct x r column-major (ld ct): S = W' W in the library's convention
  if (!w.vendor) dense_gemm_sym(st(d), r, ct, 1.0, w.W, ct, 1, w.W, 1, ct, w.S, r);               // S(a, b) = sum_j W[a][j] W[b][j]
  else if (L.dgemm(h, rocblas_operation_transpose, rocblas_operation_none, r, r, ct, &one, w.W, ct, w.W, ct, &zero, w.S, r) != rocblas_status_success)
    throw DeviceError("osqp_hip: rocblas_dgemm failed");
  LAUNCH(k_wb_adddiag, d, d);
  }
  lap(2);
  HIP_CHECK(hipMemcpyAsync(w.Sinv, w.S, sizeof(double) * (size_t)r * r, hipMemcpyDeviceToDevice, st(d)));
  double *minpiv = w.gjwork + wb_inverse_work(r);
  if (!w.vendor) {
    HIP_CHECK(hipMemsetAsync(w.info, 0, 2 * sizeof(int), st(d)));
    dense_spd_inverse(st(d), w.Sinv, r, r, w.gjwork, minpiv);      // block Gauss-Jordan on the matrix cores (dense_hip.hip)
    lap(3); lap(4);
  } else {
  if (L.dpotrf(h, rocblas_fill_lower, r, w.Sinv, r, w.info) != rocblas_status_success) throw DeviceError("osqp_hip: rocsolver_dpotrf failed");
  lap(3);
  if (L.dpotri(h, rocblas_fill_lower, r, w.Sinv, r, w.info + 1) != rocblas_status_success) throw DeviceError("osqp_hip: rocsolver_dpotri failed");
  lap(4);
  }
  hipLaunchKernelGGL(k_wb_symm, dim3(std::min(r, 8 * kGrid)), dim3(kBlock), 0, st(d), w.Sinv, r);
  // S^-1 against S on a fixed vector:  S (S^-1 g) = g
  hipLaunchKernelGGL(k_wb_seq, dim3(64), dim3(256), 0, st(d), w.g, r);
  hipLaunchKernelGGL(k_wb_gemv, dim3(std::min(r, 8 * kGrid)), dim3(kBlock), 0, st(d), w.Sinv, w.g, w.h, r, nullptr);
  hipLaunchKernelGGL(k_wb_gemv, dim3(std::min(r, 8 * kGrid)), dim3(kBlock), 0, st(d), w.S, w.h, out + 4, r, nullptr);
  hipLaunchKernelGGL(k_wb_maxdiff, dim3(1), dim3(kBlock), 0, st(d), out + 4, w.g, r, out);
  if (w.probe) probe();
  int info[2] = {0, 0};
  double chk[4] = {0, 1, 0, 1}, piv = 1.0;
  if (!w.vendor) HIP_CHECK(hipMemcpyAsync(&piv, minpiv, sizeof(double), hipMemcpyDeviceToHost, st(d)));
  HIP_CHECK(hipMemcpyAsync(info, w.info, sizeof(info), hipMemcpyDeviceToHost, st(d)));
  HIP_CHECK(hipMemcpyAsync(chk, out, sizeof(double) * (w.probe ? 4 : 2), hipMemcpyDeviceToHost, st(d)));
  HIP_CHECK(hipStreamSynchronize(st(d)));
  if (info[0] != 0 || info[1] != 0 || !(piv > 0.0)) throw DeviceError("osqp_hip: the Woodbury system of the preconditioner is not positive definite");
  const bool inv_ok = chk[0] <= 1e-8 * chk[1];
  w.exact = (w.probe && inv_ok && chk[2] <= w.exact_tol * chk[3]) ? 1 : 0;
  if (inv_ok) for (int k = 0; k < w.cache_used; k++) if (w.cache_buf[k] == w.Sinv) w.cache_rho[k] = w.rho_key;      // (the entry is valid from here on)
  if (log) {
    lap(5);
    std::fprintf(stderr, "osqp_hip woodbury (%s): order %d ct %d  |S S^-1 g - g| %.2e / %.2e   |M^-1 K v - v| %.2e / %.2e   direct %d;  D0 + W %.1f ms, GEMM %.1f ms, Cholesky %.1f ms, inverse %.1f ms, mirror + checks %.1f ms\n",
                 w.dual ? "column space" : "row space", r, ct, chk[0], chk[1], chk[2], chk[3], w.exact, 1e3 * (tlap[1] - tlap[0]), 1e3 * (tlap[2] - tlap[1]), 1e3 * (tlap[3] - tlap[2]), 1e3 * (tlap[4] - tlap[3]), 1e3 * (tlap[5] - tlap[4]));
  }
}
// D0, S, S^-1 (with its check) and the second tile of the two-launch direct mode: launches only -- conditional inside a boundary group
void wb_factor_device(Dev &d, int cond) {
  DevWb &w = d.wb;
  const int r = w.r;
  HIP_CHECK(hipSetDevice(d.device));
  static thread_local int attr_dev = -1;
  const size_t lds = sizeof(double) * ((size_t)r * r + 2 * (size_t)r + kInvT / 64);
  if (attr_dev != d.device) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wb_invert), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * ((size_t)kWbMaxRows * kWbMaxRows + 2 * kWbMaxRows + kInvT / 64))));
    attr_dev = d.device;
  }
  LAUNCH(k_wb_diag, d, d, cond);
  hipLaunchKernelGGL(k_wb_S, dim3(r), dim3(kWbMaxRows * kWbSlices), 0, st(d), d, cond);
  hipLaunchKernelGGL(k_wb_invert, dim3(1), dim3(kInvT), lds, st(d), d, cond);
  wbx_factor(d, cond);                                        // (two-launch direct mode: S^-1 A_L per column block)
}
// small form (r <= kWbMaxRows), host-synchronous: the launches above, then the verdict of the check
void wb_factor(Dev &d) {
  DevWb &w = d.wb;
  if (w.large) { wb_factor_large(d); return; }
  wb_factor_device(d, 0);
  int ok = 0;
  HIP_CHECK(hipMemcpyAsync(&ok, w.info, sizeof(int), hipMemcpyDeviceToHost, st(d)));
  HIP_CHECK(hipStreamSynchronize(st(d)));
  if (ok < 0) throw DeviceError("osqp_hip: the Woodbury system of the preconditioner is not positive definite");
  if (ok != 1) w.exact = 0;                                   // the direct mode trusts S^-1: || S S^-1 - I ||_max must be at rounding level
}

}  // namespace be
}  // namespace osqp_hip
