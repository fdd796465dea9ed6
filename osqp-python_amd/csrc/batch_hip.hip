// batch_hip.hip -- batched OSQP for many small QPs that share (P, A) and differ in q, l, u  (BASELINE configs[4]:
// 4096 MPC QPs, n = 120, m = 240; semantics = the reference's update-style batching, one solver re-used through
// update(q, l, u) + solve(), /root/reference/src/osqp/nn/torch.py:128-164).
//
// MI355X mapping: ONE WORKGROUP PER PROBLEM, one kernel launch for the whole batch.  All iterates and PCG vectors of a
// problem live in LDS (10 n + 8 m doubles = 25 KB at n=120, m=240 -> 6 workgroups per CU); the shared scaled matrices
// A (CSR) and B = [P + sigma I | A'] (CSR) are the base solver's device arrays (28 KB: L1/L2 resident for every
// workgroup).  The complete ADMM solve -- rhs, reduced-KKT PCG (Jacobi), x/z/y update, residuals, termination test,
// infeasibility tests, adaptive rho -- runs inside the kernel with __syncthreads() as the only synchronisation and
// wave64 __shfl_down + LDS reductions for every dot product / norm.  No host round trip, no global-memory iterates.
// The arithmetic is the same as the large-problem engine (backend_hip.hip / engine.cpp); formulas cite
// /root/reference/src/osqppurepy/_osqp.py.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/osqp_hip.h"
#include "backend.h"

namespace osqp_hip {
namespace be {

namespace {

__device__ __forceinline__ double nmax(double r, double a) { return (a > r || a != a) ? a : r; }
// value of v in lane `lane` (wave-uniform index) broadcast to the whole wave: two v_readlane_b32
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Wave64 reductions with DPP moves (VALU rate; HIP's __shfl_* compile to ds_bpermute_b32 -- an LDS round trip per 32-bit half
// and step; see backend_hip.hip wave_sum).  Zero fill = identity of the sums and of the maxima of magnitudes.  Result in LANE 63.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double bdpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wsum63(double v) {
  v += bdpp<0xb1>(v); v += bdpp<0x4e>(v); v += bdpp<0x114>(v); v += bdpp<0x118>(v); v += bdpp<0x142, 0xa>(v); v += bdpp<0x143, 0xc>(v);
  return v;
}
__device__ __forceinline__ double wmax63(double v) {
  v = nmax(v, bdpp<0xb1>(v)); v = nmax(v, bdpp<0x4e>(v)); v = nmax(v, bdpp<0x114>(v)); v = nmax(v, bdpp<0x118>(v));
  v = nmax(v, bdpp<0x142, 0xa>(v)); v = nmax(v, bdpp<0x143, 0xc>(v));
  return v;
}

// Block reductions; all threads get the result.  NW = waves per workgroup.  With ONE wave per problem (NW = 1) a
// reduction is six DPP steps and a v_readlane broadcast: no LDS, no barrier.
template <int NW>
struct Red {
  double *s;   // >= 16 doubles of LDS (NW > 1 only)
  __device__ __forceinline__ double sum(double v) const {
    v = wsum63(v);
    if constexpr (NW == 1) return readlane_f64(v, 63);
    else {
      if ((threadIdx.x & 63) == 63) s[threadIdx.x >> 6] = v;
      __syncthreads();
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < NW; w++) t += s[w];
      __syncthreads();
      return t;
    }
  }
  __device__ __forceinline__ double max(double v) const {
    v = wmax63(v);
    if constexpr (NW == 1) return readlane_f64(v, 63);
    else {
      if ((threadIdx.x & 63) == 63) s[threadIdx.x >> 6] = v;
      __syncthreads();
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < NW; w++) t = nmax(t, s[w]);
      __syncthreads();
      return t;
    }
  }
  // K maxima and S sums behind ONE pair of barriers (K + S <= 4: the 16 doubles of s): the same wave reductions and the same order over the waves as
  // max() / sum() -- identical results; the residuals' 22 block reductions cost 44 barriers one at a time, 14 in groups of four
  template <int K, int S>
  __device__ __forceinline__ void max_sum_n(double (&mx)[K > 0 ? K : 1], double (&sm)[S > 0 ? S : 1]) const {
    static_assert(NW * (K + S) <= 16, "scratch");
    double *scr = s;
    if constexpr (NW == 1) {
#pragma unroll
      for (int k = 0; k < K; k++) mx[k] = max(mx[k]);
#pragma unroll
      for (int k = 0; k < S; k++) sm[k] = sum(sm[k]);
    } else {
      const int w = threadIdx.x >> 6; const bool last = (threadIdx.x & 63) == 63;
#pragma unroll
      for (int k = 0; k < K; k++) { const double v = wmax63(mx[k]); if (last) scr[w * (K + S) + k] = v; }
#pragma unroll
      for (int k = 0; k < S; k++) { const double v = wsum63(sm[k]); if (last) scr[w * (K + S) + K + k] = v; }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < K; k++) { double t = 0.0; for (int q = 0; q < NW; q++) t = nmax(t, scr[q * (K + S) + k]); mx[k] = t; }
#pragma unroll
      for (int k = 0; k < S; k++) { double t = 0.0; for (int q = 0; q < NW; q++) t += scr[q * (K + S) + K + k]; sm[k] = t; }
      __syncthreads();
    }
  }
  __device__ __forceinline__ void sum_max(double &a, double &b) const {   // a: sum, b: max
    if constexpr (NW == 1) { a = sum(a); b = max(b); }
    else {
      a = wsum63(a); b = wmax63(b);
      if ((threadIdx.x & 63) == 63) { s[threadIdx.x >> 6] = a; s[8 + (threadIdx.x >> 6)] = b; }
      __syncthreads();
      a = 0.0; b = 0.0;
#pragma unroll
      for (int w = 0; w < NW; w++) { a += s[w]; b = nmax(b, s[8 + w]); }
      __syncthreads();
    }
  }
};

}  // namespace


namespace {

// Doubles of LDS taken by the index arrays of the register path: row pointers of A (m + 1) and B (n + 1), permutation (n).
__host__ __device__ inline int batch_index_doubles(int n, int m) { return (m + 2 * n + 2 + 1) / 2; }
// sum of prod[a .. z) in entry order; the loads of a batch of eight are independent (one LDS latency per batch, not per entry)
__device__ __forceinline__ double row_sum(const double *prod, int a, int z) {
  double acc = 0.0;
  int k = a;
  for (; k + 8 <= z; k += 8) {
    double v[8];
#pragma unroll
    for (int b = 0; b < 8; b++) v[b] = prod[k + b];
#pragma unroll
    for (int b = 0; b < 8; b++) acc += v[b];
  }
  if (k < z) {
    double v[8];
#pragma unroll
    for (int b = 0; b < 8; b++) v[b] = k + b < z ? prod[k + b] : 0.0;
#pragma unroll
    for (int b = 0; b < 8; b++) acc += v[b];
  }
  return acc;
}

// hi + lo += a * v in double-double: the product exactly (FMA), the sum by TwoSum.  Contraction is switched off for this body:
// a fused a * v + hi would break the error terms.
__device__ __forceinline__ void dd_acc(double a, double v, double &hi, double &lo) {
#pragma clang fp contract(off)
  const double pr = a * v;
  const double pe = __builtin_fma(a, v, -pr);
  const double sm_ = hi + pr, bv = sm_ - hi;
  lo += ((hi - (sm_ - bv)) + (pr - bv)) + pe;
  hi = sm_;
}

// kBB = threads per problem: 64 (one wave: barriers are free, reductions are pure shuffles; small problems) or 256.
// EA / EB > 0: every lane keeps EA entries of A and EB entries of B (value + column) in registers for the whole solve
// (entry k belongs to lane k % kBB); an SpMV is then  prod[k] = val * v[col]  for the lane's own entries (LDS only),
// a barrier, and one lane per row summing its segment of prod -- no matrix traffic inside the ADMM / PCG loops and a
// balanced first phase (the MPC rows have 1..13 entries).  EA = EB = 0: generic row loops reading the matrices from
// global memory (L1/L2), for patterns with more than 8 * kBB entries per matrix.
//
// DIRECT: the linear system of every ADMM iteration is solved exactly, as the reference's direct path does
// (_osqp.py:286-311), instead of by PCG.  K = P + sigma I + A' diag(rho) A is assembled in LDS under the bandwidth-reducing
// permutation prepared by the engine (column-major lower band, W = bw + 1 doubles per column), Cholesky-factorised in
// place at every rho change, and each solve is two substitutions that keep the live window of the right-hand side in
// REGISTERS: element e of the vector lives in lane e % 64 from the step that loads it until its own pivot step (bw < 64),
// a pivot is broadcast with v_readlane, every other lane applies its one update -- no LDS traffic on the dependency chain
// except the (prefetchable) column of L.
// POLISH (direct variants): a separate instantiation that ends with the polish step, so that the plain kernel's register allocation
// is not touched by code most launches never run.
// SPEC (256 threads, n <= kBatchSpecN, no polish): the direct solve in its SPECTRAL form (engine.hpp BatchSpectral).  K(rho)^-1 = V diag(1 / (1 + (rho -
// rho_ref) lambda)) V' with V, lambda shared by the whole batch: the workgroup keeps K^-1 in REGISTERS (thread t: row t / 2, the 64 columns of half t % 2),
// rebuilds it from V at a rho update (V streamed through LDS in chunks of columns; ~20 us, no factorisation) and solves with one dense product
// (64 FMAs per thread against broadcast LDS reads + one lane exchange): the 240-pivot substitution chain of the banded form (9 of an iteration's 14 us)
// becomes ~0.5 us, and no band lives in LDS.
// SPW: workgroups per CU the spectral instantiation is compiled for.  1: K^-1 and everything else in registers (370 of the 512 a lone workgroup may
// use) -- 5.0 us per ADMM iteration, the form for batches that fit the chip in two rounds (latency: 256 QPs 1.0 ms, 512 QPs 1.5 ms).  2: the 256
// registers of two resident workgroups, ~150 of the kernel's values in scratch -- 8 us per iteration, but twice the problems in flight: 4096 QPs
// 8.4 ms against 10.1 (and 10.6 for the banded form).  batch_solve picks by batch size.
template <int kBB, int EA, int EB, bool DIRECT, bool POLISH = false, bool N128 = false, bool SPEC = false, int SPW = 1>
__global__ __launch_bounds__(kBB, (DIRECT && kBB == 256 && EA <= 8) ? (SPEC ? SPW : 2) : 1) void k_batch_admm(BatchParams P) {
  static_assert(!SPEC || (DIRECT && !POLISH && kBB == 256), "the spectral form: 256 threads, direct, no polish");
#ifdef OSQP_HIP_KTRACE
  // diagnostic build: 100 MHz clock ticks spent in the phases; reported in rec[5..7] INSTEAD of rho / rho_updates / pcg_iters
  unsigned long long tk_all = wall_clock64(), tk_fact = 0, tk_solve = 0, tk0 = 0, tk1 = 0, tk_rhs = 0, tk_upd = 0, tk_fwd = 0, tk_res = 0;
#define BT_BEGIN() (tk0 = wall_clock64())
#define BT_END(acc) (acc += wall_clock64() - tk0)
#define BT2_BEGIN() (tk1 = wall_clock64())
#define BT2_END(acc) (acc += wall_clock64() - tk1)
#else
#define BT_BEGIN() ((void)0)
#define BT_END(acc) ((void)0)
#define BT2_BEGIN() ((void)0)
#define BT2_END(acc) ((void)0)
#endif
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = P.n, m = P.m, tid = threadIdx.x;
  if ((int)blockIdx.x >= P.nbatch) return;
  const int b = P.order ? P.order[blockIdx.x] : (int)blockIdx.x;      // (workgroups are dispatched in index order: expected-longest problems first)
  if (!SPEC && P.only_marked && P.rec[(size_t)b * kBatchRec] != kBatchUnsolved) return;      // (second launch behind the spectral one: only what that one left)
  // per-problem matrices (BatchParams::mat_on): this workgroup's copy of the parameter block points at ITS problem's scaled values, equilibration and
  // products -- everything below reads them through the same names as the shared case (block-uniform: scalar registers)
  if (P.mat_on) {
    P.A.val = P.Aval_b + (size_t)b * P.A.nnz; P.B.val = P.Bval_b + (size_t)b * P.B.nnz;
    P.D = P.D_b + (size_t)b * P.n; P.Dinv = P.Dinv_b + (size_t)b * P.n; P.E = P.E_b + (size_t)b * P.m; P.Einv = P.Einv_b + (size_t)b * P.m;
    P.c = P.c_b[b]; P.cinv = 1.0 / P.c; P.kp_val = P.kp_val_b + (size_t)b * P.nprod;
  }
  // ---- LDS carve ----
  double *x = sm, *xs = x + n, *r = xs + n, *zv = r + n, *p = zv + n, *Kp = p + n, *q = Kp + n, *Minv = q + n, *dx = Minv + n, *tn = dx + n;
  double *z = tn + n, *y = z + m, *t = y + m, *l = t + m, *u = l + m, *rho = u + m, *zt = rho + m, *dy = zt + m;
  Red<kBB / 64> red{dy + m};
  double *prod = dy + m + 16;                       // max(nnzA, nnzB) products (register path only)
  const DevCsr &A = P.A, &B = P.B;
  // DIRECT: band factor.  Column c occupies Lb[c W .. c W + bw] (W = bw + kBatchNB: kBatchNB zeros of padding per column, and
  // kBatchNB zeros in front of column 0), for n rounded up to a multiple of kBatchNB columns, + 64 doubles of read slack.
  const int bw = P.bw, W = P.bw + kBatchNB, n8 = (n + kBatchNB - 1) / kBatchNB * kBatchNB;
  // register path: the row pointers of A and B and the direct variant's permutation live in LDS too (they are read in every
  // SpMV / solve of every iteration: an L2 round trip each time otherwise)
  const int prod_len = ((A.nnz > B.nnz ? A.nnz : B.nnz) + 1) & ~1;
  int *rpA = reinterpret_cast<int *>(prod + prod_len), *rpB = rpA + (m + 1), *perm_l = rpB + (n + 1);
  double *Lb = prod + prod_len + batch_index_doubles(n, m) + kBatchNB;
  double *dinv = zv, *wbuf = r;                     // DIRECT: 1/D and the permuted right-hand side / solution reuse PCG vectors
  constexpr bool kReg = EA > 0;
  double aA[EA > 0 ? EA : 1], aB[EB > 0 ? EB : 1];
  int cA[EA > 0 ? EA : 1], cB[EB > 0 ? EB : 1];
  if constexpr (kReg) {
#pragma unroll
    for (int e = 0; e < EA; e++) { const int k = tid + e * kBB; const bool ok = k < A.nnz; aA[e] = ok ? A.val[k] : 0.0; cA[e] = ok ? A.col[k] : 0; }
#pragma unroll
    for (int e = 0; e < EB; e++) { const int k = tid + e * kBB; const bool ok = k < B.nnz; aB[e] = ok ? B.val[k] : 0.0; cB[e] = ok ? B.col[k] : 0; }
  }
  // out_i = f(i, (A v)_i)  for every row i of A
  auto applyA = [&](const double *v, auto &&f) {
    if constexpr (kReg) {
#pragma unroll
      for (int e = 0; e < EA; e++) { const int k = tid + e * kBB; if (k < A.nnz) prod[k] = aA[e] * v[cA[e]]; }
      __syncthreads();
      for (int i = tid; i < m; i += kBB) f(i, row_sum(prod, rpA[i], rpA[i + 1]));
      __syncthreads();
    } else {
      for (int i = tid; i < m; i += kBB) { double a = 0.0; for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) a += A.val[k] * v[A.col[k]]; f(i, a); }
      __syncthreads();
    }
  };
  // f(j, (B [pn; pm])_j) for every row j of B; pn == nullptr drops the P + sigma I part, pm == nullptr the A' part
  auto applyB = [&](const double *pn, const double *pm, auto &&f) {
    if constexpr (kReg) {
#pragma unroll
      for (int e = 0; e < EB; e++) {
        const int k = tid + e * kBB;
        if (k < B.nnz) { const int c = cB[e]; prod[k] = c < n ? (pn ? aB[e] * pn[c] : 0.0) : (pm ? aB[e] * pm[c - n] : 0.0); }
      }
      __syncthreads();
      for (int j = tid; j < n; j += kBB) f(j, row_sum(prod, rpB[j], rpB[j + 1]));
      __syncthreads();
    } else {
      for (int j = tid; j < n; j += kBB) {
        double a = 0.0;
        for (int k = B.rowptr[j]; k < B.rowptr[j + 1]; k++) { const int c = B.col[k]; a += c < n ? (pn ? B.val[k] * pn[c] : 0.0) : (pm ? B.val[k] * pm[c - n] : 0.0); }
        f(j, a);
      }
      __syncthreads();
    }
  };
  if constexpr (kReg) {
    for (int i = tid; i <= m; i += kBB) rpA[i] = A.rowptr[i];
    for (int j = tid; j <= n; j += kBB) rpB[j] = B.rowptr[j];
    if constexpr (DIRECT) { for (int j = tid; j < n; j += kBB) perm_l[j] = P.perm[j]; }
  }
  // ---- load the problem ----
  // inputs arrive UNSCALED; the scaling of update_lin_cost / update_bounds / warm_start (_osqp.py:1328, :1357-1358, :1505-1506)
  // is applied here:  q <- c D q,  l,u <- E clamp(l,u),  x <- Dinv x,  y <- c Einv y
  for (int j = tid; j < n; j += kBB) {
    q[j] = P.c * P.D[j] * (P.q ? P.q[(size_t)b * n + j] : P.q0[j]);
    x[j] = P.warm ? P.x[(size_t)b * n + j] * P.Dinv[j] : 0.0; dx[j] = 0.0;
  }
  int n_ineq_local = 0;
  for (int i = tid; i < m; i += kBB) {
    const double li = P.E[i] * fmax(P.l ? P.l[(size_t)b * m + i] : P.l0[i], -OSQP_INFTY), ui = P.E[i] * fmin(P.u ? P.u[(size_t)b * m + i] : P.u0[i], OSQP_INFTY);
    l[i] = li; u[i] = ui; y[i] = P.warm ? P.y[(size_t)b * m + i] * P.Einv[i] * P.c : 0.0; dy[i] = 0.0;
    int ty = (li < -OSQP_INFTY * 1e-4 && ui > OSQP_INFTY * 1e-4) ? -1 : ((ui - li < 1e-4) ? 1 : 0);   // _osqp.py:505-518
    if (!P.rho_is_vec) ty = 0;
    n_ineq_local += (ty == 0);
  }
  __syncthreads();
  const double n_ineq = red.sum((double)n_ineq_local);
  const double eqf = (n_ineq == 0.0) ? 1e3 : (DIRECT ? P.eq_factor_direct : P.eq_factor);  // engine.cpp classify_constraints()
  double rho_bar = P.rho0;
  // SPEC: K^-1 of this problem, and the LDS the form needs (in the band's place): the zero-padded right-hand side, 1 / (1 + delta lambda)
  // SPEC: K^-1 in the register layout of the f64 matrix instruction's result (v_mfma_f64_16x16x4: lane l, register r of a 16 x 16 tile = element
  // (l / 16 + 4 r, l % 16); tools/mfma_f64_layout.hip).  Wave w owns rows 32 w .. 32 w + 31: tiles kacc[tr * 8 + tc], tr = 0, 1 (16 rows each), tc = 0 .. 7.
  typedef double v4d __attribute__((ext_vector_type(4)));
  [[maybe_unused]] v4d kacc[SPEC ? 16 : 1];
  [[maybe_unused]] double *sp_rhs = Lb, *sp_dk = Lb + kBatchSpecN + 2;
  if constexpr (SPEC) {
    // V was built for ONE set of constraint classes: a problem whose own bounds give other classes (or another equality weight) is not ours
    double mism = (eqf != P.sp_eqf) ? 1.0 : 0.0;
    for (int i = tid; i < m; i += kBB) {
      const double li = l[i], ui = u[i];
      int ty = (li < -OSQP_INFTY * 1e-4 && ui > OSQP_INFTY * 1e-4) ? -1 : ((ui - li < 1e-4) ? 1 : 0);
      if (!P.rho_is_vec) ty = 0;
      if (ty != P.sp_ctype[i]) mism = 1.0;
    }
    mism = red.sum(mism);
    if (mism != 0.0) { if (tid == 0) P.rec[(size_t)b * kBatchRec] = kBatchUnsolved; return; }
  }
  // SPEC: K^-1 <- V diag(dk) V',  dk = 1 / (1 + (rb - rho_ref) lambda): V (column-major, kBatchSpecN rows per column) streamed through `prod` in chunks
  [[maybe_unused]] auto update_kinv = [&](double rb) {
    if constexpr (SPEC) {
      BT_BEGIN();
      if (P.sp_K0 && rb == P.sp_K0_rho) {                      // the batch's starting rho: the host has formed this K^-1 once for everybody, in this layout
#pragma unroll
        for (int t = 0; t < 16; t++) {
#pragma unroll
          for (int r = 0; r < 4; r++) kacc[t][r] = P.sp_K0[(size_t)(t * 4 + r) * kBB + tid];
        }
        __syncthreads();
        BT_END(tk_fact);
        return;
      }
      const double dl = rb - P.sp_rho_ref;
      if (tid < kBatchSpecN) sp_dk[tid] = 1.0 / (1.0 + dl * P.sp_lam[tid]);
#pragma unroll
      for (int t = 0; t < 16; t++) kacc[t] = v4d{0.0, 0.0, 0.0, 0.0};
      // K^-1 = (V diag(dk)) V' as a 128 x 128 x 128 product on the matrix cores: per step of four columns of V a wave issues 16 v_mfma_f64_16x16x4
      // (its two row tiles against the eight column tiles); both operands are "V(16 t + l % 16, k + l / 16)" -- ten LDS reads per step.
      const int lj = tid & 15, lk = (tid >> 4) & 3, wv = tid >> 6;
      constexpr int kCol = kBatchSpecN + 2;                     // a staged column: rows 0..63, one double of padding, rows 64..127
      constexpr int kChMax = 16;
      const int CH = min(prod_len / kCol, kChMax) & ~3;         // columns of V per chunk: a multiple of four (batch_solve requires >= 4)
      constexpr int kStage = kChMax * kBatchSpecN / kBB;
      double st[kStage];                                        // the chunk after this one, fetched while this one is consumed
      auto fetch = [&](int k0) {
        const int nk = min(CH, n - k0);
#pragma unroll
        for (int t = 0; t < kStage; t++) { const int e = tid + t * kBB; st[t] = (k0 < n && e < nk * kBatchSpecN) ? P.sp_V[(size_t)k0 * kBatchSpecN + e] : 0.0; }
      };
      const int ra0 = 32 * wv + lj, ra1 = ra0 + 16;
      const int oa0 = ra0 + (ra0 >= 64), oa1 = ra1 + (ra1 >= 64);
      fetch(0);
      for (int k0 = 0; k0 < n; k0 += CH) {
        const int nk = min(CH, n - k0), nk4 = (nk + 3) & ~3;    // (columns nk .. nk4 are staged as zeros)
        __syncthreads();
#pragma unroll
        for (int t = 0; t < kStage; t++) { const int e = tid + t * kBB; if (e < nk4 * kBatchSpecN) { const int kk = e / kBatchSpecN, r_ = e % kBatchSpecN; prod[kk * kCol + r_ + (r_ >= 64)] = st[t]; } }
        __syncthreads();
        fetch(k0 + CH);
        for (int ks = 0; ks < nk4; ks += 4) {
          const double *vc = prod + (ks + lk) * kCol;
          const double dkv = sp_dk[min(k0 + ks + lk, kBatchSpecN - 1)];
          double bb[8];
#pragma unroll
          for (int tc = 0; tc < 8; tc++) bb[tc] = vc[16 * tc + lj + (tc >= 4)];
          const double a0 = vc[oa0] * dkv, a1 = vc[oa1] * dkv;
#pragma unroll
          for (int tc = 0; tc < 8; tc++) {
            kacc[tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bb[tc], kacc[tc], 0, 0, 0);
            kacc[8 + tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bb[tc], kacc[8 + tc], 0, 0, 0);
          }
        }
      }
      __syncthreads();
      BT_END(tk_fact);
    }
  };
  // DIRECT: assemble  K = (P + sigma I) + shift I + A' diag(rho) A  in the band and factorise it (shift = 0 for the ADMM system;
  // the polish step factorises P + delta I + A_act' (1/delta) A_act with shift = delta - sigma and rho = the active-row weights)
  [[maybe_unused]] auto factorize = [&](double shift) {
    BT_BEGIN();
    // ---- assemble K (lower band, permuted) ----
    for (int s_ = tid - kBatchNB; s_ < n8 * W + 64; s_ += kBB) Lb[s_] = 0.0;
    __syncthreads();
    for (int k = tid; k < B.nnz; k += kBB) { const int s_ = P.bp_slot[k]; if (s_ >= 0) Lb[s_] = B.val[k]; }      // P + sigma I
    __syncthreads();
    if (shift != 0.0) { for (int c = tid; c < n; c += kBB) Lb[c * W] += shift; __syncthreads(); }
    for (int e = tid; e < P.nents; e += kBB) {                                                              // + A' rho A
      double acc = 0.0;
      for (int q_ = P.ke_ptr[e]; q_ < P.ke_ptr[e + 1]; q_++) acc += rho[P.kp_row[q_]] * P.kp_val[q_];
      Lb[P.ke_slot[e]] += acc;
    }
    __syncthreads();
    // ---- banded Cholesky, right-looking: column c, then the (bw x bw)/2 trailing update spread over the wave ----
    for (int c = 0; c < n; c++) {
      const double di = 1.0 / sqrt(Lb[c * W]);
      const int kmax = min(bw, n - 1 - c);
      const bool mine = tid >= 1 && tid <= kmax;
      double v = 0.0;
      if (mine) v = Lb[c * W + tid] * di;
      __syncthreads();
      if (mine) Lb[c * W + tid] = v;
      if (tid == 0) dinv[c] = di;
      __syncthreads();
      for (int t_ = tid; t_ < P.ntri; t_ += kBB) {
        const int ab = P.tri[t_], a = ab & 255, b_ = ab >> 8;
        if (b_ <= kmax) Lb[(c + a) * W + (b_ - a)] -= Lb[c * W + b_] * Lb[c * W + a];
      }
      __syncthreads();
    }
    // K = L L' = L^ D L^' with unit-lower L^ = L diag(1/L_jj), D = diag(L_jj^2): the substitutions then carry no
    // division or pivot scaling on their dependency chain.  Lb <- L^ (strictly lower part), dinv <- 1/D.
    for (int s_ = tid; s_ < n * W; s_ += kBB) { const int c = s_ / W, k = s_ - c * W; if (k >= 1 && k <= bw) Lb[s_] *= dinv[c]; }
    __syncthreads();
    for (int c = tid; c < n; c += kBB) { const double di = dinv[c]; dinv[c] = di * di; Lb[c * W] = 0.0; }   // (diagonal slots read as L^ = 0)
    __syncthreads();
    BT_END(tk_fact);
  };
  auto set_rho = [&](double rb) {
    for (int i = tid; i < m; i += kBB) {
      const double li = l[i], ui = u[i];
      int ty = (li < -OSQP_INFTY * 1e-4 && ui > OSQP_INFTY * 1e-4) ? -1 : ((ui - li < 1e-4) ? 1 : 0);
      if (!P.rho_is_vec) ty = 0;
      rho[i] = ty == -1 ? 1e-6 : (ty == 1 ? eqf * rb : rb);                                       // _osqp.py:520-522
    }
    __syncthreads();
    if constexpr (SPEC) update_kinv(rb);
    else if constexpr (DIRECT) factorize(0.0);
    else {
      for (int j = tid; j < n; j += kBB) {                   // Jacobi preconditioner = 1/diag(K)
        double sacc = 0.0, dg = 0.0;
        for (int k = B.rowptr[j]; k < B.rowptr[j + 1]; k++) { const int c = B.col[k]; const double a = B.val[k]; if (c == j) dg = a; if (c >= n) sacc += rho[c - n] * a * a; }
        Minv[j] = P.precond ? 1.0 / (dg + sacc) : 1.0;
      }
      __syncthreads();
    }
  };
  // DIRECT: out = K^-1 rhs  (both in the caller's variable order):  L^ v = P rhs ;  g = D^-1 v ;  L^' x = g ;  out = P' x.
  // Substitutions on ONE wave, kBatchNB pivots per block.  Element e lives in lane e % 64 while it is within 64 of the
  // pivots.  Per block a lane fetches its kBatchNB entries of L^ with plain strided LDS reads one block AHEAD (the padded
  // band makes every out-of-band read a zero, lanes beyond the block's reach are masked once), then for each pivot:
  // v_readlane broadcast + one FMA.  Lanes whose element has pivoted store it and continue with the element 64 further on,
  // already waiting in a register.  ~8 instructions per pivot, no LDS access, branch or division on the dependency chain.
  // What a pivot costs is the broadcast itself (tools/lane_bcast_bench.hip, one wave: v_readlane_b32 ~14 cycles whether or not it
  // is on a dependency chain -> 43.5 cycles per pivot for the two halves + FMA; DPP row_newbcast 30.6 but only inside a row of 16;
  // an LDS round trip 178 per 8 values): resolving a block of 8 through its inverted diagonal block (two rounds of 8 INDEPENDENT
  // broadcasts instead of a chain of 8) was tried and is slower, 12.6 vs 9.0 us per solve -- twice the broadcasts, and they do
  // not pipeline.
  auto ksolve = [&](const double *rhs, double *out) {
    if constexpr (SPEC) {
      if (tid < kBatchSpecN) sp_rhs[tid + (tid >= 64)] = tid < n ? rhs[tid] : 0.0;
      __syncthreads();
      BT2_BEGIN();
      // out = K^-1 rhs in the tile layout: a lane multiplies its 64 elements by the right-hand side entries of ITS column (l % 16 of each of the eight
      // column tiles: eight LDS reads, one line per wave instruction), sums over the tiles per row, and the sixteen lanes of a DPP row -- the sixteen
      // columns of a tile -- are summed with row-local DPP moves.  Fixed order, no atomics.
      const int lj = tid & 15, lk = (tid >> 4) & 3, wv = tid >> 6;
      double bb[8];
#pragma unroll
      for (int tc = 0; tc < 8; tc++) bb[tc] = sp_rhs[16 * tc + lj + (tc >= 4)];
      double pv[8];                                             // pv[tr * 4 + r]: this lane's share of row 32 wv + 16 tr + lk + 4 r
#pragma unroll
      for (int tr = 0; tr < 2; tr++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          double a = kacc[tr * 8][r] * bb[0];
#pragma unroll
          for (int tc = 1; tc < 8; tc++) a = fma(kacc[tr * 8 + tc][r], bb[tc], a);
          pv[tr * 4 + r] = a;
        }
      }
      // eight values per lane, each to be summed over the 16 lanes of its DPP row: two halving exchanges (lane ^ 1, lane ^ 2: a lane keeps the values
      // whose index has its own low bits and hands the others over -- 4 + 2 exchanges instead of 8 + 8), then two rotations by 4 and 8 lanes on the two
      // values left.  Lane lj < 4 ends with the complete sums of rows  16 i + lk + 4 lj,  i = 0, 1.
      const bool b0 = tid & 1, b1 = tid & 2;
      double w4[4], u2[2];
#pragma unroll
      for (int i = 0; i < 4; i++) { const double keep = b0 ? pv[2 * i + 1] : pv[2 * i], send = b0 ? pv[2 * i] : pv[2 * i + 1]; w4[i] = keep + bdpp<0xb1>(send); }
#pragma unroll
      for (int i = 0; i < 2; i++) { const double keep = b1 ? w4[2 * i + 1] : w4[2 * i], send = b1 ? w4[2 * i] : w4[2 * i + 1]; u2[i] = keep + bdpp<0x4e>(send); }
#pragma unroll
      for (int i = 0; i < 2; i++) { u2[i] += bdpp<0x124>(u2[i]); u2[i] += bdpp<0x128>(u2[i]); }      // row_ror:4, row_ror:8
      if (lj < 4) {
#pragma unroll
        for (int i = 0; i < 2; i++) { const int row = 32 * wv + 16 * i + lk + 4 * lj; if (row < n) out[row] = u2[i]; }
      }
      BT2_END(tk_fwd);
      __syncthreads();
      return;
    }
    const double *__restrict__ Lr = Lb;
    double *__restrict__ buf = wbuf;
    constexpr int NB = kBatchNB;
    // N128 (n <= 128, chosen by the host for the 256-thread kernels): a lane's (at most) two elements e = tid, tid + 64 stay in
    // registers from the right-hand side to the solution -- no store / refill of finished elements (an LDS round trip behind the
    // block's eight factor reads on the in-order LDS counter), no barrier between the passes, the D^-1 scaling in registers.
    // Same operations in the same order as the general form below: bit-identical results.  Two things the compiler has to be
    // told: not to unroll the block loop (it then keeps every block's LDS addresses and masks live: 80 spilled VGPRs), and not to
    // sink the next block's factor reads below the chain they are meant to overlap (without an LDS write in the loop nothing
    // stops it): 286 k QP/s without the pin, 327 k with it, 310 k for the general form.
    if constexpr (N128) {
#define KSOLVE_PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
      BT2_BEGIN();
      if (tid < 64) {
        const int e0 = tid, e1 = tid + 64;
        double cur = e0 < n ? rhs[perm_l[e0]] : 0.0, nxt = e1 < n ? rhs[perm_l[e1]] : 0.0;
        const double di0 = e0 < n ? dinv[e0] : 0.0, di1 = e1 < n ? dinv[e1] : 0.0;
        double v0 = 0.0, v1 = 0.0;
        const int nblk = n8 / NB;
        {
          auto fetch = [&](int p0, double (&l)[NB]) {
            const int dl = (tid - p0) & 63;
            const bool act = dl < bw + NB && p0 < n8;
            const double *col = act ? Lr + p0 * W + dl : Lr - 1;
            const int stride = act ? W - 1 : 0;
#pragma unroll
            for (int q = 0; q < NB; q++) l[q] = col[q * stride];
          };
          auto block = [&](int p0, const double (&l)[NB]) {
#pragma unroll
            for (int q = 0; q < NB; q++) { const double vq = readlane_f64(cur, (p0 + q) & 63); cur -= l[q] * vq; }
            const bool piv = ((tid - p0) & 63) < NB, lo = p0 < 64;   // pivoted in this block: final (blocks never straddle element 64)
            v0 = (piv && lo) ? cur : v0; v1 = (piv && !lo) ? cur : v1;
            cur = piv ? nxt : cur;
          };
          double la[NB], lb[NB];
          fetch(0, la);
#pragma clang loop unroll(disable)
          for (int b = 0; b < nblk; b += 2) {
            fetch((b + 1) * NB, lb); KSOLVE_PIN();
            block(b * NB, la);
            if (b + 1 < nblk) { fetch((b + 2) * NB, la); KSOLVE_PIN(); block((b + 1) * NB, lb); }
          }
        }
        BT2_END(tk_fwd);
        v0 *= di0; v1 *= di1;                                     // g = D^-1 v
        {
          const bool two = e1 <= n8 - 1;                          // this lane holds an element of the upper half
          cur = two ? v1 : v0; nxt = two ? v0 : 0.0;
          double x0 = 0.0, x1 = 0.0;
          auto fetch = [&](int top, double (&l)[NB]) {
            const int dl = (top - tid) & 63, i = top - dl;
            const bool act = dl < bw + NB && i >= 0 && top >= 0;
            const double *row = act ? Lr + i * W + dl : Lr - 1;
            const int stride = act ? 1 : 0;
#pragma unroll
            for (int q = 0; q < NB; q++) l[q] = row[-q * stride];
          };
          auto block = [&](int top, const double (&l)[NB]) {
#pragma unroll
            for (int q = 0; q < NB; q++) { const double xq = readlane_f64(cur, (top - q) & 63); cur -= l[q] * xq; }
            const bool piv = ((top - tid) & 63) < NB, hi = top >= 64;
            x1 = (piv && hi) ? cur : x1; x0 = (piv && !hi) ? cur : x0;
            cur = piv ? nxt : cur;
          };
          double la[NB], lb[NB];
          fetch(n8 - 1, la);
#pragma clang loop unroll(disable)
          for (int b = nblk - 1; b >= 0; b -= 2) {
            fetch(b * NB - 1, lb); KSOLVE_PIN();
            block(b * NB + NB - 1, la);
            if (b >= 1) { fetch(b * NB - NB - 1, la); KSOLVE_PIN(); block(b * NB - 1, lb); }
          }
          if (e0 < n) out[perm_l[e0]] = x0;
          if (e1 < n) out[perm_l[e1]] = x1;
        }
      }
      __syncthreads();
      return;
#undef KSOLVE_PIN
    }
    for (int k = tid; k < n; k += kBB) buf[k] = rhs[perm_l[k]];
    __syncthreads();
    const int nblk = n8 / NB;
    const bool w0 = tid < 64;                                   // the substitutions run on wave 0; other waves wait at the barriers
    BT2_BEGIN();
    // ---- forward, unit lower:  v_e = w_e - sum_{j in [e-bw, e)} L^[e][j] v_j ;  L^[p0 + dl][p0 + q] = Lr[(p0 + q) W + dl - q] ----
    if (w0) {
      double cur = tid < n ? buf[tid] : 0.0, nxt = 64 + tid < n ? buf[64 + tid] : 0.0;
      // lanes beyond the block's reach read the zero in front of column 0 eight times (stride 0): no masking after the load
      auto fetch = [&](int p0, double (&l)[NB]) {
        const int dl = (tid - p0) & 63;
        const bool act = dl < bw + NB && p0 < n8;
        const double *col = act ? Lr + p0 * W + dl : Lr - 1;
        const int stride = act ? W - 1 : 0;
#pragma unroll
        for (int q = 0; q < NB; q++) l[q] = col[q * stride];
      };
      auto block = [&](int p0, const double (&l)[NB]) {
#pragma unroll
        for (int q = 0; q < NB; q++) { const double vq = readlane_f64(cur, (p0 + q) & 63); cur -= l[q] * vq; }
        const int dl = (tid - p0) & 63;
        if (dl < NB) {                                          // pivoted in this block: final
          const int e = p0 + dl;
          if (e < n) buf[e] = cur;
          cur = nxt; nxt = e + 128 < n ? buf[e + 128] : 0.0;   // (requesting this before the chain was tried: the compiler then drains the LDS counter in front of the chain, +17 %)
        }
      };
      double la[NB], lb[NB];                                    // two blocks in flight, roles alternate (no register rotation)
      fetch(0, la);
      for (int b = 0; b < nblk; b += 2) {
        fetch((b + 1) * NB, lb);
        block(b * NB, la);
        if (b + 1 < nblk) { fetch((b + 2) * NB, la); block((b + 1) * NB, lb); }
      }
    }
    __syncthreads();
    BT2_END(tk_fwd);
    for (int k = tid; k < n; k += kBB) buf[k] *= dinv[k];                          // g = D^-1 v
    __syncthreads();
    // ---- backward, unit upper (L^'):  x_i = g_i - sum_{j in (i, i+bw]} L^[j][i] x_j ; blocks from the top, pivots top - q ;
    //      lane's element i = top - dl ;  L^[top - q][i] = Lr[i W + dl - q] ----
    if (w0) {
      auto elem = [&](int top) { return top - ((top - tid) & 63); };
      const int i0 = elem(n8 - 1);
      double cur = (i0 >= 0 && i0 < n) ? buf[i0] : 0.0, nxt = i0 - 64 >= 0 ? buf[i0 - 64] : 0.0;
      auto fetch = [&](int top, double (&l)[NB]) {
        const int dl = (top - tid) & 63, i = top - dl;
        const bool act = dl < bw + NB && i >= 0 && top >= 0;
        const double *row = act ? Lr + i * W + dl : Lr - 1;
        const int stride = act ? 1 : 0;
#pragma unroll
        for (int q = 0; q < NB; q++) l[q] = row[-q * stride];
      };
      auto block = [&](int top, const double (&l)[NB]) {
#pragma unroll
        for (int q = 0; q < NB; q++) { const double xq = readlane_f64(cur, (top - q) & 63); cur -= l[q] * xq; }
        const int dl = (top - tid) & 63;
        if (dl < NB) {
          const int i = top - dl;
          if (i < n) buf[i] = cur;
          cur = nxt; nxt = i - 128 >= 0 ? buf[i - 128] : 0.0;
        }
      };
      double la[NB], lb[NB];
      fetch(n8 - 1, la);
      for (int b = nblk - 1; b >= 0; b -= 2) {
        fetch(b * NB - 1, lb);
        block(b * NB + NB - 1, la);
        if (b >= 1) { fetch(b * NB - NB - 1, la); block(b * NB - 1, lb); }
      }
    }
    __syncthreads();
    for (int k = tid; k < n; k += kBB) out[perm_l[k]] = buf[k];
    __syncthreads();
  };
  set_rho(rho_bar);
  // y = A v   (thread per row of A)
  auto spmv_A = [&](const double *v, double *out, bool times_rho) {
    for (int i = tid; i < m; i += kBB) {
      double a = 0.0;
      for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) a += A.val[k] * v[A.col[k]];
      out[i] = times_rho ? rho[i] * a : a;
    }
    __syncthreads();
  };
  // z = A x, zt = A xs (xs = x)   (_osqp.py:1509 / cold start)
  for (int j = tid; j < n; j += kBB) xs[j] = x[j];
  __syncthreads();
  spmv_A(x, z, false);
  for (int i = tid; i < m; i += kBB) { zt[i] = z[i]; if (P.zs && P.warm) z[i] = P.zs[(size_t)b * m + i]; }      // (a continued solve keeps its z iterate)
  __syncthreads();

  // residuals of the current (x, z, y): returns through references; all threads hold identical values
  double pri_u, ax_u, z_u, pri_s, ax_s, z_s, dy_u, dy_s, pinf_lhs, dua_u, px_u, aty_u, dua_s, px_s, aty_s, dxn_u, dxn_s, xpx, qx, qdx, qn_s, qn_u;
  auto residuals = [&]() {
    double a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0, s1 = 0;
    for (int i = tid; i < m; i += kBB) {
      double ax = 0.0;
      for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) ax += A.val[k] * x[A.col[k]];
      const double pr = ax - z[i], ei = P.Einv[i], dyi = dy[i];
      a1 = nmax(a1, fabs(ei * pr)); a2 = nmax(a2, fabs(ei * ax)); a3 = nmax(a3, fabs(ei * z[i]));
      a4 = nmax(a4, fabs(pr)); a5 = nmax(a5, fabs(ax)); a6 = nmax(a6, fabs(z[i]));
      a7 = nmax(a7, fabs(P.E[i] * dyi)); a8 = nmax(a8, fabs(dyi));
      s1 += u[i] * fmax(dyi, 0.0) + l[i] * fmin(dyi, 0.0);
    }
    { double g1[4] = {a1, a2, a3, a4}, g2[4] = {a5, a6, a7, a8}, g3[1] = {s1}, none[1] = {0.0};
      red.template max_sum_n<4, 0>(g1, none); red.template max_sum_n<4, 0>(g2, none); red.template max_sum_n<0, 1>(none, g3);
      pri_u = g1[0]; ax_u = g1[1]; z_u = g1[2]; pri_s = g1[3]; ax_s = g2[0]; z_s = g2[1]; dy_u = g2[2]; dy_s = g2[3]; pinf_lhs = g3[0]; }
    double b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0, b8 = 0, b9 = 0, b10 = 0, t1 = 0, t2 = 0, t3 = 0;
    for (int j = tid; j < n; j += kBB) {
      double sp = 0.0, sa = 0.0;
      for (int k = B.rowptr[j]; k < B.rowptr[j + 1]; k++) { const int c = B.col[k]; if (c < n) sp += B.val[k] * x[c]; else sa += B.val[k] * y[c - n]; }
      const double px = sp - P.sigma * x[j], dr = px + q[j] + sa, di = P.Dinv[j];
      b1 = nmax(b1, fabs(di * dr)); b2 = nmax(b2, fabs(di * px)); b3 = nmax(b3, fabs(di * sa));
      b4 = nmax(b4, fabs(dr)); b5 = nmax(b5, fabs(px)); b6 = nmax(b6, fabs(sa));
      b7 = nmax(b7, fabs(P.D[j] * dx[j])); b8 = nmax(b8, fabs(dx[j])); b9 = nmax(b9, fabs(q[j])); b10 = nmax(b10, fabs(di * q[j]));
      t1 += x[j] * px; t2 += q[j] * x[j]; t3 += q[j] * dx[j];
    }
    { double g1[4] = {b1, b2, b3, b4}, g2[4] = {b5, b6, b7, b8}, g3[2] = {b9, b10}, sm[2] = {t1, t2}, g4[1] = {t3}, none[1] = {0.0};
      red.template max_sum_n<4, 0>(g1, none); red.template max_sum_n<4, 0>(g2, none); red.template max_sum_n<2, 2>(g3, sm); red.template max_sum_n<0, 1>(none, g4);
      dua_u = g1[0]; px_u = g1[1]; aty_u = g1[2]; dua_s = g1[3]; px_s = g2[0]; aty_s = g2[1]; dxn_u = g2[2]; dxn_s = g2[3]; qn_s = g3[0]; qn_u = g3[1];
      xpx = sm[0]; qx = sm[1]; qdx = g4[0]; }
  };

  int status = OSQP_UNSOLVED, iter = 0, rho_updates = 0;
  long pcg_total = 0;
  double obj = 0, prim_res = 0, dual_res = 0;
  residuals();
  double eps_cg = P.cg_frac * dua_s, eps_prev = INFINITY;
  bool rel_rule = !(eps_cg > 1e-13) || !isfinite(eps_cg);
  const bool unsc = P.unscaled != 0;

  while (true) {
    iter++;
    // ---- rhs = sigma x - q + A'(rho z - y);  r = rhs - K xs with K xs = B[xs; rho zt]   (_osqp.py:649-650) ----
    [[maybe_unused]] double rz_l = 0, rn_l = 0, bn_l = 0;
    BT2_BEGIN();
    for (int i = tid; i < m; i += kBB) t[i] = rho[i] * z[i] - y[i];
    __syncthreads();
    applyB(nullptr, t, [&](int j, double sA) { Kp[j] = P.sigma * x[j] - q[j] + sA; });          // Kp holds rhs for a moment
    BT2_END(tk_rhs);
    if constexpr (DIRECT) {
      BT_BEGIN();
      ksolve(Kp, xs);                                                                    // x~ = K^-1 rhs   (_osqp.py:307-311, reduced form)
      BT_END(tk_solve);
    } else {
      for (int i = tid; i < m; i += kBB) t[i] = rho[i] * zt[i];
      __syncthreads();
      applyB(xs, t, [&](int j, double sK) {
        const double rhs = Kp[j], rr = rhs - sK, zz = Minv[j] * rr;
        r[j] = rr; zv[j] = zz; p[j] = zz;
        rz_l += rr * zz; rn_l = nmax(rn_l, fabs(rr)); bn_l = nmax(bn_l, fabs(rhs));
      });
      double rz = rz_l, rn = rn_l;
      red.sum_max(rz, rn);
      const double bn = red.max(bn_l);
      const double tol = rel_rule ? fmax(0.1 * bn, 1e-13) : fmax(1e-14 * bn, eps_cg);
      // ---- PCG on K = P + sigma I + A' diag(rho) A ----
      for (int it = 0; it < P.cg_max && rn > tol; it++) {
        applyA(p, [&](int i, double a) { t[i] = rho[i] * a; });              // t = rho .* (A p)
        double pkp_l = 0.0;
        applyB(p, t, [&](int j, double a) { Kp[j] = a; pkp_l += a * p[j]; });
        const double pkp = red.sum(pkp_l);
        const double al = rz / pkp;
        double rz2 = 0.0, rn2 = 0.0;
        for (int j = tid; j < n; j += kBB) {
          xs[j] += al * p[j];
          const double rr = r[j] - al * Kp[j], zz = Minv[j] * rr;
          r[j] = rr; zv[j] = zz;
          rz2 += rr * zz; rn2 = nmax(rn2, fabs(rr));
        }
        red.sum_max(rz2, rn2);
        const double be = rz2 / rz;
        rz = rz2; rn = rn2;
        for (int j = tid; j < n; j += kBB) p[j] = zv[j] + be * p[j];
        __syncthreads();
        pcg_total++;
      }
    }
    // ---- z~ = A xs; x, z, y update (_osqp.py:660-703) ----
    BT2_BEGIN();
    applyA(xs, [&](int i, double a) {
      const double rh = rho[i], yi = y[i];
      const double zr = P.alpha * a + (1.0 - P.alpha) * z[i];
      const double zn = fmin(fmax(zr + yi / rh, l[i]), u[i]);
      const double dyi = rh * (zr - zn);
      y[i] = yi + dyi; dy[i] = dyi; z[i] = zn; zt[i] = a;
    });
    for (int j = tid; j < n; j += kBB) { const double xo = x[j], xn = P.alpha * xs[j] + (1.0 - P.alpha) * xo; dx[j] = xn - xo; x[j] = xn; }
    __syncthreads();
    BT2_END(tk_upd);

    const bool at_check = (P.check > 0 && iter % P.check == 0) || iter >= P.max_iter;
    const bool at_rho = P.rho_interval > 0 && iter % P.rho_interval == 0;
    if (!at_check && !at_rho) continue;
    BT2_BEGIN();
    residuals();
    BT2_END(tk_res);
    obj = (0.5 * xpx + qx) * (P.scaling ? P.cinv : 1.0);                               // _osqp.py:705-712
    prim_res = m == 0 ? 0.0 : (unsc ? pri_u : pri_s);
    dual_res = unsc ? P.cinv * dua_u : dua_s;
    bool stop = false;
    for (int approx = 0; approx < 2 && !stop && at_check; approx++) {                  // _osqp.py:998-1077, :1264-1266
      if (approx && iter < P.max_iter) break;
      const double f = approx ? 10.0 : 1.0;
      const double ea = f * P.eps_abs, er = f * P.eps_rel, epi = f * P.eps_pinf, edi = f * P.eps_dinf;
      if (prim_res > OSQP_INFTY || dual_res > OSQP_INFTY || prim_res != prim_res || dual_res != dual_res) { status = OSQP_NON_CVX; obj = NAN; stop = true; break; }
      bool pri_ok = false, dua_ok = false, pinf = false, dinf = false;
      if (m == 0) pri_ok = true;
      else if (prim_res < ea + er * (unsc ? fmax(ax_u, z_u) : fmax(ax_s, z_s))) pri_ok = true;
      else {                                                                          // is_primal_infeasible :796-820
        const double nd = unsc ? dy_u : dy_s;
        if (nd > epi && pinf_lhs < -epi * nd) {
          double mu = 0, ms = 0;
          for (int j = tid; j < n; j += kBB) {
            double sa = 0.0;
            for (int k = B.rowptr[j]; k < B.rowptr[j + 1]; k++) { const int c = B.col[k]; if (c >= n) sa += B.val[k] * dy[c - n]; }
            mu = nmax(mu, fabs(P.Dinv[j] * sa)); ms = nmax(ms, fabs(sa));
          }
          mu = red.max(mu); ms = red.max(ms);
          pinf = (unsc ? mu : ms) < epi * nd;
        }
      }
      const double mx = unsc ? P.cinv * fmax(fmax(aty_u, px_u), qn_u) : fmax(fmax(aty_s, px_s), qn_s);
      if (dual_res < ea + er * mx) dua_ok = true;
      else {                                                                          // is_dual_infeasible :822-878
        const double nd = unsc ? dxn_u : dxn_s, sc = unsc ? P.c : 1.0;
        if (nd > edi && qdx < -sc * edi * nd) {
          double mu = 0, ms = 0, viol = 0;
          for (int j = tid; j < n; j += kBB) {
            double sp = 0.0;
            for (int k = B.rowptr[j]; k < B.rowptr[j + 1]; k++) { const int c = B.col[k]; if (c < n) sp += B.val[k] * dx[c]; }
            sp -= P.sigma * dx[j];
            mu = nmax(mu, fabs(P.Dinv[j] * sp)); ms = nmax(ms, fabs(sp));
          }
          mu = red.max(mu); ms = red.max(ms);
          if ((unsc ? mu : ms) < sc * edi * nd) {
            for (int i = tid; i < m; i += kBB) {
              double a = 0.0;
              for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) a += A.val[k] * dx[A.col[k]];
              if (unsc) a *= P.Einv[i];
              if ((u[i] < OSQP_INFTY * 1e-4 && a > edi * nd) || (l[i] > -OSQP_INFTY * 1e-4 && a < -edi * nd)) viol += 1.0;
            }
            viol = red.sum(viol);
            dinf = viol == 0.0;
          }
        }
      }
      if (pri_ok && dua_ok) { status = approx ? OSQP_SOLVED_INACCURATE : OSQP_SOLVED; stop = true; }
      else if (pinf) { status = approx ? OSQP_PRIMAL_INFEASIBLE_INACCURATE : OSQP_PRIMAL_INFEASIBLE; obj = OSQP_INFTY; stop = true; }
      else if (dinf) { status = approx ? OSQP_DUAL_INFEASIBLE_INACCURATE : OSQP_DUAL_INFEASIBLE; obj = -OSQP_INFTY; stop = true; }
    }
    if (stop) break;
    if (iter >= P.max_iter) { status = OSQP_MAX_ITER_REACHED; break; }
    if (at_rho) {                                                                      // adapt_rho :880-930
      const double pr = pri_s / (fmax(ax_s, z_s) + 1e-10), du = dua_s / (fmax(fmax(aty_s, px_s), qn_s) + 1e-10);
      double rn_ = rho_bar * sqrt(pr / (du + 1e-10));
      rn_ = fmin(fmax(rn_, 1e-6), 1e6);
      if (rn_ > P.rho_tol * rho_bar || rn_ < rho_bar / P.rho_tol) { rho_bar = rn_; set_rho(rho_bar); rho_updates++; }
    }
    double e2 = P.cg_frac * dua_s;                                                     // inner tolerance: engine.cpp solve()
    if (m == 0) e2 = P.cg_frac * dua_s;
    e2 = fmax(fmin(e2, eps_prev), 1e-13);
    if (isfinite(e2)) { eps_prev = e2; eps_cg = e2; rel_rule = false; }
  }
  // ---- polish (_osqp.py:1710-1828), direct variants only: the reference's algorithm on the factor already in LDS ----
  // Active rows guessed from the scaled (z, y) (:1719-1720); the regularised reduced KKT system
  //     [P + delta I, A_act'; A_act, -delta I] [dx; dy] = [r1; r2]
  // is solved through its Schur complement  (P + delta I + A_act' A_act / delta) dx = r1 + A_act' r2 / delta,  dy = (A_act dx - r2) / delta,
  // ONE banded factorisation (factorize: weights 1/delta on active rows, 0 elsewhere, diagonal shift delta - sigma), and the
  // first solve plus polish_refine_iter refinement steps (:1692-1708) all have the same form: with (x, y) = 0 at the start,
  //     t = y - (b - A_act x) / delta,   rhs = -q - P x - A_act' t,   dx = K^-1 rhs,   x += dx,   y = t + A_act dx / delta.
  // Then z = A x, the normal-cone projection (:1773-1780) and the accept test on the residuals (:1786-1793).
  // rho estimate of the ADMM point (_osqp.py:1275, :880-908), before any polish
  const double rho_est = fmin(fmax(rho_bar * sqrt((pri_s / (fmax(ax_s, z_s) + 1e-10)) / (dua_s / (fmax(fmax(aty_s, px_s), qn_s) + 1e-10) + 1e-10)), 1e-6), 1e6);
  int status_polish = 0;
  [[maybe_unused]] unsigned long long pol_ticks = 0;
  if constexpr (DIRECT && POLISH) {
    if (status == OSQP_SOLVED) {
      const unsigned long long tp0 = wall_clock64();
      const double idel = 1.0 / P.delta;
      // ADMM point kept in p (x), zt (z), dy (y): none of them is live in the direct variant after the loop
      for (int j = tid; j < n; j += kBB) { p[j] = x[j]; x[j] = 0.0; }
      for (int i = tid; i < m; i += kBB) {
        const double zi = z[i], yi = y[i];
        zt[i] = zi; dy[i] = yi;
        const bool low = zi - l[i] < -yi, upp = !low && (u[i] - zi < yi);      // (a row active on both sides enters once, at its lower bound)
        rho[i] = (low || upp) ? idel : 0.0;
        z[i] = low ? l[i] : u[i];                                              // b_i of an active row (unused otherwise)
        y[i] = 0.0;
      }
      __syncthreads();
      factorize(P.delta - P.sigma);
      // The Schur complement amplifies an error of A x - b by 1/delta: the constraint residual of the current x is accumulated in
      // double-double (exact products by FMA, compensated sums), the classic higher-precision residual of iterative refinement,
      // and y advances by the correction of the step (t + A dx / delta) -- its noise floor is eps |y|, not eps |A||x| / delta.
      for (int k = 0; k <= P.refine; k++) {
        for (int i = tid; i < m; i += kBB) {
          double tv = 0.0;
          if (rho[i] != 0.0) {
            double hi = -z[i], lo = 0.0;                                        // (A x)_i - b_i
            for (int e = A.rowptr[i]; e < A.rowptr[i + 1]; e++) {
              const double a = A.val[e], v = x[A.col[e]];
              dd_acc(a, v, hi, lo);
            }
            tv = y[i] + (hi + lo) * idel;                                       // y - r2 / delta,  r2 = b - A x
          }
          t[i] = tv;
        }
        __syncthreads();
        applyB(x, t, [&](int j, double s_) { Kp[j] = -q[j] + P.sigma * x[j] - s_; });       // (B carries P + sigma I)
        ksolve(Kp, Minv);
        for (int j = tid; j < n; j += kBB) x[j] += Minv[j];
        __syncthreads();
        applyA(Minv, [&](int i, double adx) { if (rho[i] != 0.0) y[i] = t[i] + adx * idel; });
      }
      applyA(x, [&](int i, double ax) { const double tmp = ax + y[i], zc = fmin(fmax(tmp, l[i]), u[i]); z[i] = zc; y[i] = tmp - zc; });
      const double pri0 = prim_res, dua0 = dual_res;
      residuals();
      const double ppri = m == 0 ? 0.0 : (unsc ? pri_u : pri_s), pdua = unsc ? P.cinv * dua_u : dua_s;
      const bool ok = (ppri < pri0 && pdua < dua0) || (ppri < pri0 && dua0 < 1e-10) || (pdua < dua0 && pri0 < 1e-10);
      if (ok) { obj = (0.5 * xpx + qx) * (P.scaling ? P.cinv : 1.0); prim_res = ppri; dual_res = pdua; status_polish = 1; }
      else {
        status_polish = -1;
        for (int j = tid; j < n; j += kBB) x[j] = p[j];
        for (int i = tid; i < m; i += kBB) { z[i] = zt[i]; y[i] = dy[i]; }
      }
      __syncthreads();
      pol_ticks = wall_clock64() - tp0;
    }
  }
  // ---- store: x = D x, y = cinv E y (_osqp.py:1110-1112); certificates in place of x / y for infeasible problems ----
  const bool pinf = status == OSQP_PRIMAL_INFEASIBLE || status == OSQP_PRIMAL_INFEASIBLE_INACCURATE;
  const bool dinf = status == OSQP_DUAL_INFEASIBLE || status == OSQP_DUAL_INFEASIBLE_INACCURATE;
  for (int j = tid; j < n; j += kBB) P.x[(size_t)b * n + j] = dinf ? (unsc ? P.D[j] * dx[j] : dx[j]) : (pinf ? NAN : (P.scaling ? P.D[j] * x[j] : x[j]));
  if (P.zs) for (int i = tid; i < m; i += kBB) P.zs[(size_t)b * m + i] = z[i];
  for (int i = tid; i < m; i += kBB) P.y[(size_t)b * m + i] = pinf ? (unsc ? P.E[i] * dy[i] : dy[i]) : (dinf ? NAN : (P.scaling ? P.cinv * P.E[i] * y[i] : y[i]));
  if (tid == 0) {
    double *rc = P.rec + (size_t)b * kBatchRec;
    rc[0] = status; rc[1] = iter; rc[2] = obj; rc[3] = prim_res; rc[4] = dual_res; rc[5] = rho_bar; rc[6] = rho_updates; rc[7] = (double)pcg_total;
    rc[8] = status_polish; rc[9] = 1e-8 * (double)pol_ticks;      // (100 MHz wall clock -> seconds)
    rc[10] = rho_est; rc[11] = 0.0;
    if (P.iters_out) P.iters_out[b] = iter;
#ifdef OSQP_HIP_KTRACE
    rc[5] = (double)tk_fact; rc[6] = (double)tk_solve; rc[7] = (double)(wall_clock64() - tk_all);
    rc[3] = (double)tk_rhs; rc[4] = (double)tk_upd; rc[8] = (double)tk_fwd; rc[9] = (double)tk_res;
#endif
  }
}


// ---------------------------------------------------------------------------------------------------------------- one WAVE per problem, spectral form
// k_batch_admm gives a problem a whole workgroup -- and, in its spectral form, the whole register file of a CU for K^-1: ONE problem in flight per CU,
// 3.7 us per ADMM iteration of which the dense product is 0.6 (profiles/r06_batch_trace.txt).  Here a problem is ONE WAVE and a CU runs kBatchWaveW of
// them at once, with what they share kept once:
//   * V (the batch's common eigenvectors, engine.hpp BatchSpectral) in LDS, row-major with an odd row stride -- 116 KB at n = 120.  A solve is
//     x~ = V (d . (V' rhs)),  d_k = 1 / (1 + (rho - rho_ref) lambda_k):  two passes over V by the wave itself (lane k reads column k of row j for V' rhs --
//     consecutive words; lane i reads row i for V t -- stride 121 doubles: conflict-free), the vector entries broadcast with v_readlane from the lane
//     that owns them.  No K^-1 exists: a rho update is two divisions per lane (the workgroup kernel rebuilds K^-1 from V: ~20 us).
//   * the matrices' values once in LDS as well, in ELL steps with one lane per row (BatchParams::wv_*: groups of 64 rows, A's rows sorted by length so
//     that a step's rows are about equally long -- 26 KB for the MPC pattern); every wave keeps its lanes' column indices in registers (two per
//     register) and gathers through one staged vector of its own in LDS.  (Values in registers -- 112 more per lane -- was the first form: 185 of them
//     spilled at eight waves per CU.)
//   * every iterate in registers (n <= 128: two slots per lane, m <= 256: four); reductions are DPP moves + v_readlane.  No barrier after the prologue:
//     a wave takes the next position of the launch order from a device counter when its problem is done (longest-expected problems first, as before).
// Same arithmetic as k_batch_admm's direct path up to the order of the sums (FMA chains per row here, products summed in entry order there; the two-stage
// solve instead of K^-1 times rhs): iteration counts agree, iterates to ~1e-12 (tests/test_gpu_batch_wave.py).
#ifdef OSQP_HIP_KTRACE
#define WT_MARK(v) const unsigned long long v = wall_clock64()
#define WT_ADD(acc, d) (acc += (d))
#else
#define WT_MARK(v) ((void)0)
#define WT_ADD(acc, d) ((void)0)
#endif
template <int N8>
__global__ __launch_bounds__(64 * kBatchWaveW, 1) void k_batch_wave(BatchParams P) {
  static_assert(N8 % 8 == 0 && N8 <= kBatchSpecN, "rows / columns of V in LDS (n rounded up; zero beyond n): compile-time -- the dense products are straight-line code");
  constexpr int S = N8 + 1;                            // row stride: odd (conflict-free column walks)
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = P.n, m = P.m, tid = threadIdx.x, L = tid & 63, wv = tid >> 6;
  constexpr int n8 = N8;                               // (>= n: batch_solve picks the instantiation)
  double *Vl = sm;
  const int stg_len = ((n > m ? n : m) + 1) & ~1;
  const int ae0 = P.wv_aend[0], ae1 = P.wv_aend[1], ae2 = P.wv_aend[2], ae3 = P.wv_aend[3], te0 = P.wv_tend[0], te1 = P.wv_tend[1];
  double *eA = sm + (((size_t)n8 * S + 1) & ~(size_t)1), *eT = eA + (size_t)ae3 * 64;
  double *stage = eT + (size_t)te1 * 64 + (size_t)wv * stg_len;
  unsigned short *kA = reinterpret_cast<unsigned short *>(eT + (size_t)te1 * 64 + (size_t)kBatchWaveW * stg_len), *kT = kA + (size_t)ae3 * 64;
  for (int e = tid; e < n8 * S; e += 64 * kBatchWaveW) Vl[e] = 0.0;
  for (int e = tid; e < ae3 * 64; e += 64 * kBatchWaveW) { const int ix = P.wv_Aidx[e]; eA[e] = ix >= 0 ? P.A.val[ix] : 0.0; kA[e] = (unsigned short)P.wv_Acol[e]; }
  for (int e = tid; e < te1 * 64; e += 64 * kBatchWaveW) { const int ix = P.wv_Tidx[e]; eT[e] = ix >= 0 ? P.B.val[ix] : 0.0; kT[e] = (unsigned short)P.wv_Tcol[e]; }
  __syncthreads();
  for (int e = tid; e < n * kBatchSpecN; e += 64 * kBatchWaveW) { const int k = e / kBatchSpecN, j = e % kBatchSpecN; if (j < n) Vl[j * S + k] = P.sp_V[e]; }      // V(j, k)
  __syncthreads();                                       // (the last barrier of the kernel)
  const DevCsr &B = P.B;
  // ---- the matrices: ELL values and columns (16 bits) in LDS, one copy for the eight waves ----
  int rowm[4]; bool vm[4], vn[2];
#pragma unroll
  for (int s = 0; s < 4; s++) { rowm[s] = (s * 64 + L < 256) ? P.wv_row[s * 64 + L] : -1; vm[s] = rowm[s] >= 0; if (!vm[s]) rowm[s] = 0; }
#pragma unroll
  for (int s = 0; s < 2; s++) vn[s] = s * 64 + L < n;
  const int jn[2] = {min(L, n - 1), min(64 + L, n - 1)};          // (clamped: lanes without an element read a valid one and drop the result)
  const double lam0 = P.sp_lam[jn[0]], lam1 = P.sp_lam[jn[1]];
  auto wave_sync = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
  auto wsum = [&](double v) { return readlane_f64(wsum63(v), 63); };
  auto wmax = [&](double v) { return readlane_f64(wmax63(v), 63); };
  auto stage_n = [&](const double (&v)[2]) {
    wave_sync();
#pragma unroll
    for (int s = 0; s < 2; s++) if (vn[s]) stage[s * 64 + L] = v[s];
    wave_sync();
  };
  auto stage_m = [&](const double (&v)[4]) {
    wave_sync();
#pragma unroll
    for (int s = 0; s < 4; s++) if (vm[s]) stage[rowm[s]] = v[s];
    wave_sync();
  };
  // one group of 64 rows: steps [s0, s1) of an ELL array (values ev, 16-bit columns kv; a lane's entries beyond its row are padding: value 0, column 0)
  // against the vector staged in LDS.  A runtime loop per group with nothing but loads and one FMA chain in it (first form: one unrolled sequence over
  // all groups with the group picked by uniform branches per step -- the compiler turned that into a branch and two dependent LDS round trips per step,
  // 6 of an iteration's 12 us).
  auto ell_group = [&](const double *ev, const unsigned short *kv, int s0, int s1) {
    // (blocks of four steps: four column reads and four value reads, then the four gathers, then the FMAs -- two LDS round trips per block; written as a
    //  plain loop the compiler serialises column read -> gather -> FMA per step.  The steps of a last, partial block re-read the group's last step with
    //  the value replaced by zero.)
    double acc = 0.0;
    for (int b0 = s0; b0 < s1; b0 += 4) {
      int c[4]; double v[4], pv[4];
#pragma unroll
      for (int j = 0; j < 4; j++) { const bool on = b0 + j < s1; const int si = on ? b0 + j : s1 - 1; c[j] = kv[si * 64 + L]; v[j] = ev[si * 64 + L]; v[j] = on ? v[j] : 0.0; }
#pragma unroll
      for (int j = 0; j < 4; j++) pv[j] = stage[c[j]];
#pragma unroll
      for (int j = 0; j < 4; j++) acc = fma(v[j], pv[j], acc);
    }
    return acc;
  };
  // out[slot] = (A v)_row for the n-vector v staged in LDS
  auto mulA = [&](double (&out)[4]) { out[0] = ell_group(eA, kA, 0, ae0); out[1] = ell_group(eA, kA, ae0, ae1); out[2] = ell_group(eA, kA, ae1, ae2); out[3] = ell_group(eA, kA, ae2, ae3); };
  // out[slot] = (A' w)_j for the m-vector w staged in LDS (original row numbering)
  auto mulT = [&](double (&out)[2]) { out[0] = ell_group(eT, kT, 0, te0); out[1] = ell_group(eT, kT, te0, te1); };
  // out[slot] = ((P + sigma I) v)_j for the n-vector staged in LDS: B's entries with column < n, from memory (residuals and certificates only)
  auto mulP = [&](double (&out)[2]) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
      double a = 0.0;
      if (vn[s]) { const int j = s * 64 + L; for (int k = B.rowptr[j]; k < B.rowptr[j + 1]; k++) { const int c = B.col[k]; if (c < n) a = fma(B.val[k], stage[c], a); } }
      out[s] = a;
    }
  };
  // The two dense products of a solve, x~ = V (d . (V' rhs)), fully unrolled, straight-line code (the size n8 of the LDS copy is a template parameter): every
  // LDS offset and every v_readlane lane index is then an immediate -- per step one or two reads, two v_readlane, two FMAs.  (First form: a runtime loop,
  // indices in registers: a v_add per read, an s_add + wait states per v_readlane -- 2 250 instructions per solve, 10 of an iteration's 23 us with eight
  // waves per CU.  Rows / columns n .. n8 of the LDS copy are zero.)
  // (Every block of eight steps sits behind a uniform branch on an OPAQUE copy of n8: the branch is never taken, but it keeps the compiler from treating a
  //  product as one straight line -- it then hoists all 240 reads and the v_readlanes in front of the FMAs and spills 500 - 1 000 registers, with or
  //  without scheduling barriers between hand-pipelined blocks: 12.5 ms per batch instead of 4.9.)
  // w = V' rhs: lane = column k (and k + 64), step = row j;  rhs_j from lane j % 64 of rhs[j / 64]
  auto vt_mul = [&](const double (&rhs)[2], double &o0, double &o1) {
    const double *vc = Vl + L;
    int n8r = N8;
    asm volatile("" : "+s"(n8r));
    double a0[2] = {0.0, 0.0}, a1[2] = {0.0, 0.0};
    double va[8], vb[8], na[8], nb[8];
    auto load = [&](int j0, double (&ua)[8], double (&ub)[8]) {
#pragma unroll
      for (int jj = 0; jj < 8; jj++) { ua[jj] = vc[(j0 + jj) * S]; ub[jj] = vc[(j0 + jj) * S + 64]; }
    };
    auto compute = [&](int j0, const double (&ua)[8], const double (&ub)[8]) {
#pragma unroll
      for (int jj = 0; jj < 8; jj++) { const double r = readlane_f64(rhs[j0 >> 6], (j0 & 63) + jj); a0[jj & 1] = fma(ua[jj], r, a0[jj & 1]); a1[jj & 1] = fma(ub[jj], r, a1[jj & 1]); }
    };
    load(0, va, vb);
#pragma unroll
    for (int j0 = 0; j0 < N8; j0 += 16) {
      if (j0 < n8r) { if (j0 + 8 < N8) load(j0 + 8, na, nb); compute(j0, va, vb); }            // (the block's reads are the NEXT block's: one LDS latency behind eight steps of FMAs)
      if (j0 + 8 < N8 && j0 + 8 < n8r) { if (j0 + 16 < N8) load(j0 + 16, va, vb); compute(j0 + 8, na, nb); }
    }
    o0 = a0[0] + a0[1]; o1 = a1[0] + a1[1];
  };
  // x = V t: lane = row i (and i + 64), step = column k
  auto v_mul = [&](double t0, double t1, double &o0, double &o1) {
    const double *vr0 = Vl + jn[0] * S, *vr1 = Vl + jn[1] * S;
    int n8r = N8;
    asm volatile("" : "+s"(n8r));
    double a0[2] = {0.0, 0.0}, a1[2] = {0.0, 0.0};
    double va[8], vb[8], na[8], nb[8];
    auto load = [&](int k0, double (&ua)[8], double (&ub)[8]) {
#pragma unroll
      for (int kk = 0; kk < 8; kk++) { ua[kk] = vr0[k0 + kk]; ub[kk] = vr1[k0 + kk]; }
    };
    auto compute = [&](int k0, const double (&ua)[8], const double (&ub)[8]) {
#pragma unroll
      for (int kk = 0; kk < 8; kk++) { const double r = readlane_f64(k0 < 64 ? t0 : t1, (k0 & 63) + kk); a0[kk & 1] = fma(ua[kk], r, a0[kk & 1]); a1[kk & 1] = fma(ub[kk], r, a1[kk & 1]); }
    };
    load(0, va, vb);
#pragma unroll
    for (int k0 = 0; k0 < N8; k0 += 16) {
      if (k0 < n8r) { if (k0 + 8 < N8) load(k0 + 8, na, nb); compute(k0, va, vb); }
      if (k0 + 8 < N8 && k0 + 8 < n8r) { if (k0 + 16 < N8) load(k0 + 16, va, vb); compute(k0 + 8, na, nb); }
    }
    o0 = a0[0] + a0[1]; o1 = a1[0] + a1[1];
  };
  auto ksolve = [&](const double (&rhs)[2], double dk0, double dk1, double (&out)[2]) {
    double w0, w1;
    vt_mul(rhs, w0, w1);
    v_mul(vn[0] ? dk0 * w0 : 0.0, vn[1] ? dk1 * w1 : 0.0, out[0], out[1]);
  };
  const bool unsc = P.unscaled != 0;
  bool first = true;

  while (true) {
    // ---- the next problem of the launch order ----
    // The first problem of a wave is position (wave index) x (workgroups) + (workgroup): the longest-expected problems land one per CU.  Later ones
    // come from the device counter.  (EVERY lane adds one -- the compiler folds that into a single atomic per wave -- and the counter runs in units of
    //  64.  The obvious form, lane 0 alone behind `if (L == 0)`, hangs: the loop is then compiled with a second version for the lanes that never run the
    //  atomic, whose position stays 0.)
    int pos;
    if (first) { pos = P.wv_first + __builtin_amdgcn_readfirstlane(wv) * (int)gridDim.x + (int)blockIdx.x; first = false; }
    else pos = P.wv_first + (int)gridDim.x * kBatchWaveW + (__builtin_amdgcn_readfirstlane(atomicAdd(P.wv_queue, 1)) >> 6);
    if (pos >= P.nbatch) break;
    const int b = P.order ? P.order[pos] : pos;
    double x[2], q[2], dx[2], xs[2], z[4], y[4], l[4], u[4], dy[4];
    int ty[4];
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const int j = jn[s];
      q[s] = vn[s] ? P.c * P.D[j] * (P.q ? P.q[(size_t)b * n + j] : P.q0[j]) : 0.0;
      x[s] = (vn[s] && P.warm) ? P.x[(size_t)b * n + j] * P.Dinv[j] : 0.0; dx[s] = 0.0; xs[s] = 0.0;
    }
    double n_ineq_l = 0.0, mism = 0.0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int i = rowm[s];
      const double li = P.E[i] * fmax(P.l ? P.l[(size_t)b * m + i] : P.l0[i], -OSQP_INFTY), ui = P.E[i] * fmin(P.u ? P.u[(size_t)b * m + i] : P.u0[i], OSQP_INFTY);
      l[s] = vm[s] ? li : 0.0; u[s] = vm[s] ? ui : 0.0;
      y[s] = (vm[s] && P.warm) ? P.y[(size_t)b * m + i] * P.Einv[i] * P.c : 0.0; dy[s] = 0.0; z[s] = 0.0;
      int t_ = (li < -OSQP_INFTY * 1e-4 && ui > OSQP_INFTY * 1e-4) ? -1 : ((ui - li < 1e-4) ? 1 : 0);   // _osqp.py:505-518
      if (!P.rho_is_vec) t_ = 0;
      ty[s] = t_;
      if (vm[s]) { n_ineq_l += (t_ == 0); if (t_ != P.sp_ctype[i]) mism = 1.0; }
    }
    const double n_ineq = wsum(n_ineq_l);
    const double eqf = (n_ineq == 0.0) ? 1e3 : P.eq_factor_direct;                      // engine.cpp classify_constraints()
    // V was built for ONE set of constraint classes: a problem whose own bounds give other classes (or another equality weight) is the banded kernel's
    mism = wsum(mism) + (eqf != P.sp_eqf ? 1.0 : 0.0);
    if (mism != 0.0) { if (L == 0) P.rec[(size_t)b * kBatchRec] = kBatchUnsolved; continue; }
    double rho_bar = P.rho0, rho_eq = 0.0, dk0 = 0.0, dk1 = 0.0;
    auto rho_of = [&](int s) { return ty[s] == -1 ? 1e-6 : (ty[s] == 1 ? rho_eq : rho_bar); };             // _osqp.py:520-522 (three values: not kept per row)
    auto set_rho = [&](double rb) {
      rho_eq = eqf * rb;
      const double dl = rb - P.sp_rho_ref;
      dk0 = 1.0 / (1.0 + dl * lam0); dk1 = 1.0 / (1.0 + dl * lam1);
    };
    set_rho(rho_bar);
    // z = A x   (_osqp.py:1509 / cold start); a continued solve keeps its z iterate
    stage_n(x);
    mulA(z);
    if (P.zs && P.warm) {
#pragma unroll
      for (int s = 0; s < 4; s++) if (vm[s]) z[s] = P.zs[(size_t)b * m + rowm[s]];
    }
    double pri_u = 0, ax_u = 0, z_u = 0, pri_s = 0, ax_s = 0, z_s = 0, dy_u = 0, dy_s = 0, pinf_lhs = 0, dua_u = 0, px_u = 0, aty_u = 0, dua_s = 0, px_s = 0, aty_s = 0,
           dxn_u = 0, dxn_s = 0, xpx = 0, qx = 0, qdx = 0, qn_s = 0, qn_u = 0;
    auto residuals = [&]() {
      double a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0, s1 = 0;
      stage_n(x);
      double ax[4], sp[2], sa[2];
      mulA(ax); mulP(sp);
#pragma unroll
      for (int s = 0; s < 4; s++) if (vm[s]) {
        const int i = rowm[s];
        const double pr = ax[s] - z[s], ei = P.Einv[i], dyi = dy[s];
        a1 = nmax(a1, fabs(ei * pr)); a2 = nmax(a2, fabs(ei * ax[s])); a3 = nmax(a3, fabs(ei * z[s]));
        a4 = nmax(a4, fabs(pr)); a5 = nmax(a5, fabs(ax[s])); a6 = nmax(a6, fabs(z[s]));
        a7 = nmax(a7, fabs(P.E[i] * dyi)); a8 = nmax(a8, fabs(dyi));
        s1 += u[s] * fmax(dyi, 0.0) + l[s] * fmin(dyi, 0.0);
      }
      pri_u = wmax(a1); ax_u = wmax(a2); z_u = wmax(a3); pri_s = wmax(a4); ax_s = wmax(a5); z_s = wmax(a6); dy_u = wmax(a7); dy_s = wmax(a8); pinf_lhs = wsum(s1);
      stage_m(y);
      mulT(sa);
      double b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0, b8 = 0, b9 = 0, b10 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
      for (int s = 0; s < 2; s++) if (vn[s]) {
        const int j = s * 64 + L;
        const double px = sp[s] - P.sigma * x[s], dr = px + q[s] + sa[s], di = P.Dinv[j];
        b1 = nmax(b1, fabs(di * dr)); b2 = nmax(b2, fabs(di * px)); b3 = nmax(b3, fabs(di * sa[s]));
        b4 = nmax(b4, fabs(dr)); b5 = nmax(b5, fabs(px)); b6 = nmax(b6, fabs(sa[s]));
        b7 = nmax(b7, fabs(P.D[j] * dx[s])); b8 = nmax(b8, fabs(dx[s])); b9 = nmax(b9, fabs(q[s])); b10 = nmax(b10, fabs(di * q[s]));
        t1 += x[s] * px; t2 += q[s] * x[s]; t3 += q[s] * dx[s];
      }
      dua_u = wmax(b1); px_u = wmax(b2); aty_u = wmax(b3); dua_s = wmax(b4); px_s = wmax(b5); aty_s = wmax(b6); dxn_u = wmax(b7); dxn_s = wmax(b8); qn_s = wmax(b9); qn_u = wmax(b10);
      xpx = wsum(t1); qx = wsum(t2); qdx = wsum(t3);
    };
    int status = OSQP_UNSOLVED, iter = 0, rho_updates = 0;
    double obj = 0, prim_res = 0, dual_res = 0;
#ifdef OSQP_HIP_KTRACE
    unsigned long long tkT = 0, tkS = 0, tkA = 0, tkR = 0;
#endif
    WT_MARK(tk0);
    if (P.max_iter <= 0) residuals();
    while (true) {
      iter++;
      // ---- rhs = sigma x - q + A'(rho z - y);  x~ = K^-1 rhs   (_osqp.py:649-650, :307-311 in reduced form) ----
      double t[4], rhs[2], sA[2];
      WT_MARK(c0);
#pragma unroll
      for (int s = 0; s < 4; s++) t[s] = rho_of(s) * z[s] - y[s];
      stage_m(t);
      mulT(sA);
      WT_MARK(c1); WT_ADD(tkT, c1 - c0);
#pragma unroll
      for (int s = 0; s < 2; s++) rhs[s] = vn[s] ? P.sigma * x[s] - q[s] + sA[s] : 0.0;
      ksolve(rhs, dk0, dk1, xs);
      WT_MARK(c2); WT_ADD(tkS, c2 - c1);
      // ---- z~ = A x~; x, z, y update (_osqp.py:660-703) ----
      stage_n(xs);
      double zt[4];
      mulA(zt);
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const double rh = rho_of(s), yi = y[s];
        const double zr = P.alpha * zt[s] + (1.0 - P.alpha) * z[s];
        const double zn = fmin(fmax(zr + yi / rh, l[s]), u[s]);
        const double dyi = rh * (zr - zn);
        y[s] = yi + dyi; dy[s] = dyi; z[s] = zn;
      }
#pragma unroll
      for (int s = 0; s < 2; s++) { const double xo = x[s], xn = P.alpha * xs[s] + (1.0 - P.alpha) * xo; dx[s] = xn - xo; x[s] = xn; }
      WT_MARK(c3); WT_ADD(tkA, c3 - c2);
      const bool at_check = (P.check > 0 && iter % P.check == 0) || iter >= P.max_iter;
      const bool at_rho = P.rho_interval > 0 && iter % P.rho_interval == 0;
      if (!at_check && !at_rho) continue;
      WT_MARK(c4);
      residuals();
      WT_MARK(c5); WT_ADD(tkR, c5 - c4);
      obj = (0.5 * xpx + qx) * (P.scaling ? P.cinv : 1.0);                               // _osqp.py:705-712
      prim_res = m == 0 ? 0.0 : (unsc ? pri_u : pri_s);
      dual_res = unsc ? P.cinv * dua_u : dua_s;
      bool stop = false;
      for (int approx = 0; approx < 2 && !stop && at_check; approx++) {                  // _osqp.py:998-1077, :1264-1266
        if (approx && iter < P.max_iter) break;
        const double f = approx ? 10.0 : 1.0;
        const double ea = f * P.eps_abs, er = f * P.eps_rel, epi = f * P.eps_pinf, edi = f * P.eps_dinf;
        if (prim_res > OSQP_INFTY || dual_res > OSQP_INFTY || prim_res != prim_res || dual_res != dual_res) { status = OSQP_NON_CVX; obj = NAN; stop = true; break; }
        bool pri_ok = false, dua_ok = false, pinf = false, dinf = false;
        if (m == 0) pri_ok = true;
        else if (prim_res < ea + er * (unsc ? fmax(ax_u, z_u) : fmax(ax_s, z_s))) pri_ok = true;
        else {                                                                          // is_primal_infeasible :796-820
          const double nd = unsc ? dy_u : dy_s;
          if (nd > epi && pinf_lhs < -epi * nd) {
            double sa[2], mu = 0, ms = 0;
            stage_m(dy); mulT(sa);
#pragma unroll
            for (int s = 0; s < 2; s++) if (vn[s]) { mu = nmax(mu, fabs(P.Dinv[s * 64 + L] * sa[s])); ms = nmax(ms, fabs(sa[s])); }
            mu = wmax(mu); ms = wmax(ms);
            pinf = (unsc ? mu : ms) < epi * nd;
          }
        }
        const double mx = unsc ? P.cinv * fmax(fmax(aty_u, px_u), qn_u) : fmax(fmax(aty_s, px_s), qn_s);
        if (dual_res < ea + er * mx) dua_ok = true;
        else {                                                                          // is_dual_infeasible :822-878
          const double nd = unsc ? dxn_u : dxn_s, sc = unsc ? P.c : 1.0;
          if (nd > edi && qdx < -sc * edi * nd) {
            double sp[2], mu = 0, ms = 0, viol = 0;
            stage_n(dx); mulP(sp);
#pragma unroll
            for (int s = 0; s < 2; s++) if (vn[s]) { const double v = sp[s] - P.sigma * dx[s]; mu = nmax(mu, fabs(P.Dinv[s * 64 + L] * v)); ms = nmax(ms, fabs(v)); }
            mu = wmax(mu); ms = wmax(ms);
            if ((unsc ? mu : ms) < sc * edi * nd) {
              double adx[4];
              mulA(adx);
#pragma unroll
              for (int s = 0; s < 4; s++) if (vm[s]) {
                double a = adx[s];
                if (unsc) a *= P.Einv[rowm[s]];
                if ((u[s] < OSQP_INFTY * 1e-4 && a > edi * nd) || (l[s] > -OSQP_INFTY * 1e-4 && a < -edi * nd)) viol += 1.0;
              }
              viol = wsum(viol);
              dinf = viol == 0.0;
            }
          }
        }
        if (pri_ok && dua_ok) { status = approx ? OSQP_SOLVED_INACCURATE : OSQP_SOLVED; stop = true; }
        else if (pinf) { status = approx ? OSQP_PRIMAL_INFEASIBLE_INACCURATE : OSQP_PRIMAL_INFEASIBLE; obj = OSQP_INFTY; stop = true; }
        else if (dinf) { status = approx ? OSQP_DUAL_INFEASIBLE_INACCURATE : OSQP_DUAL_INFEASIBLE; obj = -OSQP_INFTY; stop = true; }
      }
      if (stop) break;
      if (iter >= P.max_iter) { status = OSQP_MAX_ITER_REACHED; break; }
      if (at_rho) {                                                                      // adapt_rho :880-930
        const double pr = pri_s / (fmax(ax_s, z_s) + 1e-10), du = dua_s / (fmax(fmax(aty_s, px_s), qn_s) + 1e-10);
        double rn_ = rho_bar * sqrt(pr / (du + 1e-10));
        rn_ = fmin(fmax(rn_, 1e-6), 1e6);
        if (rn_ > P.rho_tol * rho_bar || rn_ < rho_bar / P.rho_tol) { rho_bar = rn_; set_rho(rho_bar); rho_updates++; }
      }
    }
    const double rho_est = fmin(fmax(rho_bar * sqrt((pri_s / (fmax(ax_s, z_s) + 1e-10)) / (dua_s / (fmax(fmax(aty_s, px_s), qn_s) + 1e-10) + 1e-10)), 1e-6), 1e6);
    // ---- store: x = D x, y = cinv E y (_osqp.py:1110-1112); certificates in place of x / y for infeasible problems ----
    const bool pinf = status == OSQP_PRIMAL_INFEASIBLE || status == OSQP_PRIMAL_INFEASIBLE_INACCURATE;
    const bool dinf = status == OSQP_DUAL_INFEASIBLE || status == OSQP_DUAL_INFEASIBLE_INACCURATE;
#pragma unroll
    for (int s = 0; s < 2; s++) if (vn[s]) { const int j = s * 64 + L; P.x[(size_t)b * n + j] = dinf ? (unsc ? P.D[j] * dx[s] : dx[s]) : (pinf ? NAN : (P.scaling ? P.D[j] * x[s] : x[s])); }
#pragma unroll
    for (int s = 0; s < 4; s++) if (vm[s]) {
      const int i = rowm[s];
      if (P.zs) P.zs[(size_t)b * m + i] = z[s];
      P.y[(size_t)b * m + i] = pinf ? (unsc ? P.E[i] * dy[s] : dy[s]) : (dinf ? NAN : (P.scaling ? P.cinv * P.E[i] * y[s] : y[s]));
    }
    if (L == 0) {
      double *rc = P.rec + (size_t)b * kBatchRec;
      rc[0] = status; rc[1] = iter; rc[2] = obj; rc[3] = prim_res; rc[4] = dual_res; rc[5] = rho_bar; rc[6] = rho_updates; rc[7] = 0.0;
      rc[8] = 0.0; rc[9] = 0.0; rc[10] = rho_est; rc[11] = 0.0;
#ifdef OSQP_HIP_KTRACE
      rc[7] = (double)tkT; rc[8] = (double)tkS; rc[9] = (double)tkA; rc[11] = (double)tkR; rc[10] = (double)(wall_clock64() - tk0);      // 100 MHz ticks per phase (tools/batch_wave_probe.py)
#endif
      if (P.iters_out) P.iters_out[b] = iter;
    }
  }
}

}  // namespace

// LDS needed per problem (bytes); 0 if the problem does not fit one workgroup's LDS.  nnz > 0 adds the product buffer of
// the register-resident path.
size_t batch_lds_bytes_nnz(int n, int m, int nnz) {
  const size_t b = sizeof(double) * ((size_t)10 * n + (size_t)8 * m + 16 + (nnz > 0 ? (size_t)((nnz + 1) & ~1) + batch_index_doubles(n, m) : 0));
  return b <= 64 * 1024 ? b : 0;
}
size_t batch_lds_bytes(int n, int m) { return batch_lds_bytes_nnz(n, m, 0); }
size_t batch_direct_lds_bytes(int n, int m, int nnz, int bw) {
  if (bw < 0 || bw > kBatchDirectMaxBw) return 0;
  const size_t n8 = (size_t)(n + kBatchNB - 1) / kBatchNB * kBatchNB;
  const size_t b = sizeof(double) * ((size_t)10 * n + (size_t)8 * m + 16 + (size_t)((nnz + 1) & ~1) + batch_index_doubles(n, m) + kBatchNB + n8 * (bw + kBatchNB) + 64);
  return b <= 144 * 1024 ? b : 0;         // (above the default 64 KB dynamic-LDS limit: batch_solve raises it; gfx950 has 160 KB per CU)
}
__global__ void k_batch_products(DevCsr A, int nprod, const int *a, const int *b, double *out) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < nprod; p += gridDim.x * blockDim.x) out[p] = A.val[a[p]] * A.val[b[p]];
}
// rank of problem b among all by descending iteration count (ties by index) = its place in the launch order
// (sixteen lanes -- one DPP row -- share a problem: each compares against every sixteenth count, the row's partial ranks are summed with DPP moves.
//  nbatch / 16 workgroups instead of nbatch / 256: with one thread per problem the 4096 x 4096 comparisons ran on 16 CUs and took 177 us per call,
//  3 % of a 4096-problem batch)
__global__ __launch_bounds__(256) void k_batch_order(int nbatch, const int *iters, int *order) {
  const int b = blockIdx.x * 16 + (threadIdx.x >> 4), jl = threadIdx.x & 15;
  const int mine = b < nbatch ? iters[b] : 0;
  int rank = 0;
  for (int j = jl; j < nbatch; j += 16) { const int v = iters[j]; rank += (v > mine) || (v == mine && j < b); }
  rank += __builtin_amdgcn_update_dpp(0, rank, 0xb1, 0xf, 0xf, false);       // lane ^ 1
  rank += __builtin_amdgcn_update_dpp(0, rank, 0x4e, 0xf, 0xf, false);       // lane ^ 2
  rank += __builtin_amdgcn_update_dpp(0, rank, 0x124, 0xf, 0xf, false);      // row_ror:4
  rank += __builtin_amdgcn_update_dpp(0, rank, 0x128, 0xf, 0xf, false);      // row_ror:8
  if (jl == 0 && b < nbatch) order[rank] = b;
}
void batch_order(Dev &d, int nbatch, const int *iters, int *order, void *stream) {
  if (hipSetDevice(d.device) != hipSuccess) throw DeviceError("osqp_hip: hipSetDevice failed");
  if (nbatch > 0) hipLaunchKernelGGL(k_batch_order, dim3((nbatch + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream ? stream : d.stream), nbatch, iters, order);
}
void batch_products(Dev &d, int nprod, const int *a, const int *b, double *out) {
  if (hipSetDevice(d.device) != hipSuccess) throw DeviceError("osqp_hip: hipSetDevice failed");
  if (nprod > 0) hipLaunchKernelGGL(k_batch_products, dim3((nprod + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(d.stream), d.A, nprod, a, b, out);
}

// ---- per-problem matrices: assembly + equilibration of every problem's own (P_b, A_b), one workgroup per problem, everything in LDS.
// The same arithmetic, entry for entry, as the single-QP setup (backend_hip.hip k_asm_scatter, k_rowmax, k_ruiz_*: _osqp.py:389-497): an element of a
// per-element batch is scaled exactly as a solver set up with its matrices alone would scale it -- the reference's forward builds one solver per
// element (nn/torch.py:142-157).  Output: the scaled CSR values of A and B = [P + sigma I | A'], D, 1/D, E, 1/E, c and the banded variant's products.
__device__ __forceinline__ double batch_limit_scaling(double v) { return v < 1e-4 ? 1.0 : (v > 1e4 ? 1e4 : v); }     // _osqp.py:363-387
__global__ __launch_bounds__(256) void k_batch_prepare(BatchParams P, Dev d, const double *Px_b, const double *Ax_b, int iters) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = P.n, m = P.m, tid = threadIdx.x, b = blockIdx.x, nzA = P.A.nnz, nzB = P.B.nnz;
  if (b >= P.nbatch) return;
  double *Av = sm, *Bv = Av + ((nzA + 1) & ~1), *D = Bv + ((nzB + 1) & ~1), *E = D + n, *dt = E + m, *et = dt + n, *q = et + m, *np_ = q + n, *scr = np_ + n;
  Red<4> red{scr};
  const double *Ar = Ax_b ? Ax_b + (size_t)b * d.nzA : d.Araw, *Pr = Px_b ? Px_b + (size_t)b * d.nzP : d.Praw;
  for (int k = tid; k < nzB; k += 256) Bv[k] = 0.0;
  __syncthreads();
  for (int k = tid; k < d.nzA; k += 256) { const double v = Ar[k]; Av[d.AmA[k]] = v; Bv[d.AmB[k]] = v; }
  for (int k = tid; k < d.nzP; k += 256) {
    const int i = d.Pi[k], j = d.Pj[k]; const double v = Pr[k];
    if (i == j) atomicAdd(&Bv[d.Pm1[k]], v);            // (repeated (j, j) entries of a valid CSC sum up)
    else { Bv[d.Pm1[k]] = v; Bv[d.Pm2[k]] = v; }
  }
  for (int j = tid; j < n; j += 256) { D[j] = 1.0; q[j] = P.q ? P.q[(size_t)b * n + j] : P.q0[j]; }
  for (int i = tid; i < m; i += 256) E[i] = 1.0;
  double c = 1.0;
  __syncthreads();
  const int *Arp = P.A.rowptr, *Ac = P.A.col, *Brp = P.B.rowptr, *Bc = P.B.col;
  for (int it = 0; it < iters; it++) {
    for (int j = tid; j < n; j += 256) { double mx = 0.0; for (int k = Brp[j]; k < Brp[j + 1]; k++) mx = fmax(mx, fabs(Bv[k])); dt[j] = 1.0 / sqrt(batch_limit_scaling(mx)); }      // KKT column j = row j of [P | A']
    for (int i = tid; i < m; i += 256) { double mx = 0.0; for (int k = Arp[i]; k < Arp[i + 1]; k++) mx = fmax(mx, fabs(Av[k])); et[i] = 1.0 / sqrt(batch_limit_scaling(mx)); }
    __syncthreads();
    for (int i = tid; i < m; i += 256) { const double ei = et[i]; for (int k = Arp[i]; k < Arp[i + 1]; k++) Av[k] *= ei * dt[Ac[k]]; E[i] *= ei; }
    for (int j = tid; j < n; j += 256) {
      const double dj = dt[j];
      for (int k = Brp[j]; k < Brp[j + 1]; k++) { const int cc = Bc[k]; Bv[k] *= cc < n ? dt[cc] * dj : et[cc - n] * dj; }
      q[j] *= dj; D[j] *= dj;
    }
    __syncthreads();
    double sum = 0.0, nq = 0.0;
    for (int j = tid; j < n; j += 256) { double mx = 0.0; for (int k = Brp[j]; k < Brp[j + 1] && Bc[k] < n; k++) mx = fmax(mx, fabs(Bv[k])); sum += mx; nq = fmax(nq, fabs(q[j])); }
    red.sum_max(sum, nq);
    const double mean = sum / (double)(n > 0 ? n : 1);
    const double ct = 1.0 / batch_limit_scaling(fmax(batch_limit_scaling(nq), mean));
    c *= ct;
    for (int j = tid; j < n; j += 256) { for (int k = Brp[j]; k < Brp[j + 1] && Bc[k] < n; k++) Bv[k] *= ct; q[j] *= ct; }
    __syncthreads();
  }
  for (int j = tid; j < n; j += 256) { Bv[d.Bdiag[j]] += P.sigma; P.D_b[(size_t)b * n + j] = D[j]; P.Dinv_b[(size_t)b * n + j] = 1.0 / D[j]; }
  for (int i = tid; i < m; i += 256) { P.E_b[(size_t)b * m + i] = E[i]; P.Einv_b[(size_t)b * m + i] = 1.0 / E[i]; }
  if (tid == 0) P.c_b[b] = c;
  __syncthreads();
  for (int k = tid; k < nzA; k += 256) P.Aval_b[(size_t)b * nzA + k] = Av[k];
  for (int k = tid; k < nzB; k += 256) P.Bval_b[(size_t)b * nzB + k] = Bv[k];
  if (P.kp_val_b) for (int k = tid; k < P.nprod; k += 256) P.kp_val_b[(size_t)b * P.nprod + k] = Av[P.kp_a[k]] * Av[P.kp_b[k]];
}
int batch_prepare(Dev &d, const BatchParams &p, const double *Px_b, const double *Ax_b, int scaling_iters, void *stream) {
  if (hipSetDevice(d.device) != hipSuccess) return OSQP_ALGEBRA_LOAD_ERROR;
  const size_t lds = sizeof(double) * ((size_t)((p.A.nnz + 1) & ~1) + ((p.B.nnz + 1) & ~1) + 4 * (size_t)p.n + 2 * (size_t)p.m + 16);
  if (lds > 144 * 1024) return OSQP_FUNC_NOT_IMPLEMENTED;
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(&k_batch_prepare), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return OSQP_FUNC_NOT_IMPLEMENTED; }
  hipLaunchKernelGGL(k_batch_prepare, dim3(p.nbatch), dim3(256), lds, static_cast<hipStream_t>(stream ? stream : d.stream), p, d, Px_b, Ax_b, scaling_iters);
  return OSQP_NO_ERROR;
}

namespace {
struct BatchChoice { bool dir256, dir64, w64, w256, generic; int e64, e256; size_t lds_reg, lds_gen, lds_dir; };
BatchChoice choose_batch_variant(const BatchParams &p) {
  BatchChoice c{};
  const int mx = p.A.nnz > p.B.nnz ? p.A.nnz : p.B.nnz;
  c.lds_reg = batch_lds_bytes_nnz(p.n, p.m, mx); c.lds_gen = batch_lds_bytes(p.n, p.m);
  static const char *const names[] = {nullptr, "direct", "direct256", "w64", "w256", "generic"};      // OSQPHipPolicy::batch_variant (debugging / A-B runs)
  const char *force = (p.variant >= 1 && p.variant <= 5) ? names[p.variant] : nullptr;
  c.e64 = (mx + 63) / 64; c.e256 = (mx + 255) / 256;
  const bool can64 = c.lds_reg && c.e64 <= 24 && p.n <= 1024 && p.m <= 2048, can256 = c.lds_reg && c.e256 <= 8;
  c.lds_dir = batch_direct_lds_bytes(p.n, p.m, mx, p.bw);
  // (the direct variant with four waves also comes with 16 entries per lane: up to 4096 stored entries per matrix, one problem per CU)
  const bool can_dir = can64 && c.lds_dir && p.perm, can_dir256 = c.lds_reg && c.e256 <= 16 && c.lds_dir && p.perm;
  // default: the direct solve with four waves per problem (MPC batch: 14.2 ms; one wave 18.6 ms; PCG, one wave: 37 ms)
  c.dir256 = force ? !std::strcmp(force, "direct256") && can_dir256 : can_dir256;
  c.dir64 = !c.dir256 && (force ? !std::strcmp(force, "direct") && can_dir : can_dir);
  c.w64 = !c.dir64 && !c.dir256 && (force ? !std::strcmp(force, "w64") && can64 : can64);
  c.w256 = !c.dir64 && !c.dir256 && !c.w64 && (force ? !std::strcmp(force, "w256") && can256 : can256);
  c.generic = !c.dir64 && !c.dir256 && !c.w64 && !c.w256 && c.lds_gen;
  return c;
}
}  // namespace
void batch_release(Dev &d) {
  if (d.bside) { (void)hipStreamDestroy(static_cast<hipStream_t>(d.bside)); d.bside = nullptr; }
  if (d.bev0) { (void)hipEventDestroy(static_cast<hipEvent_t>(d.bev0)); d.bev0 = nullptr; }
  if (d.bev1) { (void)hipEventDestroy(static_cast<hipEvent_t>(d.bev1)); d.bev1 = nullptr; }
}
bool batch_direct_selected(const BatchParams &p) { const BatchChoice c = choose_batch_variant(p); return c.dir256 || c.dir64; }
// rows / columns of V in the wave kernel's LDS: compile-time, three instantiations (n <= 64: 33 KB; n <= 120: the MPC batch's 116 KB; n <= 128)
static int batch_wave_n8(int n) { return n <= 64 ? 64 : (n <= 120 ? 120 : 128); }
size_t batch_wave_lds_bytes(int n, int m, int steps) {
  if (n < 1 || n > kBatchSpecN || m < 1 || m > 256) return 0;
  const size_t n8 = (size_t)batch_wave_n8(n), S = n8 + 1, stg = (size_t)((n > m ? n : m) + 1) & ~(size_t)1;
  const size_t b = sizeof(double) * (((n8 * S + 1) & ~(size_t)1) + (size_t)steps * 64 + (size_t)kBatchWaveW * stg) + sizeof(unsigned short) * (size_t)steps * 64;
  // (V t reads row min(64 + lane, n - 1) and V' rhs reads 64 words past a row's start: both stay inside V + staging)
  return b <= 160 * 1024 ? b : 0;
}

int batch_solve(Dev &d, const BatchParams &p, void *stream) {
  if (hipSetDevice(d.device) != hipSuccess) return OSQP_ALGEBRA_LOAD_ERROR;
  hipStream_t st = static_cast<hipStream_t>(stream ? stream : d.stream);
  const BatchChoice ch = choose_batch_variant(p);
  const int e64 = ch.e64, e256 = ch.e256;
  const size_t lds_reg = ch.lds_reg, lds_gen = ch.lds_gen, lds_dir = ch.lds_dir;
  const bool use_dir256 = ch.dir256, use_dir = ch.dir64, use64 = ch.w64, use256 = ch.w256;
#define BATCH_LAUNCH(TB, E, LDS) hipLaunchKernelGGL((k_batch_admm<TB, E, E, false>), dim3(p.nbatch), dim3(TB), LDS, st, p)
#define BATCH_LAUNCH_DIRECT_P(TB, E, POL, SMALL) do { \
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_batch_admm<TB, E, E, true, POL, SMALL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dir) != hipSuccess) \
      throw DeviceError("osqp_hip: cannot reserve LDS for the direct batch kernel"); \
    hipLaunchKernelGGL((k_batch_admm<TB, E, E, true, POL, SMALL>), dim3(p.nbatch), dim3(TB), lds_dir, st, p); } while (0)
#define BATCH_LAUNCH_DIRECT_N(TB, E, SMALL) do { if (p.polish) BATCH_LAUNCH_DIRECT_P(TB, E, true, SMALL); else BATCH_LAUNCH_DIRECT_P(TB, E, false, SMALL); } while (0)
  // (256-thread kernels: n <= 128 takes the instantiation whose substitutions keep every element in registers, ksolve)
#define BATCH_LAUNCH_DIRECT(TB, E) do { if (TB == 256 && p.n <= 128) BATCH_LAUNCH_DIRECT_N(TB, E, (TB == 256)); else BATCH_LAUNCH_DIRECT_N(TB, E, false); } while (0)
  // The spectral form of the direct solve where the engine has prepared it (BatchParams::sp_V): every problem whose constraint classes are the
  // reference's is solved by this launch; the others are marked and left to the banded kernel launched right behind (only_marked).
  bool spectral = false;
  const int prod_len = ((p.A.nnz > p.B.nnz ? p.A.nnz : p.B.nnz) + 1) & ~1;
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d.device);
  const bool spec_ok = use_dir256 && p.sp_V && !p.mat_on && !p.polish && p.n <= kBatchSpecN && e256 <= 8 && prod_len >= 4 * (kBatchSpecN + 2) && !p.only_marked;
  // workgroup-per-problem spectral launch of the first q.nbatch positions of the launch order, on stream s; false: the device refused the LDS reservation
  // (the banded launch below then takes the whole batch)
  auto spec_launch = [&](const BatchParams &q, hipStream_t s) -> bool {
    const size_t lds_spec = lds_reg + sizeof(double) * (kBatchNB + 2 * kBatchSpecN + 4);
    bool ok = true;
#define BATCH_LAUNCH_SPEC_W(E, W) do { \
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_batch_admm<256, E, E, true, false, false, true, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_spec) != hipSuccess) { (void)hipGetLastError(); ok = false; } \
    else hipLaunchKernelGGL((k_batch_admm<256, E, E, true, false, false, true, W>), dim3(q.nbatch), dim3(256), lds_spec, s, q); } while (0)
    // One workgroup per CU (everything in registers) at every batch size: since K^-1 lives in the matrix instruction's result registers the two-per-CU
    // form (256 registers, scratch) no longer wins on large batches either -- 4096 QPs 5.9 ms against 6.2 ms.  OSQP_HIP_BATCH_WIDE_ROUNDS=r selects it for
    // batches of more than r rounds of one workgroup per CU (A/B runs).
    static const int wide_rounds = std::getenv("OSQP_HIP_BATCH_WIDE_ROUNDS") ? std::atoi(std::getenv("OSQP_HIP_BATCH_WIDE_ROUNDS")) : (1 << 20);
    const bool wide = q.nbatch > wide_rounds * cus;
#define BATCH_LAUNCH_SPEC(E) do { if (wide) BATCH_LAUNCH_SPEC_W(E, 2); else BATCH_LAUNCH_SPEC_W(E, 1); } while (0)
    if (e256 <= 2) BATCH_LAUNCH_SPEC(2); else if (e256 <= 4) BATCH_LAUNCH_SPEC(4); else if (e256 <= 6) BATCH_LAUNCH_SPEC(6); else BATCH_LAUNCH_SPEC(8);
#undef BATCH_LAUNCH_SPEC
#undef BATCH_LAUNCH_SPEC_W
    return ok;
  };
  if (spec_ok && !p.wv_on) spectral = spec_launch(p, st);
  // ... one WAVE per problem where the engine has prepared that form (wv_on): eight problems in flight per CU.  A problem on one wave takes ~12 us per ADMM
  // iteration against 3.7 for a workgroup, and a batch ends with its slowest problem: with a launch order (longest-expected first) the first wv_split
  // positions -- the outliers, 38 of the MPC batch's 4096 problems take 200 .. 375 iterations against a mean of 95 -- go to the workgroup kernel on a second
  // stream, one CU each, while the wave kernel runs on the other CUs.
  if (spec_ok && p.wv_on) {
    const size_t lds_w = batch_wave_lds_bytes(p.n, p.m, p.wv_aend[3] + p.wv_tend[1]);
    const int split = (p.order && p.wv_split > 0 && p.wv_cus > 0 && p.nbatch >= 8 * p.wv_split && cus > 2 * p.wv_cus) ? p.wv_split : 0;
    auto launch = [&](auto kern) {
      if (!lds_w || hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w) != hipSuccess) { (void)hipGetLastError(); return; }
      BatchParams pw = p;
      hipStream_t side = nullptr; hipEvent_t ev0 = nullptr, ev1 = nullptr;
      if (split) {
        if (!d.bside) {
          hipStream_t s2; hipEvent_t a, b;
          if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess)
            throw DeviceError("osqp_hip: cannot create the batch path's second stream");
          d.bside = s2; d.bev0 = a; d.bev1 = b;
        }
        side = static_cast<hipStream_t>(d.bside); ev0 = static_cast<hipEvent_t>(d.bev0); ev1 = static_cast<hipEvent_t>(d.bev1);
        BatchParams ph = p; ph.nbatch = split; ph.wv_on = 0;
        if (hipEventRecord(ev0, st) != hipSuccess || hipStreamWaitEvent(side, ev0, 0) != hipSuccess) throw DeviceError("osqp_hip: batch stream fork failed");
        if (spec_launch(ph, side)) pw.wv_first = split;               // (refused: the wave kernel takes them as well)
        if (hipEventRecord(ev1, side) != hipSuccess) throw DeviceError("osqp_hip: batch stream join failed");
      }
      if (hipMemsetAsync(p.wv_queue, 0, sizeof(int), st) != hipSuccess) throw DeviceError("osqp_hip: batch queue reset failed");
      const int wgs = std::max(1, std::min(cus - (pw.wv_first ? p.wv_cus : 0), p.nbatch - pw.wv_first));      // (fewer problems than CUs: one wave per workgroup gets one)
      hipLaunchKernelGGL(kern, dim3(wgs), dim3(64 * kBatchWaveW), lds_w, st, pw);
      if (split && hipStreamWaitEvent(st, ev1, 0) != hipSuccess) throw DeviceError("osqp_hip: batch stream join failed");
      spectral = true;
    };
    if (batch_wave_n8(p.n) == 64) launch(&k_batch_wave<64>); else if (batch_wave_n8(p.n) == 120) launch(&k_batch_wave<120>); else launch(&k_batch_wave<128>);
    if (!spectral) spectral = spec_launch(p, st);       // (no room for the wave form's LDS: the workgroup form)
  }
  BatchParams pm = p;
  if (spectral) pm.only_marked = 1;
#define p pm
  if (use_dir256) {
    if (e256 <= 2) BATCH_LAUNCH_DIRECT(256, 2); else if (e256 <= 4) BATCH_LAUNCH_DIRECT(256, 4); else if (e256 <= 6) BATCH_LAUNCH_DIRECT(256, 6); else if (e256 <= 8) BATCH_LAUNCH_DIRECT(256, 8); else BATCH_LAUNCH_DIRECT(256, 16);
  } else if (use_dir) {
    if (e64 <= 8) BATCH_LAUNCH_DIRECT(64, 8); else if (e64 <= 16) BATCH_LAUNCH_DIRECT(64, 16); else BATCH_LAUNCH_DIRECT(64, 24);
  } else if (use64) {
    if (e64 <= 8) BATCH_LAUNCH(64, 8, lds_reg); else if (e64 <= 16) BATCH_LAUNCH(64, 16, lds_reg); else BATCH_LAUNCH(64, 24, lds_reg);
  } else if (use256) {
    if (e256 <= 2) BATCH_LAUNCH(256, 2, lds_reg); else if (e256 <= 4) BATCH_LAUNCH(256, 4, lds_reg); else BATCH_LAUNCH(256, 8, lds_reg);
  } else if (lds_gen) {
    BATCH_LAUNCH(256, 0, lds_gen);
  } else {
    return OSQP_FUNC_NOT_IMPLEMENTED;
  }
#undef p
#undef BATCH_LAUNCH
#undef BATCH_LAUNCH_DIRECT
#undef BATCH_LAUNCH_DIRECT_P
#undef BATCH_LAUNCH_DIRECT_N
  hipError_t e = stream ? hipGetLastError() : hipStreamSynchronize(st);
  if (e != hipSuccess) throw DeviceError(std::string("osqp_hip: batch kernel failed: ") + hipGetErrorString(e));
  return OSQP_NO_ERROR;
}

}  // namespace be
}  // namespace osqp_hip
