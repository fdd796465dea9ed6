// dense_hip.hip -- the dense fp64 work of the device-factorised Woodbury correction (backend.h DevWb::large), hand-written for gfx950's matrix cores:
// forming the system (S = W W' / T = W' W: one GEMM) and inverting it (SPD, order up to kWbLargeMax).  Until round 5 this was rocBLAS dgemm +
// rocSOLVER dpotrf / dpotri, loaded on demand (woodbury_hip.hip keeps that route as an A/B switch, OSQPHipPolicy::woodbury_vendor).
//
//   dense_gemm      C = beta C + alpha A B  on v_mfma_f64_16x16x4 (operand layout: tools/mfma_f64_layout.hip -- A: lane l holds A(l % 16, l / 16),
//                   B: B(l / 16, l % 16), result register r of lane l = C(l / 16 + 4 r, l % 16)).  A workgroup of four waves owns a 64 x 64 tile of C
//                   (a wave: 32 x 32 = 2 x 2 instruction tiles); K advances in chunks of 16 through two LDS buffers (the next chunk is fetched into
//                   registers while the current one feeds the matrix cores).  Operands are addressed by strides: each of A, B may be contiguous along
//                   K or along its free index; the LDS image of an operand is laid out so that both its coalesced fill and the instruction's operand
//                   reads are bank-conflict free (free-index-major rows of 80 doubles when the free index is contiguous in memory, K-major rows of 17
//                   doubles when K is).  fp64 MFMA issues one 16x16x4 per 64 cycles and SIMD: the loop is bound by the matrix cores, not by LDS or L2.
//   dense_spd_inverse   in-place inverse of an SPD matrix by BLOCK Gauss-Jordan elimination without pivoting (the pivots of an SPD matrix are positive):
//                   per block step k (64 columns):  P = A_kk^-1 (one workgroup, Gauss-Jordan in registers);  R = P A_k: ;  A_ij -= A_ik R_j  (i, j outside
//                   block k: ONE rank-64 GEMM -- the panels' block-k parts are zeroed in their copies);  A_ik = -A_ik P;  A_k: = R, A_kk = P.  All of the
//                   n^3 work is dense_gemm.  Second form (end of round 6): only the tiles on and above the diagonal are kept current (n^3 flops, half the
//                   bytes) and the next pivot block is inverted on a second stream under the rank update: 17.3 -> 12.1 ms at order 5 000 (see the function).
//                   Measured (MI355X, round 6): the GEMM forms the lasso's systems at 40 TFLOP/s (order 5 000, inner dimension 10 000: 12 ms) and 54 TFLOP/s
//                   (order 10 000, inner 15 000: 55 ms; rocBLAS: 8.4 / 43 ms); the inverse takes 28.5 ms at order 5 000 and 121 ms at 10 000 (rocSOLVER
//                   potrf + potri: 23 / 84 ms) with the pivot block inverted in LDS (221 us per block: 17.5 of the 28.5 ms) -- see k_gj_pivot for its register form.  Block steps of 128
//                   columns: 30.6 / 103 ms (the 128-step pivot chain costs what the halved traffic of the rank update saves at order 5 000).
//                   Accuracy: Gauss-Jordan without pivoting loses about three digits against the Cholesky route on an ill-conditioned system (row-space S of
//                   the lasso at rho = 0.1: |M^-1 K v - v| 2.5e-6 against 5e-9) -- the correction is then used as a preconditioner only (the direct mode
//                   asks for 1e-6); a Newton-Schulz step X += X (I - S X) does NOT repair it (tried: 2e-5 after two steps -- at that conditioning the
//                   residual I - S X is rounding noise).  The column-space system the engine prefers is the better conditioned one (7e-13 .. 1.4e-9).
#include "hip_common.h"

namespace osqp_hip {
namespace be {

namespace {
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kGT = 64, kGK = 16;
constexpr int kLdI = kGT + 16;      // free-index-contiguous operand: LDS image [k][i], 80 doubles per k (rows k and k + 1 fall on the two bank halves)
constexpr int kLdK = kGK + 1;       // K-contiguous operand: LDS image [i][k], 17 doubles per i
constexpr int kOpDoubles = (kGK * kLdI > kGT * kLdK) ? kGK * kLdI : kGT * kLdK;
struct GemmArgs { int M, N, K; double alpha, beta; const double *A; long as_i, as_k; const double *B; long bs_k, bs_j; double *C; long cs_i, cs_j; int upper = 0; int skip = -1; };      // skip: the diagonal tile (skip, skip) is left alone (dense_spd_inverse updates it ahead of the others)      // upper: only the tiles on and above the diagonal (a symmetric product: dense_gemm_sym mirrors them)

// AK / BK: the operand is contiguous along K in memory (else along its free index)
template <bool AK, bool BK>
__global__ __launch_bounds__(256) void k_dgemm(GemmArgs g) {
  __shared__ double As[2][kOpDoubles], Bs[2][kOpDoubles];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  if (g.upper && blockIdx.y > blockIdx.x) return;
  if ((int)blockIdx.x == g.skip && (int)blockIdx.y == g.skip) return;
  const int i0 = blockIdx.y * kGT, j0 = blockIdx.x * kGT;
  const int rb = 32 * (w >> 1), cb = 32 * (w & 1), lj = l & 15, lk = l >> 4;
  v4d acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) acc[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
  double ra[4], rbv[4];
  // element (f, k) of a 64 x 16 operand chunk this thread moves in round q: coalesced along the operand's contiguous direction
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      { const int f = AK ? (tid >> 4) + 16 * q : (tid & 63), k = AK ? (tid & 15) : (tid >> 6) + 4 * q;
        const int gi = i0 + f, gk = k0 + k;
        ra[q] = (gi < g.M && gk < g.K) ? g.A[(long)gi * g.as_i + (long)gk * g.as_k] : 0.0; }
      { const int f = BK ? (tid >> 4) + 16 * q : (tid & 63), k = BK ? (tid & 15) : (tid >> 6) + 4 * q;
        const int gj = j0 + f, gk = k0 + k;
        rbv[q] = (gj < g.N && gk < g.K) ? g.B[(long)gk * g.bs_k + (long)gj * g.bs_j] : 0.0; }
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      { const int f = AK ? (tid >> 4) + 16 * q : (tid & 63), k = AK ? (tid & 15) : (tid >> 6) + 4 * q; As[buf][AK ? f * kLdK + k : k * kLdI + f] = ra[q]; }
      { const int f = BK ? (tid >> 4) + 16 * q : (tid & 63), k = BK ? (tid & 15) : (tid >> 6) + 4 * q; Bs[buf][BK ? f * kLdK + k : k * kLdI + f] = rbv[q]; }
    }
  };
  fetch(0);
  stage(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < g.K; k0 += kGK) {
    const bool more = k0 + kGK < g.K;
    if (more) fetch(k0 + kGK);                              // (in flight while this chunk feeds the matrix cores)
    const double *as = As[buf], *bs = Bs[buf];
#pragma unroll
    for (int ks = 0; ks < kGK; ks += 4) {
      const int kk = ks + lk;
      const double a0 = AK ? as[(rb + lj) * kLdK + kk] : as[kk * kLdI + rb + lj], a1 = AK ? as[(rb + 16 + lj) * kLdK + kk] : as[kk * kLdI + rb + 16 + lj];
      const double b0 = BK ? bs[(cb + lj) * kLdK + kk] : bs[kk * kLdI + cb + lj], b1 = BK ? bs[(cb + 16 + lj) * kLdK + kk] : bs[kk * kLdI + cb + 16 + lj];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) { stage(buf ^ 1); __syncthreads(); buf ^= 1; }      // (the other buffer: nobody reads it in this iteration; one barrier per chunk)
  }
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int gi = i0 + rb + 16 * a + lk + 4 * r, gj = j0 + cb + 16 * b + lj;
        if (gi < g.M && gj < g.N) {
          double *c = g.C + (long)gi * g.cs_i + (long)gj * g.cs_j;
          *c = g.beta == 0.0 ? g.alpha * acc[a][b][r] : g.beta * *c + g.alpha * acc[a][b][r];
        }
      }
}

constexpr int kGjNb = 64;
// P = (A_kk)^-1 by Gauss-Jordan elimination, no pivoting; the smallest pivot seen goes to minpiv (<= 0 or NaN: not positive definite).
// The 64 x 64 block lives in REGISTERS: thread (column j = tid mod 64, row group g = tid / 64) holds rows g, g + 4, .. of its column; a step publishes
// row k and column k through LDS (two buffers alternating by the parity of k: ONE barrier per step) and every thread updates its sixteen
// elements (row k's register is picked by a select chain: the compiler refuses to unroll all 64 steps).  A block of fewer than 64 columns is padded
// with the identity.  (First version: the block in LDS, sixteen elements per thread addressed through an integer division, two barriers per step --
// 221 us per block, 17.5 of the 28.5 ms of an inversion of order 5 000: profiles/r06k_lasso_kernel_stats_before_pivot.csv.)
// upd != 0: the block first receives its share of the CURRENT step's rank update,  A_kk -= Ck[k0 .., :] R[:, k0 ..]  (kc columns of the panel), in registers --
// nothing else needs the updated block (the next step's panels leave block k out, its row panel write replaces it by P): the block's inversion then
// depends on R alone and runs beside the rank update of the other tiles (dense_spd_inverse).  The wave priority is raised: the workgroup shares its CU
// with GEMM waves, and its chain of 64 dependent steps is the critical path of a block step.
__global__ __launch_bounds__(256) void k_gj_pivot(const double *A, long ld, int k0, int nb, double *P, double *minpiv, int upd, const double *Ck, const double *R, long ldr, int kc) {
  static_assert(kGjNb == 64, "thread layout: 64 columns x 4 row groups");
  __shared__ double rowb[2][kGjNb], colb[2][kGjNb];
  const int tid = threadIdx.x, j = tid & 63, g = tid >> 6;
  double m[16];
#pragma unroll
  for (int q = 0; q < 16; q++) { const int i = g + 4 * q; m[q] = (i < nb && j < nb) ? A[(long)(k0 + i) * ld + k0 + j] : (i == j ? 1.0 : 0.0); }
  if (upd) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ double sA[kGjNb][17], sB[16][kGjNb];
    for (int c0 = 0; c0 < kc; c0 += 16) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int e = tid + 256 * q;
        { const int il = e >> 4, cc = e & 15; sA[il][cc] = (il < nb && c0 + cc < kc) ? Ck[(long)(k0 + il) * kGjNb + c0 + cc] : 0.0; }
        { const int cc = e >> 6, jj = e & 63; sB[cc][jj] = (jj < nb && c0 + cc < kc) ? R[(long)(c0 + cc) * ldr + k0 + jj] : 0.0; }
      }
      __syncthreads();
#pragma unroll
      for (int cc = 0; cc < 16; cc++) {
        const double rv = sB[cc][j];
#pragma unroll
        for (int q = 0; q < 16; q++) m[q] = fma(-sA[g + 4 * q][cc], rv, m[q]);
      }
      __syncthreads();
    }
  }
  double pmin = 1e300;
  // (all 64 steps unrolled: the register that holds row k's element is then a compile-time index -- with the loop rolled it was picked, and written back,
  //  through a 16-way select chain in every step: 78 us per block)
#pragma clang loop unroll(full)
  for (int k = 0; k < kGjNb; k++) {
    const int b = k & 1, gk = k & 3, qk = k >> 2;
    if (g == gk) rowb[b][j] = m[qk];
    if (j == k) {
#pragma unroll
      for (int q = 0; q < 16; q++) colb[b][g + 4 * q] = m[q];
    }
    __syncthreads();
    const double p = rowb[b][k], pi = 1.0 / p;
    if (!(p >= pmin)) pmin = p;                            // (a NaN pivot is kept)
    const double rk = (j == k) ? pi : rowb[b][j] * pi;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const double ck = colb[b][g + 4 * q];
      m[q] = (j == k) ? -ck * pi : m[q] - ck * rk;
    }
    if (g == gk) m[qk] = rk;
  }
#pragma unroll
  for (int q = 0; q < 16; q++) { const int i = g + 4 * q; if (i < nb && j < nb) P[i * kGjNb + j] = m[q]; }
  if (tid == 0 && !(pmin >= *minpiv)) *minpiv = pmin;
}
// The panels of block step k, built from the UPPER triangle alone (tiles (i, j), i <= j, are the only ones the inverse keeps current; see dense_spd_inverse):
//   Ck (n x 64, row-major)  = the column panel A[:, k0 .. k0 + nb) with block k's own rows zeroed: rows above the block are read where they lie, rows below it
//                             are the transposed row panel (both blocks still unswept: the trailing matrix is symmetric);
//   Rw (64 x n, row-major)  = the row panel A[k0 .. k0 + nb, :]: right of the block where it lies, left of it MINUS the transposed column panel (Gauss-Jordan
//                             without pivoting keeps A_us = -A_su' between a swept block s and an unswept block u); block k's own columns zero.
// One workgroup per 64 rows of Ck: the 64 x 64 tile goes through LDS so that the read and both writes are coalesced.
__global__ __launch_bounds__(256) void k_gj_panels(const double *A, long ld, int n, int k0, int nb, double *Ck, double *Rw) {
  __shared__ double T[kGjNb][kGjNb + 1];
  const int tid = threadIdx.x, lo = tid & 63, hi = tid >> 6;
  const int i0 = blockIdx.x * kGjNb, kb = k0 / kGjNb, t = blockIdx.x;
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const int h = hi + 4 * q;
    if (t < kb) { const int il = h, j = lo; T[il][j] = (j < nb) ? A[(long)(i0 + il) * ld + k0 + j] : 0.0; }                              // (i0 + il < k0 <= n)
    else if (t > kb) { const int il = lo, j = h; T[il][j] = (j < nb && i0 + il < n) ? A[(long)(k0 + j) * ld + i0 + il] : 0.0; }
    else T[h][lo] = 0.0;
  }
  __syncthreads();
  const double sg = t < kb ? -1.0 : 1.0;
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const int h = hi + 4 * q;
    if (i0 + h < n) Ck[(long)(i0 + h) * kGjNb + lo] = T[h][lo];
    if (i0 + lo < n) Rw[(long)h * n + i0 + lo] = sg * T[lo][h];
  }
}
// row panel, the part in the upper triangle: A[k0 + i][j] = P[i][j - k0] inside block k, = R[i][j] right of it
__global__ __launch_bounds__(256) void k_gj_rowpanel(double *A, long ld, int n, int k0, int nb, const double *R, long ldr, const double *P) {
  const int w = n - k0;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < (long)nb * w; e += (long)gridDim.x * 256) {
    const int i = (int)(e / w), j = k0 + (int)(e - (long)i * w);
    A[(long)(k0 + i) * ld + j] = (j < k0 + nb) ? P[i * kGjNb + (j - k0)] : R[(long)i * ldr + j];
  }
}
__global__ void k_set1(double *p, double v) { *p = v; }

template <bool AK, bool BK>
void launch_gemm(hipStream_t s, const GemmArgs &g) {
  hipLaunchKernelGGL((k_dgemm<AK, BK>), dim3((g.N + kGT - 1) / kGT, (g.M + kGT - 1) / kGT), dim3(256), 0, s, g);
}
}  // namespace

// C (M x N) = beta C + alpha A B with A(i, k) = A[i as_i + k as_k], B(k, j) = B[k bs_k + j bs_j], C(i, j) = C[i cs_i + j cs_j]; each of A, B must have one unit stride
void dense_gemm(void *stream, int M, int N, int K, double alpha, const double *A, long as_i, long as_k, const double *B, long bs_k, long bs_j, double beta, double *C, long cs_i, long cs_j) {
  if (M <= 0 || N <= 0) return;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const GemmArgs g{M, N, K, alpha, beta, A, as_i, as_k, B, bs_k, bs_j, C, cs_i, cs_j};
  const bool ak = as_k == 1, bk = bs_k == 1;
  if (!(ak || as_i == 1) || !(bk || bs_j == 1)) throw DeviceError("osqp_hip: dense_gemm needs a unit stride in every operand");
  if (ak && bk) launch_gemm<true, true>(s, g); else if (ak) launch_gemm<true, false>(s, g); else if (bk) launch_gemm<false, true>(s, g); else launch_gemm<false, false>(s, g);
}
// C (N x N, row-major, leading dimension ld) = alpha A B for a product known to be symmetric (A B = W' W): the tiles on and above the diagonal by the
// GEMM kernel, the strict lower triangle copied from them -- half the matrix-core work (T of the lasso: 12.0 -> 6.5 ms)
__global__ __launch_bounds__(256) void k_sym_mirror(double *C, long ld, int n) {
  __shared__ double t[32][33];
  const int bi = blockIdx.y * 32, bj = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (bi > bj) return;                                   // (tile (bi, bj) of the upper triangle -> tile (bj, bi))
  for (int r = ty; r < 32; r += 8) if (bi + r < n && bj + tx < n) t[r][tx] = C[(long)(bi + r) * ld + bj + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8) { const int i = bj + r, j = bi + tx; if (i < n && j < n && i > j) C[(long)i * ld + j] = t[tx][r]; }
}
void dense_gemm_sym(void *stream, int N, int K, double alpha, const double *A, long as_i, long as_k, const double *B, long bs_k, long bs_j, double *C, long ld) {
  if (N <= 0) return;
  hipStream_t s = static_cast<hipStream_t>(stream);
  GemmArgs g{N, N, K, alpha, 0.0, A, as_i, as_k, B, bs_k, bs_j, C, ld, 1};
  g.upper = 1;
  const bool ak = as_k == 1, bk = bs_k == 1;
  if (!(ak || as_i == 1) || !(bk || bs_j == 1)) throw DeviceError("osqp_hip: dense_gemm needs a unit stride in every operand");
  if (ak && bk) launch_gemm<true, true>(s, g); else if (ak) launch_gemm<true, false>(s, g); else if (bk) launch_gemm<false, true>(s, g); else launch_gemm<false, false>(s, g);
  hipLaunchKernelGGL(k_sym_mirror, dim3((N + 31) / 32, (N + 31) / 32), dim3(256), 0, s, C, ld, N);
}
// A (n x n, row-major, leading dimension ld, SPD) <- A^-1 in place; work: dense_spd_inverse_work(n) doubles (device).  The smallest pivot any block saw is
// left in minpiv[0] (device): <= 0 (or NaN) means the matrix was not positive definite.
//
// Round 6, second form: ONE TRIANGLE and a pivot that runs AHEAD.  Gauss-Jordan without pivoting on a symmetric matrix keeps, between the swept blocks S and
// the unswept blocks U,  A_SS and A_UU symmetric and A_US = -A_SU'  -- so the tiles on and above the diagonal determine the matrix at every step.  A block
// step builds its two panels from those tiles (k_gj_panels), the rank-64 update touches the upper tiles only (half the flops, half the 2 x 8 n^2 bytes a step
// moves -- the update is bandwidth-bound: 8 flop / byte), the lower triangle is mirrored once at the end (everything swept: symmetric).  The next pivot
// block (k + 1, k + 1) takes its share of the update inside the launch that inverts it (one workgroup, raised priority, a chain of 64 dependent steps) --
// on a second stream, UNDER the rank update of the other tiles instead of in front of the next step.  Pivot blocks alternate between two buffers.
// (First form: full matrix, pivot in line: 17.3 ms at order 5 000, profiles/r06k_*.)
size_t dense_spd_inverse_work(int n) { return (size_t)n * kGjNb * 3 + (size_t)kGjNb * kGjNb * 2 + 8; }
namespace {
struct GjAux { int dev = -1; hipStream_t s2 = nullptr; hipEvent_t ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr}; };
GjAux &gj_aux() {
  static thread_local GjAux a;
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  if (a.dev != dev) {                                       // (one auxiliary stream + four events per host thread and device; never inside a capture: the factorisation is host-driven)
    HIP_CHECK(hipStreamCreateWithFlags(&a.s2, hipStreamNonBlocking));
    for (int q = 0; q < 2; q++) { HIP_CHECK(hipEventCreateWithFlags(&a.ready[q], hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&a.done[q], hipEventDisableTiming)); }
    a.dev = dev;
  }
  return a;
}
}  // namespace
void dense_spd_inverse(void *stream, double *A, long ld, int n, double *work, double *minpiv) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  double *Ck = work, *Rw = Ck + (size_t)n * kGjNb, *R = Rw + (size_t)n * kGjNb, *Pb = R + (size_t)n * kGjNb;
  static const bool lookahead = []() { const char *e = std::getenv("OSQP_HIP_GJ_LOOKAHEAD"); return !(e && e[0] == '0'); }();
  GjAux *aux = lookahead ? &gj_aux() : nullptr;
  hipLaunchKernelGGL(k_set1, dim3(1), dim3(1), 0, s, minpiv, 1e300);
  const int nblk = (n + kGjNb - 1) / kGjNb;
  hipLaunchKernelGGL(k_gj_pivot, dim3(1), dim3(256), 0, s, A, ld, 0, std::min(kGjNb, n), Pb, minpiv, 0, nullptr, nullptr, 0L, 0);
  for (int kb = 0; kb < nblk; kb++) {
    const int k0 = kb * kGjNb, nb = std::min(kGjNb, n - k0);
    double *P = Pb + (size_t)(kb & 1) * kGjNb * kGjNb;
    if (kb > 0 && aux) HIP_CHECK(hipStreamWaitEvent(s, aux->done[kb & 1], 0));                           // P of this block: inverted under the previous step's update
    hipLaunchKernelGGL(k_gj_panels, dim3(nblk), dim3(256), 0, s, A, ld, n, k0, nb, Ck, Rw);              //  (behind the wait: that launch read the previous Ck and R)
    dense_gemm(s, nb, n, nb, 1.0, P, kGjNb, 1, Rw, n, 1, 0.0, R, n, 1);                                    // R = P A_k:   (block k's own columns: zero, as in Rw)
    int skip = -1;
    if (kb + 1 < nblk) {
      // the next pivot block: updated and inverted by one workgroup beside the update of the rest
      const int k1 = k0 + kGjNb, nb1 = std::min(kGjNb, n - k1);
      double *Pn = Pb + (size_t)((kb + 1) & 1) * kGjNb * kGjNb;
      skip = kb + 1;
      hipStream_t sp = s;
      if (aux) {
        HIP_CHECK(hipEventRecord(aux->ready[kb & 1], s));
        HIP_CHECK(hipStreamWaitEvent(aux->s2, aux->ready[kb & 1], 0));
        sp = aux->s2;
      }
      hipLaunchKernelGGL(k_gj_pivot, dim3(1), dim3(256), 0, sp, A, ld, k1, nb1, Pn, minpiv, 1, Ck, R, (long)n, nb);
      if (aux) HIP_CHECK(hipEventRecord(aux->done[(kb + 1) & 1], aux->s2));
    }
    { GemmArgs g{n, n, nb, -1.0, 1.0, Ck, kGjNb, 1, R, (long)n, 1, A, ld, 1};                             // A_ij -= A_ik R_j on the upper tiles outside block row / column k
      g.upper = 1; g.skip = skip;
      launch_gemm<true, false>(s, g); }
    dense_gemm(s, k0, nb, nb, -1.0, Ck, kGjNb, 1, P, kGjNb, 1, 0.0, A + k0, ld, 1);                       // A_ik = -A_ik P  above the block
    const int gridr = std::max(1, std::min(4 * kGrid, (int)(((long)nb * (n - k0) + 255) / 256)));
    hipLaunchKernelGGL(k_gj_rowpanel, dim3(gridr), dim3(256), 0, s, A, ld, n, k0, nb, R, (long)n, P);
  }
  hipLaunchKernelGGL(k_sym_mirror, dim3((n + 31) / 32, (n + 31) / 32), dim3(256), 0, s, A, ld, n);
}

}  // namespace be
}  // namespace osqp_hip
