// Prototype + microbenchmark (MI355X, one wave): the banded substitutions  L^ D L^' x = b  of the batch kernel (batch_hip.hip ksolve)
//   (a) as they are: one v_readlane pair + one FMA per pivot, a chain of n pivots per pass;
//   (b) blocked by 16 with DPP row broadcasts: per block of 16 unknowns  t = b_R - L^[R, R-2..R-1] x_prev  (8 steps: every row of 16
//       lanes takes a quarter of the 32 columns, one row_newbcast pair serves all four rows at once) and  x_R = inv(L^_RR) t  (4 steps),
//       each followed by a sum over the four rows (v_permlane16_swap / v_permlane32_swap).  inv(L^_RR) lives IN PLACE of the diagonal
//       blocks of the band.  No v_readlane, no dependency chain longer than a block.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/blocked_subst tools/blocked_subst_bench.hip && /tmp/blocked_subst
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include <type_traits>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef unsigned u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ double dpp_f64(double old, double v) {      // (64-bit operands: one v_mov_b64_dpp for row_newbcast, two v_mov_b32_dpp otherwise)
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROWMASK, 0xf, false);
}
template <int K> __device__ __forceinline__ double rowbcast(double v) { return dpp_f64<0x150 + K>(0.0, v); }
template <bool OFF = false>
__device__ __forceinline__ double rows_sum(double p) {
  if (OFF) return p;                 // sum over the four rows of 16, result in every row
  u2v a = __builtin_amdgcn_permlane16_swap(__double2loint(p), __double2loint(p), false, false);
  u2v b = __builtin_amdgcn_permlane16_swap(__double2hiint(p), __double2hiint(p), false, false);
  const double s = __hiloint2double(b.x, a.x) + __hiloint2double(b.y, a.y);
  a = __builtin_amdgcn_permlane32_swap(__double2loint(s), __double2loint(s), false, false);
  b = __builtin_amdgcn_permlane32_swap(__double2hiint(s), __double2hiint(s), false, false);
  return __hiloint2double(b.x, a.x) + __hiloint2double(b.y, a.y);
}

constexpr int NB = 8;
#ifndef PINMODE
#define PINMODE 0
#endif
#if PINMODE == 0
#define PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PIN2() PIN()
#elif PINMODE == 1
#define PIN() do { asm volatile("" ::: "memory"); } while (0)
#define PIN2() ((void)0)
#else
#define PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PIN2() ((void)0)
#endif
struct Prob { int n, bw, W, n8, n16; };

// ---- (a) the kernel's current form (register-resident variant, n <= 128) ----
__device__ void solve_chain(const Prob P, const double *Lr, const double *dinv, const double *rhs, double *out) {
  const int tid = threadIdx.x, n = P.n, bw = P.bw, W = P.W, n8 = P.n8;
  const int e0 = tid, e1 = tid + 64;
  double cur = e0 < n ? rhs[e0] : 0.0, nxt = e1 < n ? rhs[e1] : 0.0;
  const double di0 = e0 < n ? dinv[e0] : 0.0, di1 = e1 < n ? dinv[e1] : 0.0;
  double v0 = 0.0, v1 = 0.0;
  const int nblk = n8 / NB;
  {
    auto fetch = [&](int p0, double (&l)[NB]) {
      const int dl = (tid - p0) & 63;
      const bool act = dl < bw + NB && p0 < n8;
      const double *col = act ? Lr + p0 * W + dl : Lr - 1;
      const int stride = act ? W - 1 : 0;
#ifdef CHAIN_INCR
#pragma unroll
      for (int q = 0; q < NB; q++) { l[q] = *col; col += stride; }
#else
#pragma unroll
      for (int q = 0; q < NB; q++) l[q] = col[q * stride];
#endif
    };
    auto block = [&](int p0, const double (&l)[NB]) {
#pragma unroll
      for (int q = 0; q < NB; q++) { const double vq = readlane_f64(cur, (p0 + q) & 63); cur -= l[q] * vq; }
      const bool piv = ((tid - p0) & 63) < NB, lo = p0 < 64;
      v0 = (piv && lo) ? cur : v0; v1 = (piv && !lo) ? cur : v1;
      cur = piv ? nxt : cur;
    };
    double la[NB], lb[NB];
    fetch(0, la);
#pragma clang loop unroll(disable)
    for (int b = 0; b < nblk; b += 2) {
      fetch((b + 1) * NB, lb); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
      block(b * NB, la);
      if (b + 1 < nblk) { fetch((b + 2) * NB, la); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); block((b + 1) * NB, lb); }
    }
  }
  v0 *= di0; v1 *= di1;
  {
    const bool two = e1 <= n8 - 1;
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
      fetch(b * NB - 1, lb); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
      block(b * NB + NB - 1, la);
      if (b >= 1) { fetch(b * NB - NB - 1, la); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); block(b * NB - 1, lb); }
    }
    if (e0 < n) out[e0] = x0;
    if (e1 < n) out[e1] = x1;
  }
}

// ---- (a') the same chain with COMPILE-TIME lane indices: the block loop unrolled over one rotation of the 64 lanes (8 blocks), so that
//      v_readlane takes an immediate lane (no s_add / s_nop per pivot) ----
__device__ void solve_chain_imm(const Prob P, const double *Lr, const double *dinv, const double *rhs, double *out) {
  const int tid = threadIdx.x, n = P.n, bw = P.bw, W = P.W, n8 = P.n8;
  const int e0 = tid, e1 = tid + 64;
  double cur = e0 < n ? rhs[e0] : 0.0, nxt = e1 < n ? rhs[e1] : 0.0;
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
    auto block = [&](auto BB, int p0, const double (&l)[NB]) {
      constexpr int L0 = decltype(BB)::value * NB;
#pragma unroll
      for (int q = 0; q < NB; q++) { const double vq = readlane_f64(cur, L0 + q); cur -= l[q] * vq; }
      const bool piv = ((tid - p0) & 63) < NB, lo = p0 < 64;
      v0 = (piv && lo) ? cur : v0; v1 = (piv && !lo) ? cur : v1;
      cur = piv ? nxt : cur;
    };
    double la[NB], lb[NB];
    fetch(0, la);
#pragma clang loop unroll(disable)
    for (int g = 0; g < nblk; g += 8) {
#define STEP(BBV, A, B) if (g + BBV < nblk) { fetch((g + BBV + 1) * NB, B); PIN(); block(std::integral_constant<int, BBV>{}, (g + BBV) * NB, A); }
      STEP(0, la, lb) STEP(1, lb, la) STEP(2, la, lb) STEP(3, lb, la) STEP(4, la, lb) STEP(5, lb, la) STEP(6, la, lb) STEP(7, lb, la)
#undef STEP
    }
  }
  v0 *= di0; v1 *= di1;
  {
    const bool two = e1 <= n8 - 1;
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
    // blocks from the top: block index b (elements 8b .. 8b+7), lane of pivot (top - q) & 63 with top = 8b + 7: compile-time for b mod 8
    auto block = [&](auto BB, int top, const double (&l)[NB]) {
      constexpr int T0 = decltype(BB)::value * NB + NB - 1;
#pragma unroll
      for (int q = 0; q < NB; q++) { const double xq = readlane_f64(cur, T0 - q); cur -= l[q] * xq; }
      const bool piv = ((top - tid) & 63) < NB, hi = top >= 64;
      x1 = (piv && hi) ? cur : x1; x0 = (piv && !hi) ? cur : x0;
      cur = piv ? nxt : cur;
    };
    double la[NB], lb[NB];
    // walk b = nblk-1 .. 0; within a group of 8 (b & 7 = 7 .. 0) the lanes are compile-time; the first group may be partial
    int b = nblk - 1;
    fetch(b * NB + NB - 1, la);
    bool useA = true;
#pragma clang loop unroll(disable)
    for (int gtop = (b | 7); gtop >= 7; gtop -= 8) {
#define STEP(BBV) if (gtop - (7 - BBV) <= b && gtop - (7 - BBV) >= 0) { const int bb_ = gtop - (7 - BBV); \
        if (useA) { fetch(bb_ * NB - 1, lb); PIN(); block(std::integral_constant<int, BBV>{}, bb_ * NB + NB - 1, la); } \
        else { fetch(bb_ * NB - 1, la); PIN(); block(std::integral_constant<int, BBV>{}, bb_ * NB + NB - 1, lb); } useA = !useA; }
      STEP(7) STEP(6) STEP(5) STEP(4) STEP(3) STEP(2) STEP(1) STEP(0)
#undef STEP
    }
    if (e0 < n) out[e0] = x0;
    if (e1 < n) out[e1] = x1;
  }
}

// ---- (b) blocked by 16, DPP broadcasts.  Lq = the band with inv(L^_RR) in place of the diagonal blocks; bw <= 32 ----
template <int KNOCK>
__device__ void solve_blocked(const Prob P, const double *Lq, const double *dinv, const double *rhs, double *out, double *vbuf) {
  const int tid = threadIdx.x, r = tid >> 4, c = tid & 15, bw = P.bw, W = P.W, nb16 = P.n16 / 16;
  const double *Z = Lq - 1;                                           // a zero
  const bool lowrows = r < 2;
  // ---------------- forward ----------------
  {
    // stage-1 coefficient k of lane (r, c) in block R:  L^[i][j], i = 16R + c, j = 16(R-2) + 8r + k, d = i - j = 32 + c - 8r - k  (<= bw or masked)
    // stage-2 coefficient k:  inv[c][4k + r]  at column 16R + 4k + r, d = c - 4k - r  (> 0 or masked)
    const double *a1[8], *a2[4];
    int inc1[8], inc2[4];
#pragma unroll
    for (int k = 0; k < 8; k++) { const int d = 32 + c - 8 * r - k; const bool ok = d <= bw; a1[k] = ok ? Lq + (-32 + 8 * r + k) * W + d : Z; inc1[k] = ok ? 16 * W : 0; }
#pragma unroll
    for (int k = 0; k < 4; k++) { const int d = c - 4 * k - r; const bool ok = d > 0; a2[k] = ok ? Lq + (4 * k + r) * W + d : Z; inc2[k] = ok ? 16 * W : 0; }
    double xa = 0.0, xb = 0.0;
    struct Co { double l1[8], l2[4], b; };
    auto fetch = [&](int R, Co &co) {                                 // coefficients + right-hand side of block R (calls in block order)
      if (R >= nb16) return;
#pragma unroll
      for (int k = 0; k < 8; k++) { const double *a = a1[k] < Z ? Z : a1[k]; co.l1[k] = (KNOCK & 4) ? 1e-3 * k : *a; a1[k] += inc1[k]; }    // (columns j < 0 of the first two blocks: zero)
#pragma unroll
      for (int k = 0; k < 4; k++) { co.l2[k] = (KNOCK & 4) ? 1e-3 * k : *a2[k]; a2[k] += inc2[k]; }
      co.b = rhs[16 * R + c];
    };
    auto block = [&](int R, const Co &co) {
      double src = lowrows ? xa : xb;
      src = dpp_f64<0x128, 0xa>(src, src);                            // odd rows: rotate by 8
      // (all broadcasts first, then the products: independent instructions back to back instead of mov, mov, dependent FMA)
      const double b0 = rowbcast<0>(src), b1 = rowbcast<1>(src), b2 = rowbcast<2>(src), b3 = rowbcast<3>(src), b4 = rowbcast<4>(src), b5 = rowbcast<5>(src), b6 = rowbcast<6>(src), b7 = rowbcast<7>(src);
      PIN2();
      double q0 = co.l1[0] * b0, q1 = co.l1[1] * b1, q2 = co.l1[2] * b2, q3 = co.l1[3] * b3;
      q0 += co.l1[4] * b4; q1 += co.l1[5] * b5; q2 += co.l1[6] * b6; q3 += co.l1[7] * b7;
      const double t = co.b - rows_sum<(KNOCK & 1) != 0>((q0 + q1) + (q2 + q3));
      double s2 = t;                                                  // s2[lane l of row r] = t[(l + r) & 15]
      s2 = dpp_f64<0x120 + 15, 0x2>(s2, t); s2 = dpp_f64<0x120 + 14, 0x4>(s2, t); s2 = dpp_f64<0x120 + 13, 0x8>(s2, t);
      const double c0 = rowbcast<0>(s2), c1 = rowbcast<4>(s2), c2 = rowbcast<8>(s2), c3 = rowbcast<12>(s2);
      PIN2();
      const double x = (KNOCK & 2) ? t : t + rows_sum<(KNOCK & 1) != 0>((co.l2[0] * c0 + co.l2[1] * c1) + (co.l2[2] * c2 + co.l2[3] * c3));
      xa = xb; xb = x;
      if (r == 0) vbuf[16 * R + c] = x;
    };
    Co ca, cb;
    fetch(0, ca);
#pragma clang loop unroll(disable)
    for (int R = 0; R < nb16; R += 2) {
      fetch(R + 1, cb); PIN();
      block(R, ca);
      if (R + 1 < nb16) { fetch(R + 2, ca); PIN(); block(R + 1, cb); }
    }
  }
  // ---------------- backward ----------------
  {
    // stage 1:  L^[j][i], i = 16R + c, j = 16(R+1) + 8r + k, d = j - i = 16 + 8r + k - c ; address Lq + i W + d
    // stage 2:  inv[4k + r][c]  (4k + r > c): address Lq + i W + (4k + r - c)
    const int Rt = nb16 - 1;
    const double *a1[8], *a2[4];
    int inc1[8], inc2[4];
#pragma unroll
    for (int k = 0; k < 8; k++) { const int d = 16 + 8 * r + k - c; const bool ok = d <= bw; a1[k] = ok ? Lq + (16 * Rt + c) * W + d : Z; inc1[k] = ok ? 16 * W : 0; }
#pragma unroll
    for (int k = 0; k < 4; k++) { const int d = 4 * k + r - c; const bool ok = d > 0; a2[k] = ok ? Lq + (16 * Rt + c) * W + d : Z; inc2[k] = ok ? 16 * W : 0; }
    double xa = 0.0, xb = 0.0;                                        // xa = x_{R+2}, xb = x_{R+1}
    struct Co { double l1[8], l2[4], g; };
    auto fetch = [&](int R, Co &co) {                                 // (calls in descending block order)
      if (R < 0) return;
#pragma unroll
      for (int k = 0; k < 8; k++) { co.l1[k] = (KNOCK & 4) ? 1e-3 * k : *a1[k]; a1[k] -= inc1[k]; }
#pragma unroll
      for (int k = 0; k < 4; k++) { co.l2[k] = (KNOCK & 4) ? 1e-3 * k : *a2[k]; a2[k] -= inc2[k]; }
      co.g = vbuf[16 * R + c] * dinv[16 * R + c];
    };
    auto block = [&](int R, const Co &co) {
      double src = lowrows ? xb : xa;
      src = dpp_f64<0x128, 0xa>(src, src);
      // (all broadcasts first, then the products: independent instructions back to back instead of mov, mov, dependent FMA)
      const double b0 = rowbcast<0>(src), b1 = rowbcast<1>(src), b2 = rowbcast<2>(src), b3 = rowbcast<3>(src), b4 = rowbcast<4>(src), b5 = rowbcast<5>(src), b6 = rowbcast<6>(src), b7 = rowbcast<7>(src);
      PIN2();
      double q0 = co.l1[0] * b0, q1 = co.l1[1] * b1, q2 = co.l1[2] * b2, q3 = co.l1[3] * b3;
      q0 += co.l1[4] * b4; q1 += co.l1[5] * b5; q2 += co.l1[6] * b6; q3 += co.l1[7] * b7;
      const double t = co.g - rows_sum<(KNOCK & 1) != 0>((q0 + q1) + (q2 + q3));
      double s2 = t;
      s2 = dpp_f64<0x120 + 15, 0x2>(s2, t); s2 = dpp_f64<0x120 + 14, 0x4>(s2, t); s2 = dpp_f64<0x120 + 13, 0x8>(s2, t);
      const double c0 = rowbcast<0>(s2), c1 = rowbcast<4>(s2), c2 = rowbcast<8>(s2), c3 = rowbcast<12>(s2);
      PIN2();
      const double x = (KNOCK & 2) ? t : t + rows_sum<(KNOCK & 1) != 0>((co.l2[0] * c0 + co.l2[1] * c1) + (co.l2[2] * c2 + co.l2[3] * c3));
      xa = xb; xb = x;
      if (r == 0 && 16 * R + c < P.n) out[16 * R + c] = x;
    };
    Co ca, cb;
    fetch(Rt, ca);
#pragma clang loop unroll(disable)
    for (int R = Rt; R >= 0; R -= 2) {
      fetch(R - 1, cb); PIN();
      block(R, ca);
      if (R >= 1) { fetch(R - 2, ca); PIN(); block(R - 1, cb); }
    }
  }
}

__global__ void k_check(int *o) {                                      // what the DPP controls do, on lane ids
  const int l = threadIdx.x;
  o[l] = __builtin_amdgcn_update_dpp(l, l, 0x128, 0xa, 0xf, false);                       // row_ror:8 on odd rows
  o[64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x120 + 15, 0x2, 0xf, false);            // row_ror:15 on row 1
  o[128 + l] = __builtin_amdgcn_update_dpp(0, l, 0x150 + 4, 0xf, 0xf, false);             // row_newbcast:4
  u2v a = __builtin_amdgcn_permlane16_swap(l, l, false, false);
  o[192 + l] = a.x * 100 + a.y;
  a = __builtin_amdgcn_permlane32_swap(l, l, false, false);
  o[256 + l] = a.x * 100 + a.y;
}

constexpr int kReps = 200;
__global__ void k_solve(int mode, Prob P, const double *Lg, const double *Lqg, const double *dinvg, const double *rhsg, double *outg, long long *cyc, int lds_doubles) {
  extern __shared__ double sh[];
  const int tid = threadIdx.x;
  const int tot = 8 + P.n16 * P.W + 64;
  double *La = sh + 8, *Lqa = sh + tot + 8, *dinv = sh + 2 * tot, *rhs = dinv + P.n16, *out = rhs + P.n16, *vbuf = out + P.n16;
  for (int k = tid; k < tot; k += 64) { sh[k] = Lg[k]; sh[tot + k] = Lqg[k]; }
  for (int k = tid; k < P.n16; k += 64) { dinv[k] = k < P.n ? dinvg[k] : 0.0; rhs[k] = k < P.n ? rhsg[k] : 0.0; out[k] = 0.0; vbuf[k] = 0.0; }
  __syncthreads();
  if (mode == 6) {                                                   // inv(L^_RR) computed on the device, as batch_hip.hip factorize() does
    for (int k = tid; k < tot; k += 64) sh[tot + k] = sh[k];
    __syncthreads();
    double inv[2][16];
    for (int h = 0; h < 2; h++) {
      const int vt = tid + 64 * h, R = vt >> 4, bcol = vt & 15, base = 16 * R;
      if (base < P.n) {
#pragma unroll
        for (int a = 0; a < 16; a++) {
          double s_ = a == bcol ? 1.0 : 0.0;
#pragma unroll
          for (int k = 0; k < a; k++) s_ -= Lqa[(base + k) * P.W + (a - k)] * inv[h][k];
          inv[h][a] = s_;
        }
      }
    }
    __syncthreads();
    for (int h = 0; h < 2; h++) {
      const int vt = tid + 64 * h, R = vt >> 4, bcol = vt & 15, base = 16 * R;
      if (base < P.n) {
#pragma unroll
        for (int a = 1; a < 16; a++) if (a > bcol) Lqa[(base + bcol) * P.W + (a - bcol)] = inv[h][a];
      }
    }
    __syncthreads();
  }
  const long long t0 = __builtin_readcyclecounter();
  for (int rep = 0; rep < kReps; rep++) {
    if (mode == 0) solve_chain(P, La, dinv, rhs, out); else if (mode == 7) solve_chain_imm(P, La, dinv, rhs, out); else if (mode == 1 || mode == 6) solve_blocked<0>(P, Lqa, dinv, rhs, out, vbuf); else if (mode == 2) solve_blocked<1>(P, Lqa, dinv, rhs, out, vbuf);
    else if (mode == 3) solve_blocked<2>(P, Lqa, dinv, rhs, out, vbuf); else if (mode == 4) solve_blocked<4>(P, Lqa, dinv, rhs, out, vbuf); else solve_blocked<7>(P, Lqa, dinv, rhs, out, vbuf);
    __syncthreads();
  }
  const long long t1 = __builtin_readcyclecounter();
  for (int k = tid; k < P.n; k += 64) outg[k] = out[k];
  if (tid == 0) cyc[0] = t1 - t0;
}

int main() {
  const int n = 120, bw = 26, W = bw + 8, n8 = 120, n16 = 128;
  Prob P{n, bw, W, n8, n16};
  // random SPD band -> dense Cholesky on the host (long double), unit-lower L^ and D
  std::mt19937 rng(7); std::normal_distribution<double> nd(0, 1);
  std::vector<long double> K(n * n, 0.0L);
  for (int i = 0; i < n; i++) for (int j = std::max(0, i - bw); j < i; j++) { const double v = nd(rng) * 0.3; K[i * n + j] = v; K[j * n + i] = v; }
  for (int i = 0; i < n; i++) { long double s = 0; for (int j = 0; j < n; j++) s += fabsl(K[i * n + j]); K[i * n + i] = s + 1.0L; }
  std::vector<long double> L(n * n, 0.0L);
  for (int j = 0; j < n; j++) {
    long double d = K[j * n + j]; for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k];
    L[j * n + j] = sqrtl(d);
    for (int i = j + 1; i < n; i++) { long double s = K[i * n + j]; for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k]; L[i * n + j] = s / L[j * n + j]; }
  }
  std::vector<double> Lh(n * n, 0.0), dinv(n);
  for (int j = 0; j < n; j++) { dinv[j] = (double)(1.0L / (L[j * n + j] * L[j * n + j])); for (int i = j + 1; i < n; i++) Lh[i * n + j] = (double)(L[i * n + j] / L[j * n + j]); }
  const int tot = 8 + n16 * W + 64;
  std::vector<double> band(tot, 0.0), bandq(tot, 0.0);
  for (int j = 0; j < n; j++) for (int i = j + 1; i <= std::min(n - 1, j + bw); i++) band[8 + j * W + (i - j)] = Lh[i * n + j];
  bandq = band;
  for (int R = 0; R < n16 / 16; R++) {                                 // inv(L^_RR) (unit lower) in place of the diagonal block
    double M[16][16] = {}, Inv[16][16] = {};
    for (int a = 0; a < 16; a++) for (int b = 0; b < 16; b++) { const int i = 16 * R + a, j = 16 * R + b; M[a][b] = a == b ? 1.0 : (a > b && i < n && j < n ? Lh[i * n + j] : 0.0); }
    for (int b = 0; b < 16; b++) for (int a = 0; a < 16; a++) { double s = a == b ? 1.0 : 0.0; for (int k = b; k < a; k++) s -= M[a][k] * Inv[k][b]; Inv[a][b] = a < b ? 0.0 : s; }
    for (int a = 0; a < 16; a++) for (int b = 0; b < a; b++) bandq[8 + (16 * R + b) * W + (a - b)] = Inv[a][b];
  }
  std::vector<double> rhs(n); for (auto &v : rhs) v = nd(rng);
  // reference solution (long double)
  std::vector<long double> y(n), xr(n);
  for (int i = 0; i < n; i++) { long double s = rhs[i]; for (int j = std::max(0, i - bw); j < i; j++) s -= (long double)Lh[i * n + j] * y[j]; y[i] = s; }
  for (int i = n - 1; i >= 0; i--) { long double s = y[i] * dinv[i]; for (int j = i + 1; j <= std::min(n - 1, i + bw); j++) s -= (long double)Lh[j * n + i] * xr[j]; xr[i] = s; }

  double *dL, *dLq, *dd, *dr, *dout; long long *dc; int *dchk;
  CK(hipMalloc(&dL, tot * 8)); CK(hipMalloc(&dLq, tot * 8)); CK(hipMalloc(&dd, n * 8)); CK(hipMalloc(&dr, n * 8)); CK(hipMalloc(&dout, n * 8)); CK(hipMalloc(&dc, 16)); CK(hipMalloc(&dchk, 320 * 4));
  CK(hipMemcpy(dL, band.data(), tot * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dLq, bandq.data(), tot * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dd, dinv.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, rhs.data(), n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dchk); CK(hipDeviceSynchronize());
  std::vector<int> chk(320); CK(hipMemcpy(chk.data(), dchk, 320 * 4, hipMemcpyDeviceToHost));
  const char *nm[5] = {"row_ror:8 rows 1,3 (old = lane)", "row_ror:15 row 1 (old = -1)", "row_newbcast:4", "permlane16_swap(l, l): x*100+y", "permlane32_swap(l, l): x*100+y"};
  for (int t = 0; t < 5; t++) { std::printf("%-34s", nm[t]); for (int l = 0; l < 64; l += (t < 3 ? 1 : 1)) if (l < 36 || t >= 3) std::printf(" %d", chk[t * 64 + l]); std::printf("\n"); }
  const size_t lds = (size_t)(2 * tot + 4 * n16) * 8;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_solve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const char *mn[8] = {"readlane chain", "blocked by 16, DPP", "  - without the row sums (wrong)", "  - without stage 2 (wrong)", "  - without the coefficient loads (wrong)", "  - without all three (wrong)", "blocked, inverse blocks computed on the device", "readlane chain, immediate lane indices"};
  for (int mode = 0; mode < 8; mode++) {
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_solve, dim3(1), dim3(64), lds, 0, mode, P, dL, dLq, dd, dr, dout, dc, 0); CK(hipDeviceSynchronize()); }
    std::vector<double> x(n); long long c[2];
    CK(hipMemcpy(x.data(), dout, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost));
    double err = 0, nrm = 0; for (int i = 0; i < n; i++) { err = std::max(err, (double)fabsl(x[i] - xr[i])); nrm = std::max(nrm, (double)fabsl(xr[i])); }
    std::printf("%-42s: %.0f cycles per solve (n = %d, bw = %d) = %.1f per unknown and pass; max |x - x_ref| = %.2e (|x| = %.2e)\n", mn[mode], (double)c[0] / kReps, n, bw, (double)c[0] / kReps / (2.0 * n), err, nrm);
  }
  return 0;
}
