// wbdirect_hip.hip -- the ADMM iteration of the Woodbury DIRECT mode in two launches (backend.h DevWbx).
//
// Problem class (portfolio, BASELINE configs[3]; Engine::prepare_wb decides): P diagonal, every row of A either LONG (more than kLongRow
// entries; r <= kWbMaxRows of them: the factor-model rows and the budget row) or a ONE-entry row (the box on a variable), n <= kWbxMaxN.
// Then  K = P + sigma I + A' diag(rho) A = D0 + A_L' diag(rho_L) A_L  with D0 diagonal, and the reduced KKT system of an ADMM iteration
// (/root/reference/src/osqppurepy/_osqp.py:649-658 in its Schur-complement form) is solved exactly by the Woodbury identity
//     K^-1 = D0^-1 - D0^-1 A_L' S^-1 A_L D0^-1 ,   S = diag(1 / rho_L) + A_L D0^-1 A_L'      (r x r, inverted at every rho update).
// Round 3 ran that as FIVE launches per ADMM iteration (KB over B, three kernels of M^-1, KA over A), each a 1024-workgroup pass over
// ~6 MB: launch latency, not bytes -- 8 us apiece, 1575 iterations, 85 ms for the portfolio QP.
//
// Here the iteration has the TWO global reductions the algebra really needs (g = A_L D0^-1 r_0 before h = S^-1 g; z~_L = A_L x~ before
// the z / y update of the long rows), and a launch boundary at each:
//     X:  [finish iteration k]   z~_L <- sum of the workgroups' partials;  z, y, v, t0 of the long rows (every workgroup, identically;
//                                workgroup 0 stores them), of the one-entry rows of the own columns, x / dx / x_g of the own columns
//                                                                                                       (_osqp.py:660-703)
//         [start iteration k+1]  r_0 = sigma x - q - (P + sigma I) x_g + A' (v - t0)  on the own columns   (:649-650, reduced form,
//                                start residual at the extrapolated x_g as in kb_rhs);  partial  g = A_L D0^-1 r_0
//     Y:  g <- sum of partials;  h = S^-1 g;  x~ = x_g + D0^-1 (r_0 - A_L' h)  on the own columns;  partial  z~_L = A_L x~
// A workgroup owns kWbxCols consecutive columns and keeps its dense r x kWbxCols tile of A_L in LDS for both passes of a launch (one
// coalesced read of <= 64 KB; the zero entries of the tile cost bytes, not time: the launches are latency-bound).  Every sum has a
// fixed order -- no atomics: results are reproducible run to run.  Same arithmetic as the five-launch form up to the order of the sums
// (tests/test_gpu_woodbury.py compares the two).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <stdexcept>

#include "hip_common.h"              // slot record layout (SlotState, slot_read / slot_write), HIP_CHECK

namespace osqp_hip {
namespace be {

#define WBX_CHECK(expr)                                                                                       \
  do {                                                                                                        \
    hipError_t e_ = (expr);                                                                                   \
    if (e_ != hipSuccess) {                                                                                   \
      char msg_[512];                                                                                         \
      std::snprintf(msg_, sizeof(msg_), "osqp_hip: HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #expr); \
      std::fprintf(stderr, "%s\n", msg_);                                                                     \
      throw DeviceError(msg_);                                                                                \
    }                                                                                                         \
  } while (0)

namespace {

constexpr int kT = 256;                       // threads per workgroup
constexpr int kR = kWbMaxRows;                // 128: rows of the tile (r of them in use)
constexpr int kC = kWbxCols;                  // 64: columns a workgroup owns
constexpr int kStride = kC + 1;               // LDS row stride of the tile (doubles): the row pass reads down a column of lanes
static_assert(kT == 2 * kR && kT == 4 * kC, "thread layouts: (row, half) and (column, quarter)");

struct XLds {
  double tile[kR * kStride];                  // A_L[:, own columns], zero where A_L has no entry
  double wv[kR];                              // X: v - t0 of the long rows;  Y: h = S^-1 g
  double gv[kR];                              // Y: g
  double col[kC];                             // X: D0^-1 r_0 of the own columns;  Y: x~ of the own columns
  double red[4][kC];                          // column pass: one partial per quarter of the rows
  double pz[2][kR];                           // row pass / partial sums: one partial per half
  double pz4[4][kR / 2];                      // fold of the partials: odd rows, one partial per slice
};
static_assert(kC == kR / 2, "fold_partials keeps the even rows of a slice in red[s][0 .. kC)");
struct YLds : XLds { double tile2[kR * kStride]; };      // Y also holds S^-1 A_L of its columns (refreshed at every rho update)


// These launches are chains of memory round trips, not streams: every loop over global memory below first REQUESTS a batch of
// values (kFoldBatch, a whole tile; clamped addresses, no branch around a load) and only then consumes it -- one latency per batch instead of one per value.
// tile <- global (r rows of kC doubles, contiguous per workgroup), coalesced 16-byte loads; rows r .. kR - 1 read as zero
struct TileRegs { double2 v[kR * kC / 2 / kT]; };
__device__ __forceinline__ TileRegs tile_issue(const double *tiles, int r) {
  const double2 *src = reinterpret_cast<const double2 *>(tiles + (size_t)blockIdx.x * kR * kC);
  const int cnt = r * kC / 2;
  TileRegs t;
#pragma unroll
  for (int u = 0; u < kR * kC / 2 / kT; u++) { const int e = threadIdx.x + u * kT; t.v[u] = src[min(e, cnt - 1)]; }
  return t;
}
__device__ __forceinline__ void tile_store(const TileRegs &t, int r, double *dst) {
  const int cnt = r * kC / 2;
#pragma unroll
  for (int u = 0; u < kR * kC / 2 / kT; u++) {
    const int e = threadIdx.x + u * kT, a = e / (kC / 2), c = 2 * (e - a * (kC / 2));
    const bool ok = e < cnt;
    dst[a * kStride + c] = ok ? t.v[u].x : 0.0; dst[a * kStride + c + 1] = ok ? t.v[u].y : 0.0;
  }
}
// out[a] = sum over the workgroups' partials part[w][a]: thread (row pair, slice s of four) takes the workgroups w = s mod 4 with 16-byte
// loads, kFoldBatch of them in flight; the four slices are added in index order (fixed order throughout: deterministic)
constexpr int kFoldBatch = 20;
template <int kFB = kFoldBatch>
__device__ __forceinline__ void fold_partials(const double *part, int G, XLds &L, double *out) {
  const int a2 = threadIdx.x & (kR / 2 - 1), s = threadIdx.x >> 6;
  const double2 *p2 = reinterpret_cast<const double2 *>(part);
  double acc0 = 0.0, acc1 = 0.0;
  for (int w0 = s; w0 < G; w0 += 4 * kFB) {
    double2 p[kFB];
#pragma unroll
    for (int k = 0; k < kFB; k++) p[k] = p2[(size_t)min(w0 + 4 * k, G - 1) * (kR / 2) + a2];
#pragma unroll
    for (int k = 0; k < kFB; k++) { const bool ok = w0 + 4 * k < G; acc0 += ok ? p[k].x : 0.0; acc1 += ok ? p[k].y : 0.0; }
  }
  L.red[s][a2] = acc0; L.pz4[s][a2] = acc1;                  // (red: even rows, pz4: odd rows; kC = kR / 2 slots per slice)
  __syncthreads();
  if (threadIdx.x < kR / 2) {
    out[2 * a2] = (L.red[0][a2] + L.red[1][a2]) + (L.red[2][a2] + L.red[3][a2]);
    out[2 * a2 + 1] = (L.pz4[0][a2] + L.pz4[1][a2]) + (L.pz4[2][a2] + L.pz4[3][a2]);
  }
  __syncthreads();
}
// column pass: sum_a tile[a][c] vec[a] for the own columns -> thread (c, quarter 0) returns the total
__device__ __forceinline__ double column_pass(XLds &L, const double *tile, const double *vec) {
  const int c = threadIdx.x & (kC - 1), q = threadIdx.x >> 6;
  double acc0 = 0.0, acc1 = 0.0;
#pragma unroll 8
  for (int a = q * (kR / 4); a < (q + 1) * (kR / 4); a += 2) { acc0 += tile[a * kStride + c] * vec[a]; acc1 += tile[(a + 1) * kStride + c] * vec[a + 1]; }
  L.red[q][c] = acc0 + acc1;
  __syncthreads();
  return (L.red[0][c] + L.red[1][c]) + (L.red[2][c] + L.red[3][c]);
}
// row pass: part[blockIdx.x][a] = sum_c tile[a][c] col[c]
__device__ __forceinline__ void row_pass(XLds &L, double *part) {
  const int a = threadIdx.x & (kR - 1), s = threadIdx.x >> 7;
  double acc0 = 0.0, acc1 = 0.0;
#pragma unroll 8
  for (int c = s * (kC / 2); c < (s + 1) * (kC / 2); c += 2) { acc0 += L.tile[a * kStride + c] * L.col[c]; acc1 += L.tile[a * kStride + c + 1] * L.col[c + 1]; }
  L.pz[s][a] = acc0 + acc1;
  __syncthreads();
  if (threadIdx.x < kR) part[(size_t)blockIdx.x * kR + a] = L.pz[0][a] + L.pz[1][a];
}
// the z / y update of one row (_osqp.py:682-703) and the quantities the next right-hand side needs; returns v - t0
struct RowState { double z, y, zt; };
__device__ __forceinline__ double row_update(const Dev &d, int i, double ztil, const RowState in, RowState &out, bool store) {
  const double rho = d.rho[i], rinv = d.rho_inv[i], lo = d.l[i], up = d.u[i];
  const double zr = d.alpha * ztil + (1.0 - d.alpha) * in.z;                  // :686-690
  const double zn = fmin(fmax(zr + rinv * in.y, lo), up);                      // :674
  const double dyi = rho * (zr - zn), yn = in.y + dyi;                         // :698-703
  const double zg = ztil + d.theta * (ztil - in.zt);                           // A x_g (Dev::ztg)
  const double v = rho * zn - yn, t0 = rho * zg;
  out = RowState{zn, yn, ztil};
  if (store) { d.y[i] = yn; d.dy[i] = dyi; d.z[i] = zn; d.zt[i] = ztil; d.v[i] = v; d.ztg[i] = zg; d.t0[i] = t0; }
  return v - t0;
}

// X.  upd: finish an ADMM iteration (needs Y's partials);  rhs: start the next one
template <bool UPD, bool RHS>
__device__ __forceinline__ void wbx_x_body(const Dev &d, XLds &L) {
  const DevWbx &x = d.wb.x;
  const int tid = threadIdx.x, r = d.wb.r, G = x.G, j0 = blockIdx.x * kC, n = d.n;
  TileRegs tr;
  if (RHS) tr = tile_issue(x.tile, r);                        // requested first: consumed after the fold of the partials
  // ---- long rows
  if (UPD) {
    fold_partials(x.partZ, G, L, L.gv);                 // z~ of the long rows
    if (tid < kR) {
      double w = 0.0;
      if (tid < r) {
        const int i = d.wb.rows[tid];
        const RowState in{x.ls0[3 * tid], x.ls0[3 * tid + 1], x.ls0[3 * tid + 2]};
        RowState out;
        w = row_update(d, i, L.gv[tid], in, out, blockIdx.x == 0);
        if (blockIdx.x == 0) { x.ls1[3 * tid] = out.z; x.ls1[3 * tid + 1] = out.y; x.ls1[3 * tid + 2] = out.zt; }
      }
      L.wv[tid] = w;
    }
  } else if (tid < kR) {
    double w = 0.0;
    if (tid < r) {
      const int i = d.wb.rows[tid];
      w = d.v[i] - d.t0[i];
      if (blockIdx.x == 0) { x.ls1[3 * tid] = d.z[i]; x.ls1[3 * tid + 1] = d.y[i]; x.ls1[3 * tid + 2] = d.zt[i]; }
      if (blockIdx.x == 0 && x.one) { x.lz0[4 * tid] = d.z[i]; x.lz0[4 * tid + 1] = d.y[i]; x.lz0[4 * tid + 2] = d.zt[i]; x.lz0[4 * tid + 3] = d.ztg[i]; }      // (one-launch form: what its first launch reads)
    }
    L.wv[tid] = w;
  }
  // ---- own columns: one-entry rows and the x update (thread (c, quarter 0))
  const int c = tid & (kC - 1), q = tid >> 6, j = j0 + c;
  const bool own = q == 0 && j < n;
  double ssum = 0.0, xj = 0.0, xgj = 0.0, qj = 0.0, bjj = 0.0;
  if (own) {
    xj = d.x[j]; xgj = d.xg[j];
    if (RHS) { qj = d.q[j]; bjj = x.bjj[j]; }
    if (UPD) {
      const double xt = d.xs[j], xn = d.alpha * xt + (1.0 - d.alpha) * xj;     // :664-668
      d.dx[j] = xn - xj; d.x[j] = xn; xj = xn;
      xgj = xt + d.theta * (xt - d.xsp[j]); d.xg[j] = xgj; d.xsp[j] = xt;       // next start (Dev::xg)
      for (int k = x.sc_ptr[j]; k < x.sc_ptr[j + 1]; k++) {
        const int i = x.sc_row[k]; const double av = x.sc_val[k];
        const RowState in{d.z[i], d.y[i], d.zt[i]};
        RowState out;
        const double w = row_update(d, i, av * xt, in, out, true);
        if (RHS) ssum += av * w;
      }
    } else if (RHS) {
      for (int k = x.sc_ptr[j]; k < x.sc_ptr[j + 1]; k++) { const int i = x.sc_row[k]; ssum += x.sc_val[k] * (d.v[i] - d.t0[i]); }
    }
  }
  if (UPD && blockIdx.x == 0 && tid == 0) {             // PCG statistics of the finished iteration: one exact step
    d.flags[F_STAT_SUM] += 1; d.flags[F_STAT_SUMSQ] += 1; d.flags[F_STAT_N] += 1;
    if (d.flags[F_STAT_MAX] < 1) d.flags[F_STAT_MAX] = 1;
  }
  if (!RHS) return;
  tile_store(tr, r, L.tile);
  __syncthreads();                                       // tile, wv
  // ---- r_0 on the own columns, D0^-1 r_0, partial g
  const double lsum = column_pass(L, L.tile, L.wv);
  if (q == 0) {
    double yv = 0.0;
    if (j < n) {
      const double r0 = d.sigma * xj - qj - bjj * xgj + ssum + lsum;
      d.r[j] = r0; yv = d.wb.Dinv0[j] * r0;
    }
    L.col[c] = yv;
  }
  __syncthreads();
  row_pass(L, x.partG);
}

template <bool UPD, bool RHS>
__global__ __launch_bounds__(kT) void k_wbx_x(Dev d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  wbx_x_body<UPD, RHS>(d, *reinterpret_cast<XLds *>(smem));
}

// Y.  (A_L' S^-1 g is taken as (S^-1 A_L)' g with the second tile: no r x r product, no S^-1 traffic on the iteration's critical path)
__device__ __forceinline__ void wbx_y_body(const Dev &d, YLds &L) {
  const DevWbx &x = d.wb.x;
  const int tid = threadIdx.x, r = d.wb.r, G = x.G, j0 = blockIdx.x * kC, n = d.n;
  const TileRegs tr = tile_issue(x.tile, r), tr2 = tile_issue(x.tile2, r);
  const int c = tid & (kC - 1), q = tid >> 6, j = j0 + c;
  double rj = 0.0, dj = 0.0, xgy = 0.0;
  if (q == 0 && j < n) { rj = d.r[j]; dj = d.wb.Dinv0[j]; xgy = d.xg[j]; }
  fold_partials(x.partG, G, L, L.gv);                    // g (rows >= r: zero partials)
  tile_store(tr, r, L.tile); tile_store(tr2, r, L.tile2);
  __syncthreads();
  const double s = column_pass(L, L.tile2, L.gv);        // (A_L' S^-1 g) on the own columns
  if (q == 0) {
    double xt = 0.0;
    if (j < n) {
      const double u = dj * (rj - s);
      xt = xgy + u;
      d.xs[j] = xt; d.uu[j] = u;
    }
    L.col[c] = xt;
  }
  __syncthreads();
  row_pass(L, x.partZ);                                  // partial z~ of the long rows
  if (blockIdx.x == 0) {
    if (tid < 3 * r) x.ls0[tid] = x.ls1[tid];            // the long rows' state X will read: handed over outside X (no workgroup of X reads what X writes)
    if (tid + kT < 3 * r) x.ls0[tid + kT] = x.ls1[tid + kT];
    if (tid == 0) { d.flags[F_DONE] = 1; d.flags[F_ITERS] = 1; }
  }
}

__global__ __launch_bounds__(kT) void k_wbx_y(Dev d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  wbx_y_body(d, *reinterpret_cast<YLds *>(smem));
}

// Z: ONE launch per ADMM iteration (backend.h DevWbx::one).  par: parity of the launch in its chunk -- partials of g are read from (par ? partZ : partG)
// and written to the other, the long rows' state is read from (par ? lz1 : lz0) and written to the other.  RHS = false: the chunk's last launch.
struct ZLds : XLds { double hv[kR]; };
struct SinvRegs { double2 v[kR / 4]; };                      // thread (row a = tid mod kR, half = tid / kR): 64 entries of row a of S^-1
__device__ __forceinline__ SinvRegs sinv_issue(const double *sinvp) {
  const int a = threadIdx.x & (kR - 1), hf = threadIdx.x >> 7;
  const double2 *src = reinterpret_cast<const double2 *>(sinvp + (size_t)a * kR + hf * (kR / 2));
  SinvRegs s;
#pragma unroll
  for (int u = 0; u < kR / 4; u++) s.v[u] = src[u];
  return s;
}
template <bool RHS>
__device__ __forceinline__ void wbz_body(const Dev &d, ZLds &L, const int par) {
  const DevWbx &x = d.wb.x;
  const int tid = threadIdx.x, r = d.wb.r, G = x.G, j0 = blockIdx.x * kC, n = d.n;
  const double *pin = par ? x.partZ : x.partG; double *pout = par ? x.partG : x.partZ;
  const double *lin = par ? x.lz1 : x.lz0; double *lout = par ? x.lz0 : x.lz1;
  const TileRegs tr = tile_issue(x.tile, r);                  // requested first: consumed after the fold
  const SinvRegs sr = sinv_issue(x.sinvp);
  const int c = tid & (kC - 1), q = tid >> 6, j = j0 + c;
  const bool own = q == 0 && j < n;
  double rj = 0.0, dj = 0.0, xgj = 0.0, xj = 0.0, xspj = 0.0, qj = 0.0, bjj = 0.0;
  if (own) { rj = d.r[j]; dj = d.wb.Dinv0[j]; xgj = d.xg[j]; xj = d.x[j]; xspj = d.xsp[j]; if (RHS) { qj = d.q[j]; bjj = x.bjj[j]; } }
  fold_partials(pin, G, L, L.gv);                             // g (rows >= r: zero partials)
  { // h = S^-1 g: two half rows per row, S^-1 from registers, g broadcast from LDS
    const int a = tid & (kR - 1), hf = tid >> 7;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int u = 0; u < kR / 4; u++) { acc0 += sr.v[u].x * L.gv[hf * (kR / 2) + 2 * u]; acc1 += sr.v[u].y * L.gv[hf * (kR / 2) + 2 * u + 1]; }
    L.pz[hf][a] = acc0 + acc1;
  }
  __syncthreads();
  if (tid < kR) {                                             // the long rows, identically in every workgroup (workgroup 0 stores)
    double w = 0.0, h = 0.0;
    if (tid < r) {
      h = L.pz[0][tid] + L.pz[1][tid];
      const int i = d.wb.rows[tid];
      const RowState in{lin[4 * tid], lin[4 * tid + 1], lin[4 * tid + 2]};
      const double ztil = lin[4 * tid + 3] + h * d.rho_inv[i];      // z~ = A x_g + h / rho
      RowState out;
      w = row_update(d, i, ztil, in, out, blockIdx.x == 0);
      if (blockIdx.x == 0) { lout[4 * tid] = out.z; lout[4 * tid + 1] = out.y; lout[4 * tid + 2] = out.zt; lout[4 * tid + 3] = ztil + d.theta * (ztil - in.zt); }
    }
    L.wv[tid] = w; L.hv[tid] = h;
  }
  tile_store(tr, r, L.tile);
  __syncthreads();                                            // tile, wv, hv
  const double s = column_pass(L, L.tile, L.hv);              // (A_L' h) on the own columns
  double ssum = 0.0, xn = xj, xgn = xgj;
  if (own) {
    const double u = dj * (rj - s), xt = xgj + u;             // x~ = x_g + D0^-1 (r_0 - A_L' h)
    d.xs[j] = xt; d.uu[j] = u;
    xn = d.alpha * xt + (1.0 - d.alpha) * xj;                 // :664-668
    d.dx[j] = xn - xj; d.x[j] = xn;
    xgn = xt + d.theta * (xt - xspj); d.xg[j] = xgn; d.xsp[j] = xt;
    // (tried: the long rows' and the first one-entry row's constants and state requested at the head, under the fold's round trips: 11.8 -> 12.3 us;
    //  all 160 partials of the fold in flight at once: scratch, 11.9 us -- the launch is bound by its barriers and the fold's two round trips)
    for (int k = x.sc_ptr[j]; k < x.sc_ptr[j + 1]; k++) {
      const int i = x.sc_row[k]; const double av = x.sc_val[k];
      const RowState in{d.z[i], d.y[i], d.zt[i]};
      RowState out;
      const double w = row_update(d, i, av * xt, in, out, true);
      if (RHS) ssum += av * w;
    }
  }
  if (blockIdx.x == 0 && tid == 0) {                          // PCG statistics of the finished iteration: one exact step
    d.flags[F_STAT_SUM] += 1; d.flags[F_STAT_SUMSQ] += 1; d.flags[F_STAT_N] += 1;
    if (d.flags[F_STAT_MAX] < 1) d.flags[F_STAT_MAX] = 1;
    d.flags[F_DONE] = 1; d.flags[F_ITERS] = 1;
  }
  if (!RHS) return;
  __syncthreads();                                            // (column_pass reuses its partial buffer)
  const double lsum = column_pass(L, L.tile, L.wv);           // (A_L' (v - t0)) on the own columns
  if (q == 0) {
    double yv = 0.0;
    if (j < n) {
      const double r0 = d.sigma * xn - qj - bjj * xgn + ssum + lsum;
      d.r[j] = r0; yv = dj * r0;
    }
    L.col[c] = yv;
  }
  __syncthreads();
  row_pass(L, pout);
}
template <bool RHS>
__global__ __launch_bounds__(kT) void k_wbz(Dev d, int par) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  wbz_body<RHS>(d, *reinterpret_cast<ZLds *>(smem), par);
}
// sinvp <- S^-1, rows padded to kR (after every inversion of S)
__global__ __launch_bounds__(kT) void k_wbz_pad(Dev d, int cond) {
  if (cond && !d.ctl->rho_flag) return;
  const int r = d.wb.r;
  for (int e = blockIdx.x * kT + threadIdx.x; e < kR * kR; e += gridDim.x * kT) { const int a = e / kR, b = e - a * kR; d.wb.x.sinvp[e] = (a < r && b < r) ? d.wb.Sinv[(size_t)a * r + b] : 0.0; }
}

// Slot forms (device-side scheduling, pcg_hip.hip "slot kernels"): the same two launches as a pair of slots.  X reads record A and
// writes record B, Y the other way round; P_KB = an iteration may start (right-hand side), P_K1 = Y is due, P_KA = Y has run, the
// update is pending.  A chunk of N iterations takes N + 1 pairs (the last X only updates, the last Y passes the record on); launches
// behind a finished chunk idle, every launch counts itself in SR_SEQ -- what is computed does not depend on how many are enqueued.
__global__ __launch_bounds__(kT) void k_wbx_slot_x(Dev d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  XLds &L = *reinterpret_cast<XLds *>(smem);
  SlotState st = slot_read(d.slot);
  int *W = d.slot + SR_WORDS;
  const bool upd = st.ph == P_KA;
  const int admm_after = st.admm + (upd ? 1 : 0);
  const bool rhs = (upd || st.ph == P_KB) && admm_after < st.target;
  if (upd && rhs) wbx_x_body<true, true>(d, L);
  else if (upd) wbx_x_body<true, false>(d, L);
  else if (rhs) wbx_x_body<false, true>(d, L);
  if (upd) {
    if (d.ctl && admm_after >= st.target && blockIdx.x == 0 && threadIdx.x == 0) { d.ctl->seq_end = st.seq + 1; d.ctl->chunk_done = 1; }      // (read by LATER launches)
    st.admm = admm_after; st.used = 1; st.conv = 1;
  }
  st.ph = rhs ? P_K1 : (st.ph == P_IDLE ? P_IDLE : P_KB);      // (P_KB with admm >= target: the chunk is over, the next X idles)
  slot_write(W, st);
}
__global__ __launch_bounds__(kT) void k_wbx_slot_y(Dev d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  SlotState st = slot_read(d.slot + SR_WORDS);
  if (st.ph == P_K1) { wbx_y_body(d, *reinterpret_cast<YLds *>(smem)); st.ph = P_KA; }
  slot_write(d.slot, st);
}

// tile2 <- S^-1 tile (per column block; after every inversion of S): thread (c, quarter q) forms the rows a = q mod 4 of its column
__global__ __launch_bounds__(kT) void k_wbx_t2(Dev d, int cond) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (cond && !d.ctl->rho_flag) return;
  XLds &L = *reinterpret_cast<XLds *>(smem);
  const DevWbx &x = d.wb.x;
  const int r = d.wb.r, c = threadIdx.x & (kC - 1), q = threadIdx.x >> 6;
  tile_store(tile_issue(x.tile, r), r, L.tile);
  __syncthreads();
  double *out = x.tile2 + (size_t)blockIdx.x * kR * kC;
  for (int a = q; a < r; a += 4) {
    const double *srow = d.wb.Sinv + (size_t)a * r;      // (workgroup-uniform per a: scalar loads)
    double acc0 = 0.0, acc1 = 0.0;
    int b = 0;
    for (; b + 1 < r; b += 2) { acc0 += srow[b] * L.tile[b * kStride + c]; acc1 += srow[b + 1] * L.tile[(b + 1) * kStride + c]; }
    if (b < r) acc0 += srow[b] * L.tile[b * kStride + c];
    out[(size_t)a * kC + c] = acc0 + acc1;
  }
}

// tiles and one-entry rows <- A.val (after assembly / equilibration / matrix updates)
__global__ __launch_bounds__(kT) void k_wbx_fill(Dev d) {
  const DevWbx &x = d.wb.x;
  const DevCsr &AL = d.wb.AL;
  const int stride = gridDim.x * kT;
  for (int k = blockIdx.x * kT + threadIdx.x; k < AL.nnz; k += stride) {
    // row of entry k: binary search in the r + 1 row pointers
    int lo = 0, hi = d.wb.r;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (AL.rowptr[mid] <= k) lo = mid; else hi = mid; }
    const int j = AL.col[k];
    x.tile[((size_t)(j / kC) * kR + lo) * kC + (j % kC)] = d.A.val[d.wb.al_src[k]];
  }
  for (int k = blockIdx.x * kT + threadIdx.x; k < x.nsc; k += stride) x.sc_val[k] = d.A.val[x.sc_src[k]];
  for (int j = blockIdx.x * kT + threadIdx.x; j < d.n; j += stride) x.bjj[j] = d.B.val[d.Bdiag[j]];      // P_jj + sigma (P is diagonal)
}

}  // namespace

bool wbx_supported() { return true; }
void wbx_refresh(Dev &d) {
  if (!d.wb.on || !d.wb.x.on) return;
  WBX_CHECK(hipSetDevice(d.device));
  hipLaunchKernelGGL(k_wbx_fill, dim3(256), dim3(kT), 0, static_cast<hipStream_t>(d.stream), d);
  WBX_CHECK(hipGetLastError());
}
void wbx_factor(Dev &d, int cond) {            // after S^-1 has changed (rho update): tile2 = S^-1 A_L;  cond: only when the boundary group updated rho
  if (!d.wb.on || !d.wb.x.on) return;
  WBX_CHECK(hipSetDevice(d.device));
  hipLaunchKernelGGL(k_wbx_t2, dim3(d.wb.x.G), dim3(kT), sizeof(XLds), static_cast<hipStream_t>(d.stream), d, cond);
  if (d.wb.x.one) hipLaunchKernelGGL(k_wbz_pad, dim3(16), dim3(kT), 0, static_cast<hipStream_t>(d.stream), d, cond);
  WBX_CHECK(hipGetLastError());                // (these kernels need 75-142 KB of dynamic LDS, granted by wbx_init: a refused launch must not pass silently)
}
void wbx_slot_pair(Dev &d) {
  hipStream_t s = static_cast<hipStream_t>(d.stream);
  const dim3 grid(d.wb.x.G), block(kT);
  hipLaunchKernelGGL(k_wbx_slot_x, grid, block, sizeof(XLds), s, d);
  hipLaunchKernelGGL(k_wbx_slot_y, grid, block, sizeof(YLds), s, d);
  WBX_CHECK(hipGetLastError());
}
void wbx_init(Dev &d) {                        // (more than the default 64 KB of dynamic LDS: gfx950 has 160 KB per CU, one workgroup per CU here)
  WBX_CHECK(hipSetDevice(d.device));
  const int lds = (int)sizeof(XLds);
  WBX_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wbx_x<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  WBX_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wbx_x<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  WBX_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wbx_x<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  WBX_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wbx_t2), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  WBX_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wbx_y), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(YLds)));
  WBX_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wbx_slot_x), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  WBX_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wbz<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ZLds)));
  WBX_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wbz<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ZLds)));
  WBX_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wbx_slot_y), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(YLds)));
}
// One chunk of `niter` ADMM iterations:  X(rhs), { Y, X(update + rhs) } x (niter - 1), Y, X(update)  -- 2 niter + 1 launches (two-launch form)
void wbx_chunk(Dev &d, int niter) {
  if (niter <= 0) return;
  WBX_CHECK(hipSetDevice(d.device));
  hipStream_t s = static_cast<hipStream_t>(d.stream);
  const size_t lds = sizeof(XLds);
  const dim3 grid(d.wb.x.G), block(kT);
  hipLaunchKernelGGL((k_wbx_x<false, true>), grid, block, lds, s, d);
  if (d.wb.x.one && !d.wb.x.slots) {             // one launch per ADMM iteration: X(rhs), Z x (niter - 1), Z(last) -- niter + 1 launches
    for (int it = 1; it < niter; it++) hipLaunchKernelGGL((k_wbz<true>), grid, block, sizeof(ZLds), s, d, (it - 1) & 1);
    hipLaunchKernelGGL((k_wbz<false>), grid, block, sizeof(ZLds), s, d, (niter - 1) & 1);
    WBX_CHECK(hipGetLastError());
    return;
  }
  for (int it = 1; it < niter; it++) {
    hipLaunchKernelGGL(k_wbx_y, grid, block, sizeof(YLds), s, d);
    hipLaunchKernelGGL((k_wbx_x<true, true>), grid, block, lds, s, d);
  }
  hipLaunchKernelGGL(k_wbx_y, grid, block, sizeof(YLds), s, d);
  hipLaunchKernelGGL((k_wbx_x<true, false>), grid, block, lds, s, d);
  WBX_CHECK(hipGetLastError());
}

}  // namespace be
}  // namespace osqp_hip
