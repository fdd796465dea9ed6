// hip_common.h -- what the HIP translation units of the engine share: the error macro, the per-handle Impl block, the device helpers
// (DPP wave reductions, per-workgroup partials, the CSR-stream row processing with its LDS layouts and hooks), the few functors and the
// SpMV probe kernel used by more than one unit, the slot-record layout, and the launch macro.  Everything lives in an anonymous
// namespace: each unit compiles its own copy (templates and forceinline device functions; no relocatable device code needed).
//   backend_hip.hip    handle / memory / graphs, residual + boundary kernels, rho / preconditioner / init, vector updates, assembly + Ruiz
//   pcg_hip.hip        the ADMM / PCG hot path: KB, K1, K2, Kv, K2F, K1F, KA, the slot kernels, the one-launch form (F1), the timing probes
//   woodbury_hip.hip   Woodbury-corrected preconditioner (both forms), the on-demand dense libraries
//   wbdirect_hip.hip   the Woodbury direct mode in two launches per ADMM iteration (self-contained)
//   batch_hip.hip      one workgroup per small QP (self-contained)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <type_traits>
#include <vector>
#include <cmath>

#include "../../include/osqp_hip.h"
#include "backend.h"
#include "policy.h"

namespace osqp_hip {
namespace be {

#define HIP_CHECK(expr)                                                                                       \
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

// partial-reduction slots inside Dev::part (each kGrid doubles)
// Diagnostic build (-DOSQP_HIP_KTRACE, tools/ktrace.py): lane 0 of every workgroup stamps the 100 MHz wall clock at a few
// phase boundaries; read back with be::ktrace_read.  Compiles to nothing in the product library.
#ifdef OSQP_HIP_KTRACE
constexpr int kTraceSlots = 16;
__device__ unsigned long long g_ktrace[kGrid * kTraceSlots];
#define KT(p) do { if (threadIdx.x == 0) g_ktrace[blockIdx.x * kTraceSlots + (p)] = wall_clock64(); } while (0)
#else
#define KT(p) do { } while (0)
#endif
// Ablation builds (-DOSQP_HIP_KNOCK=mask, timing experiments only -- results are WRONG): which phases of the windowed k_k2f cost what
#ifndef OSQP_HIP_KNOCK
#define OSQP_HIP_KNOCK 0
#endif
#define KNOCKED(bit) ((OSQP_HIP_KNOCK & (bit)) != 0)
enum Slot { SL_GAMMA0 = 0, SL_GAMMA1, SL_RN0, SL_RN1, SL_BN, SL_DELTA, SL_DELTA1 /* F1 form: delta by parity (SL_DELTA + (k & 1)) */, SL_RES0 /* .. SL_RES0 + R_COUNT - 1 */ };
static_assert(SL_RES0 + R_COUNT <= kPartSlots, "Dev::part is too small");

struct Impl {
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_ext = nullptr, ev_wait = nullptr;      // ev_ext: end of a batch kernel on a caller's stream (ext_record / ext_wait); ev_wait: stream_wait
  hipEvent_t ev_s0 = nullptr, ev_s1 = nullptr;   // around one osqp_solve on the solver's stream (ev_mark / ev_ms)
  bool ext_pending = false;
  double *pin_res = nullptr;
  int *pin_flags = nullptr;      // [F_COUNT + 16]: the flags block followed by the two slot records
  hipStream_t side = nullptr;    // slot_poll(): reads the slot records while the chunk's launches are still running on d.stream
  int *pin_poll = nullptr;       // [kSlotInts]
  Ctl *pin_ctl = nullptr, *pin_ctl2 = nullptr;   // staging of the state block (upload / poll + download)
  int epoch = 0;                 // chunks begun (slot_begin); the device copy sits behind the two records
  void *blas = nullptr;          // rocblas_handle of the large-rank Woodbury factorisation (created on first use)
  // device-resident copy of Dev for the F1 slot kernel (pcg_hip.hip dev_publish): {head block, Dev}, and what was last uploaded
  void *dev_block = nullptr; unsigned char *pin_block = nullptr; unsigned char *shadow_block = nullptr;
};
inline Impl &im(Dev &d) { return *static_cast<Impl *>(d.impl); }
inline hipStream_t st(Dev &d) { return static_cast<hipStream_t>(d.stream); }

// ---------------------------------------------------------------------------------------------- device helpers
// A pointer that was LOADED from memory (the F1 slot kernel reads Dev from a device-resident copy) has no known address space: the compiler would
// use flat_* instructions for every access through it (both memory counters, slower issue).  Every one of them points to device global memory:
// gptr() says so in the TYPE -- a pointer into the global address space; indexing and arithmetic work as usual, the accesses are global_* ones.
#if defined(__HIP_DEVICE_COMPILE__)
#define OSQP_GLOBAL_AS __attribute__((address_space(1)))
#else
#define OSQP_GLOBAL_AS
#endif
template <class T> using gp = OSQP_GLOBAL_AS T *;
template <class T> __device__ __forceinline__ gp<T> gptr(T *p) { return (gp<T>)p; }
__device__ __forceinline__ double nanmax(double r, double a) { return (a > r || a != a) ? a : r; }
// Wave64 reductions with DPP moves (VALU rate).  HIP's __shfl_* compile to ds_bpermute_b32 -- an LDS round trip of ~100+ cycles
// per 32-bit half and step: the three block reductions of a PCG kernel cost ~1 us each that way (tools/ktrace.py: 1.07 us in the
// late hook of k_k2f, 0.9 us in k_k1f's exit).  dpp<CTRL, ROWS>(v): v of the DPP source lane, 0.0 where there is none / the row is
// masked (0 = identity of the sums and of the maxima of magnitudes taken here).  The wave's result ends up in LANE 63.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
constexpr int kReduceLane = 63;          // the lane that holds a wave_sum / wave_max result
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp<0xb1>(v);            // quad_perm [1,0,3,2]
  v += dpp<0x4e>(v);            // quad_perm [2,3,0,1]: every lane holds its quad's total
  v += dpp<0x114>(v);           // row_shr:4
  v += dpp<0x118>(v);           // row_shr:8: lane 15 of every row of 16 holds the row's total
  v += dpp<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
  v += dpp<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's total
  return v;
}
__device__ __forceinline__ double wave_max(double v) {                 // of non-negative values (or NaN)
  v = nanmax(v, dpp<0xb1>(v)); v = nanmax(v, dpp<0x4e>(v)); v = nanmax(v, dpp<0x114>(v)); v = nanmax(v, dpp<0x118>(v));
  v = nanmax(v, dpp<0x142, 0xa>(v)); v = nanmax(v, dpp<0x143, 0xc>(v));
  return v;
}
constexpr int kWaves = kBlock / 64;      // block reductions: one value per wave through LDS; sred needs 2 * kWaves doubles
__device__ __forceinline__ double sred_sum(const double *s) { double t = 0; for (int w = 0; w < kWaves; w += 2) t += s[w] + s[w + 1]; return t; }
__device__ __forceinline__ double sred_max(const double *s) { double t = s[0]; for (int w = 1; w < kWaves; w++) t = nanmax(t, s[w]); return t; }
// all threads receive the block total
__device__ __forceinline__ double block_sum(double v, double *sred) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == kReduceLane) sred[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = sred_sum(sred);
  __syncthreads();
  return t;
}
__device__ __forceinline__ double block_max(double v, double *sred) {
  v = wave_max(v);
  if ((threadIdx.x & 63) == kReduceLane) sred[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = sred_max(sred);
  __syncthreads();
  return t;
}
// two quantities behind ONE barrier pair
__device__ __forceinline__ void block_sum2(double &a, double &b, double *sred) {
  a = wave_sum(a); b = wave_sum(b);
  if ((threadIdx.x & 63) == kReduceLane) { sred[threadIdx.x >> 6] = a; sred[kWaves + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  a = sred_sum(sred); b = sred_sum(sred + kWaves);
  __syncthreads();
}
__device__ __forceinline__ void block_max2(double &a, double &b, double *sred) {
  a = wave_max(a); b = wave_max(b);
  if ((threadIdx.x & 63) == kReduceLane) { sred[threadIdx.x >> 6] = a; sred[kWaves + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  a = sred_max(sred); b = sred_max(sred + kWaves);
  __syncthreads();
}
// a = sum, b = max, ONE barrier pair
__device__ __forceinline__ void block_sum_max(double &a, double &b, double *sred) {
  a = wave_sum(a); b = wave_max(b);
  if ((threadIdx.x & 63) == kReduceLane) { sred[threadIdx.x >> 6] = a; sred[kWaves + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  a = sred_sum(sred); b = sred_max(sred + kWaves);
  __syncthreads();
}
// per-thread slices of the kGrid partials of a slot: issue the loads early, reduce later.  Every lane reads kPart
// CONSECUTIVE partials with 16-byte loads where it can (half the requests of strided 8-byte ones: +3.5 %); with fewer
// partials than threads the first kGrid lanes read one each.
constexpr int kPart = kGrid >= kBlock ? kGrid / kBlock : 1;
static_assert(kGrid >= kBlock ? kGrid % kBlock == 0 : kBlock % kGrid == 0, "kGrid and kBlock must divide one another");
static_assert(kWaves % 2 == 0, "block reductions pair the waves");
struct PartRegs { double v[kPart]; };
template <class P>
__device__ __forceinline__ PartRegs partial_load(P slot) {      // P: const double * or its global-address-space form (gptr)
  PartRegs r;
  if (kPart % 2 == 0) {
    typedef typename std::conditional<std::is_same<P, gp<const double>>::value || std::is_same<P, gp<double>>::value, gp<const double2>, const double2 *>::type P2;
    P2 s2 = (P2)slot + (kPart / 2) * threadIdx.x;
#pragma unroll
    for (int k = 0; k < kPart / 2; k++) { const double2 a = s2[k]; r.v[2 * k] = a.x; r.v[2 * k + 1] = a.y; }
  } else {
    r.v[0] = (kGrid >= kBlock || (int)threadIdx.x < kGrid) ? slot[threadIdx.x] : 0.0;     // 0: identity of both folds (maxima are of magnitudes)
  }
  return r;
}
__device__ __forceinline__ double partial_fold_sum(const PartRegs &r) { double v = 0; for (int k = 0; k < kPart; k++) v += r.v[k]; return v; }
__device__ __forceinline__ double partial_fold_max(const PartRegs &r) { double v = 0; for (int k = 0; k < kPart; k++) v = nanmax(v, r.v[k]); return v; }
__device__ __forceinline__ double partial_sum(const double *slot, double *sred) { return block_sum(partial_fold_sum(partial_load(slot)), sred); }
__device__ __forceinline__ double partial_max(const double *slot, double *sred) { return block_max(partial_fold_max(partial_load(slot)), sred); }
// A result store of a latency-bound launch (two-kernel form, one-launch Woodbury form): written through the XCD's L2 (relaxed agent-scope store) when Dev::wt says the working set sits in the Infinity
// Cache -- pcg_hip.hip gst_ (F1 form, where the switch is a template parameter); here a wave-uniform branch per store.
__device__ __forceinline__ void stw(int wt, double *p, double v) {
  if (wt) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
// KB ------------------------------------------------------------------------------------------
template <class P>
__device__ __forceinline__ void put_partial(P part, int slot, double v) {
  if (threadIdx.x == 0) part[slot * kGrid + blockIdx.x] = v;
}

// CSR-stream / CSR-vector row processing shared by every sparse kernel.
//   G: gather functor   void operator()(int col, double val, double (&prod)[NS])
//   E: row epilogue     void prefetch(int row)                                (optional loads issued before the barrier)
//                       void operator()(int row, const double (&sum)[NS])    (called by exactly one lane per row)
//   Pre: bool pre()     block-uniform hook run ONCE, after the first row block's streaming loads have been issued (so
//                       whatever it waits for -- a reduction of partials, a flag -- overlaps those loads); returning
//                       false abandons the kernel for this workgroup.
template <int NS, int NBUF = (NS == 1 ? 2 : 1)>
struct StreamLds { static constexpr int kBuf = NBUF; static constexpr bool kWin = false; double prod[NBUF][NS][kChunk]; double red[3 * kWaves]; };
// Variant for kernels whose gather functor can stage a block's input-vector window in LDS (DevCsr::blkwin): single product
// buffer + the window (T = what one column contributes: a double, or a 16-byte pair).  <= 40 KB: four workgroups per CU.
template <int NS, class T>
struct StreamLdsW { static constexpr int kBuf = 1; static constexpr bool kWin = true; double prod[1][NS][kChunk]; T win[kWinCap]; double red[3 * kWaves]; };
struct NoPre { [[maybe_unused]] static constexpr int kTraceBase = 0; __device__ __forceinline__ bool operator()() const { return true; } };

__device__ __forceinline__ bool wg_has_rows(const DevCsr &M) {      // same mapping as process_rows
  const int per = (M.nblk + 7) >> 3;
  const int sl = blockIdx.x >> 3;
  return sl < per && (int)(blockIdx.x & 7) * per + sl < M.nblk;
}
// Optional two-phase forms (detected by a nested type), which let process_rows issue every load in the order it is needed
// -- the memory counter retires loads in issue order, so a wait for a late-issued load drains everything before it:
//   G:   using Ops;  Ops fetch(int col) const;                 the gathered operand(s), requested as soon as col arrives
//                    void prod(const Ops &, double val, double (&prod)[NS]) const;   evaluated after the hook
//   Pre: using Tok;  Tok begin() const;                        the hook's own loads, requested BEFORE the matrix loads
//                    bool finish(const Tok &) const;           the rest of the hook (runs while the gathers are in flight)
template <class T, class = void> struct has_ops : std::false_type {};
template <class T> struct has_ops<T, std::void_t<typename T::Ops>> : std::true_type {};
template <class T, class = void> struct has_tok : std::false_type {};
template <class T> struct has_tok<T, std::void_t<typename T::Tok>> : std::true_type {};
template <class G, bool = has_ops<G>::value> struct GatherOps {
  struct Ops {};
  static __device__ __forceinline__ Ops fetch(const G &, int) { return Ops(); }
  template <int NS> static __device__ __forceinline__ void prod(const G &g, const Ops &, int c, double a, double (&pr)[NS]) { g(c, a, pr); }
};
template <class G> struct GatherOps<G, true> {
  using Ops = typename G::Ops;
  static __device__ __forceinline__ Ops fetch(const G &g, int c) { return g.fetch(c); }
  template <int NS> static __device__ __forceinline__ void prod(const G &g, const Ops &o, int, double a, double (&pr)[NS]) { g.prod(o, a, pr); }
};
template <class P, bool = has_tok<P>::value> struct PreOps {
  struct Tok {};
  static __device__ __forceinline__ Tok begin(const P &) { return Tok(); }
  static __device__ __forceinline__ bool finish(const P &p, const Tok &) { return p(); }
};
template <class P> struct PreOps<P, true> {
  using Tok = typename P::Tok;
  static __device__ __forceinline__ Tok begin(const P &p) { return p.begin(); }
  static __device__ __forceinline__ bool finish(const P &p, const Tok &t) { return p.finish(t); }
};
// LATE hooks (static constexpr bool kLate = true): nothing the hook computes is needed before the row EPILOGUE, so its loads are
// requested after the matrix stream and   bool finish(const Tok &, const double (&acc)[NS], bool owner)   runs once, between the
// first block's row sums and its epilogue calls (acc: this lane's row sum, owner: this lane runs the epilogue of a row).  The
// reductions of partials then cost no time at the front of the kernel (k_k2f: 1.8 of 7.6 us, tools/ablate.py).
template <class T, class = void> struct is_late : std::false_type {};
template <class T> struct is_late<T, std::void_t<decltype(T::kLate)>> : std::bool_constant<T::kLate> {};
template <class P, bool = is_late<P>::value> struct LateOps {
  static __device__ __forceinline__ typename PreOps<P>::Tok begin(const P &) { return typename PreOps<P>::Tok(); }
  template <int NS> static __device__ __forceinline__ bool finish(const P &, const typename PreOps<P>::Tok &, const double (&)[NS], bool) { return true; }
};
template <class P> struct LateOps<P, true> {
  static __device__ __forceinline__ typename P::Tok begin(const P &p) { return p.begin(); }
  template <int NS> static __device__ __forceinline__ bool finish(const P &p, const typename P::Tok &t, const double (&acc)[NS], bool owner) { return p.finish(t, acc, owner); }
};
//   done:           optional device flag; when set the workgroup abandons the kernel.  It is read TOGETHER with the first
//                   block descriptor (one wait for both scalar loads) instead of ahead of it.
//   Windowed blocks (L::kWin, DevCsr::blkwin):  G additionally provides
//                    using Win;  Win stage(int seg, int c) const;      element c of the input vector(s) of column segment seg
//                    void wprod(const Win &, double val, double (&prod)[NS]) const;
// The first row block's descriptors, loadable AHEAD of process_rows (the slot kernels request them together with the phase record
// they branch on, so that the record's latency is not added to the kernel's dependent-load chain).
struct FirstDesc { int4 ds, ws; };
template <bool WIN>
__device__ __forceinline__ FirstDesc first_desc(const DevCsr &M) {
  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, per = (M.nblk + 7) >> 3;
  const int b0 = xcd * per + slot0;
  FirstDesc f{make_int4(0, 0, 0, 0), make_int4(0, -1, 0, 0)};
  if (slot0 < per && b0 < M.nblk) {
    f.ds = reinterpret_cast<const int4 *>(M.blkdesc)[b0];
    if (WIN) f.ws = reinterpret_cast<const int4 *>(M.blkwin)[b0];
  }
  return f;
}
template <int NS, bool HAS_DONE, class G, class E, class Pre, class L>
__device__ __forceinline__ bool process_rows_impl(const DevCsr &M, const G &g, E &e, L &lds, Pre pre, const int *done, const FirstDesc *fd = nullptr) {
  int buf = 0;
  const int4 *desc = reinterpret_cast<const int4 *>(M.blkdesc);
  [[maybe_unused]] const int4 *wdesc = reinterpret_cast<const int4 *>(M.blkwin);
  // XCD-contiguous mapping (speed only; correctness never depends on placement): workgroup id b is observed to run on
  // XCD b % 8, so XCD x is given the contiguous row-block range [x*per, (x+1)*per).  Neighbouring row blocks gather
  // overlapping windows of the input vector; on one XCD they share those lines in one L2 instead of every XCD's L2
  // fetching (nearly) the whole vector.
  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int per = (M.nblk + 7) >> 3;
  // The done flag and the first descriptor are requested back to back, ahead of any branch that depends on either, so
  // the kernel's dependent-load chain is  {flag, descriptor} -> {col, val, rowptr} -> gather  (three levels, not five).
  const int b0 = xcd * per + slot0;
  const bool has0 = slot0 < per && b0 < M.nblk;
  int dn = 0;
  if (HAS_DONE) dn = *done;
  int4 ds = make_int4(0, 0, 0, 0), ws = make_int4(0, -1, 0, 0);
  if (fd) { ds = fd->ds; if constexpr (L::kWin) ws = fd->ws; }
  else if (has0) { ds = desc[b0]; if constexpr (L::kWin) ws = wdesc[b0]; }
  KT(Pre::kTraceBase + 1);      // flag + first descriptor arrived
  if (dn) return false;
  constexpr bool LATE = is_late<Pre>::value;
  if (!has0) {      // a workgroup without rows still runs the hook (e.g. workgroup 0 owns the PCG flags)
    if constexpr (LATE) { const double zero[NS] = {}; return LateOps<Pre>::template finish<NS>(pre, LateOps<Pre>::begin(pre), zero, false); }
    else return PreOps<Pre>::finish(pre, PreOps<Pre>::begin(pre));
  }
  // One row block.  The first one (compile-time tag) also runs the hook; it is a separate instantiation so that no
  // control-flow join sits between the hook's loads and the matrix loads (a join makes the compiler drain the counter).
  auto block = [&](auto first_tag, const int4 ds, [[maybe_unused]] const int4 ws) -> bool {
    constexpr bool FIRST = decltype(first_tag)::value;
    const int r0 = ds.x, k0 = ds.z, k1 = ds.w;
    const int r1 = ds.y < 0 ? r0 + 1 : ds.y;                    // (ds.y < 0: a long row, -(1 + index of its run table))
    const int cnt = k1 - k0;
    if (ds.y < 0) {                                             // one long row: whole workgroup reduces it
      const int *runs = M.runinfo - (1 + ds.y);
      typename PreOps<Pre>::Tok ltok = typename PreOps<Pre>::Tok();
      if constexpr (FIRST && !LATE) { if (!PreOps<Pre>::finish(pre, PreOps<Pre>::begin(pre))) return false; }
      if (threadIdx.x == 0) e.prefetch(r0);                     // epilogue operands requested before the stream, not after it
      if constexpr (FIRST && LATE) ltok = LateOps<Pre>::begin(pre);
      double acc[NS];
#pragma unroll
      for (int s = 0; s < NS; s++) acc[s] = 0.0;
      // kChunk entries per step, as in the short-row path: 8 index loads, 8 value loads, 8 gathers per lane in flight
      // together (one load batch per lane and step reached 2.9 TB/s on dense 5000-entry rows; this form is the fix)
      int c0 = runs[0];                                         // (requested one slice ahead: it gates the slice's first loads)
      for (int base = k0, j = 0; base < k1; base += kChunk, j++) {
        int cc[kChunk / kBlock];
        double vv[kChunk / kBlock];
        const int crun = c0;
        if (base + kChunk < k1) c0 = runs[j + 1];
        if (crun >= 0) {                                        // consecutive columns (dense block): no index loads for this slice
#pragma unroll
          for (int u = 0; u < kChunk / kBlock; u++) { const int k = base + (int)threadIdx.x + u * kBlock; cc[u] = k < k1 ? crun + (k - base) : -1; }
        } else {
#pragma unroll
          for (int u = 0; u < kChunk / kBlock; u++) { const int k = base + (int)threadIdx.x + u * kBlock; cc[u] = k < k1 ? M.col[k] : -1; }
        }
#pragma unroll
        for (int u = 0; u < kChunk / kBlock; u++) { const int k = base + (int)threadIdx.x + u * kBlock; vv[u] = k < k1 ? M.val[k] : 0.0; }
        typename GatherOps<G>::Ops ops[kChunk / kBlock];
#pragma unroll
        for (int u = 0; u < kChunk / kBlock; u++) if (cc[u] >= 0) ops[u] = GatherOps<G>::fetch(g, cc[u]);
#pragma unroll
        for (int u = 0; u < kChunk / kBlock; u++) {
          if (cc[u] < 0) continue;
          double pr[NS];
          GatherOps<G>::template prod<NS>(g, ops[u], cc[u], vv[u], pr);
#pragma unroll
          for (int s = 0; s < NS; s++) acc[s] += pr[s];
        }
      }
#pragma unroll
      for (int s = 0; s < NS; s++) acc[s] = block_sum(acc[s], lds.red);
      if constexpr (FIRST && LATE) { if (!LateOps<Pre>::template finish<NS>(pre, ltok, acc, threadIdx.x == 0)) return false; }
      if (threadIdx.x == 0) e(r0, acc);
      return true;
    }
    // many short rows: stage products in LDS
    typename PreOps<Pre>::Tok tok = typename PreOps<Pre>::Tok();
    if constexpr (FIRST && !LATE) tok = PreOps<Pre>::begin(pre);   // the hook's loads go out first: they are needed first
    const int lpr = 2 * (r1 - r0) <= kBlock ? 2 : 1;
    const int sub = lpr == 2 ? (int)(threadIdx.x & 1) : 0;
    const int myr = r0 + (lpr == 2 ? (int)(threadIdx.x >> 1) : (int)threadIdx.x);   // the row this lane reduces in the first pass
    int rp0 = 0, rp1 = 0;                                        // raw row pointers: not touched before the barrier
    bool staged = false;
    if constexpr (L::kWin) {
      if (ws.y >= 0) {
        // Windowed block: the window of the input vector(s) is fetched with coalesced loads (requested FIRST: it is needed
        // first), written to LDS, and the per-entry gathers are LDS reads through 16-bit local indices.
        // Instruction count matters as much as bytes here (the load-issue phase was ~700 instructions per wave with one exec-mask
        // branch per load: 1.7 us of the kernel's 6.7): trip counts are BLOCK-UNIFORM (scalar branches), lanes past the end of the
        // last partial batch re-read the last element (same cache line as their neighbours) instead of branching around the load.
        using T = typename G::Win;
        constexpr int CW = (kWinCap + kBlock - 1) / kBlock;
        constexpr int CE = kChunk / kBlock;
        const int wl = ws.y + ws.w;
        const int nw = (wl + kBlock - 1) / kBlock, nu = (cnt + kBlock - 1) / kBlock;      // (cnt >= 1, wl >= 1 for a windowed block)
        T we[CW];
#pragma unroll
        for (int u = 0; u < CW; u++) {
          if (u < nw) {
            const int p = min((int)threadIdx.x + u * kBlock, wl - 1);
            we[u] = p < ws.y ? g.stage(0, ws.x + p) : g.stage(1, ws.z + (p - ws.y));
          }
        }
        int lc[CE];
        double vw[CE];
        const unsigned short *lcp = M.lcol + k0;
        const double *vp = M.val + k0;
#pragma unroll
        for (int u = 0; u < CE; u++) { if (u < nu) lc[u] = (int)lcp[min((int)threadIdx.x + u * kBlock, cnt - 1)]; }
#pragma unroll
        for (int u = 0; u < CE; u++) { if (u < nu) vw[u] = vp[min((int)threadIdx.x + u * kBlock, cnt - 1)]; }
        if (myr < r1) { rp0 = M.rowptr[myr]; rp1 = M.rowptr[myr + 1]; if (sub == 0) e.prefetch(myr); }
        if constexpr (FIRST && LATE) tok = LateOps<Pre>::begin(pre);            // a late hook's loads go out last
        KT(Pre::kTraceBase + 2);
#pragma unroll
        for (int u = 0; u < CW; u++) { if (u < nw) lds.win[min((int)threadIdx.x + u * kBlock, wl - 1)] = we[u]; }     // (clamped lanes store the same value)
        if constexpr (FIRST && !LATE) { if (!PreOps<Pre>::finish(pre, tok)) return false; KT(Pre::kTraceBase + 3); }
        __syncthreads();
        T gv[CE];
#pragma unroll
        for (int u = 0; u < CE; u++) { if (u < nu) gv[u] = lds.win[lc[u]]; }            // all LDS gathers in flight together
#pragma unroll
        for (int u = 0; u < CE; u++) {
          if (u < nu) {
            double pr[NS];
            g.wprod(gv[u], vw[u], pr);         // (slots past cnt hold a copy of the last product: never read by the row sums)
#pragma unroll
            for (int s = 0; s < NS; s++) lds.prod[buf][s][threadIdx.x + u * kBlock] = pr[s];
          }
        }
        staged = true;
      }
    }
    if (!staged) {
    int cc[kChunk / kBlock];
    double vv[kChunk / kBlock];
    // (masked, not clamped: a lane past the block's last entry issues nothing.  Re-reading the last entry instead makes the
    // code branch-free but was measured 6 % slower -- row blocks are ~2/3 full, and the extra requests cost more than the branches)
#pragma unroll
    for (int u = 0; u < kChunk / kBlock; u++) { const int k = threadIdx.x + u * kBlock; cc[u] = k < cnt ? M.col[k0 + k] : -1; }     // indices first:
#pragma unroll
    for (int u = 0; u < kChunk / kBlock; u++) { const int k = threadIdx.x + u * kBlock; vv[u] = k < cnt ? M.val[k0 + k] : 0.0; }    // the gathers wait only for them
    // Row sums: one lane per row, or -- when the block has at most kBlock/2 rows -- TWO lanes per row (even/odd entries,
    // combined with one shuffle): halves the serial chain of LDS reads of the row-sum phase for matrices with ~100 rows of
    // ~14 entries per block (B at config 2: k_k2f 7.9 -> 7.5 us back to back).
    if (myr < r1) { rp0 = M.rowptr[myr]; rp1 = M.rowptr[myr + 1]; if (sub == 0) e.prefetch(myr); }
    KT(Pre::kTraceBase + 2);
    typename GatherOps<G>::Ops ops[kChunk / kBlock];
#pragma unroll
    for (int u = 0; u < kChunk / kBlock; u++) if (cc[u] >= 0) ops[u] = GatherOps<G>::fetch(g, cc[u]);   // gathers requested as the indices arrive
    if constexpr (FIRST && LATE) tok = LateOps<Pre>::begin(pre);
    if constexpr (FIRST && !LATE) { if (!PreOps<Pre>::finish(pre, tok)) return false; KT(Pre::kTraceBase + 3); }
#pragma unroll
    for (int u = 0; u < kChunk / kBlock; u++) {
      if (cc[u] < 0) continue;
      double pr[NS];
      GatherOps<G>::template prod<NS>(g, ops[u], cc[u], vv[u], pr);
#pragma unroll
      for (int s = 0; s < NS; s++) lds.prod[buf][s][threadIdx.x + u * kBlock] = pr[s];
    }
    }  // !staged
    __syncthreads();
    KT(Pre::kTraceBase + 4);    // products staged
    {
      double acc[NS];
#pragma unroll
      for (int s = 0; s < NS; s++) acc[s] = 0.0;
      const bool owner = myr < r1 && sub == 0;
      if (KNOCKED(16)) { if (owner) for (int s = 0; s < NS; s++) acc[s] = lds.prod[buf][s][threadIdx.x]; }
      else {
        // The first kRowBatch entries of the lane's share are read with INDEPENDENT LDS loads (one latency, not one per entry:
        // the variable-trip-count loop below serialised a ds_read + wait per entry, 0.75 us for the 12-entry rows of B at config
        // 2) and added in entry order -- the same sum as the plain loop, since the masked slots add 0.0.
        constexpr int kRowBatch = 8;
        if (myr < r1) {
          const int ra = rp0 - k0, rz = rp1 - k0;
          double v[NS][kRowBatch];
#pragma unroll
          for (int b = 0; b < kRowBatch; b++) {
            const int k = ra + sub + lpr * b;
#pragma unroll
            for (int s = 0; s < NS; s++) v[s][b] = k < rz ? lds.prod[buf][s][k] : 0.0;
          }
#pragma unroll
          for (int b = 0; b < kRowBatch; b++) {
#pragma unroll
            for (int s = 0; s < NS; s++) acc[s] += v[s][b];
          }
          for (int k = ra + sub + lpr * kRowBatch; k < rz; k += lpr) {
#pragma unroll
            for (int s = 0; s < NS; s++) acc[s] += lds.prod[buf][s][k];
          }
        }
        if (lpr == 2) {
#pragma unroll
          for (int s = 0; s < NS; s++) acc[s] += dpp<0xb1>(acc[s]);            // lane ^ 1 (whole waves take this path: lpr is block-uniform)
        }
      }
      if constexpr (FIRST && LATE) { KT(Pre::kTraceBase + 7); if (!LateOps<Pre>::template finish<NS>(pre, tok, acc, owner)) return false; KT(Pre::kTraceBase + 3); }
      if (owner) e(myr, acc);
    }
    for (int r = myr + kBlock; lpr == 1 && r < r1; r += kBlock) {   // blocks with more than kBlock (mostly empty) rows
      const int a = M.rowptr[r] - k0, z = M.rowptr[r + 1] - k0;
      double acc[NS];
#pragma unroll
      for (int s = 0; s < NS; s++) acc[s] = 0.0;
      for (int k = a; k < z; k++) {
#pragma unroll
        for (int s = 0; s < NS; s++) acc[s] += lds.prod[buf][s][k];
      }
      e.prefetch(r); e(r, acc);
    }
    KT(Pre::kTraceBase + 5);    // row sums + epilogue done
    if (L::kBuf == 2) buf ^= 1;               // the next row block fills the other buffer: one barrier per block suffices
    else __syncthreads();                      // single buffer (two-sum and windowed kernels): protect it before the next fill
    return true;
  };
  if (!block(std::true_type(), ds, ws)) return false;
  for (int sl = slot0 + slots; sl < per; sl += slots) {
    const int b = xcd * per + sl;
    if (b >= M.nblk) break;
    int4 wn = make_int4(0, -1, 0, 0);
    if constexpr (L::kWin) wn = wdesc[b];
    block(std::false_type(), desc[b], wn);
  }
  return true;
}
template <int NS, class G, class E, class Pre, class L>
__device__ __forceinline__ bool process_rows(const DevCsr &M, const G &g, E &e, L &lds, Pre pre) { return process_rows_impl<NS, false>(M, g, e, lds, pre, nullptr); }
template <int NS, class G, class E, class Pre, class L>
__device__ __forceinline__ bool process_rows(const DevCsr &M, const G &g, E &e, L &lds, Pre pre, const int *done) { return process_rows_impl<NS, true>(M, g, e, lds, pre, done); }
template <int NS, class G, class E, class L>
__device__ __forceinline__ void process_rows(const DevCsr &M, const G &g, E &e, L &lds) { process_rows_impl<NS, false>(M, g, e, lds, NoPre(), nullptr); }
template <int NS, class G, class E, class Pre, class L>
__device__ __forceinline__ bool process_rows_fd(const DevCsr &M, const G &g, E &e, L &lds, Pre pre, const FirstDesc &fd) { return process_rows_impl<NS, false>(M, g, e, lds, pre, nullptr, &fd); }
struct NoPrefetch { __device__ __forceinline__ void prefetch(int) {} };


// ---- shared by several units
struct GVec {
  const double *x;
  __device__ __forceinline__ void operator()(int c, double a, double (&pr)[1]) const { pr[0] = a * x[c]; }
  using Win = double;                                     // windowed row blocks: x's window staged in LDS
  __device__ __forceinline__ Win stage(int, int c) const { return x[c]; }
  __device__ __forceinline__ void wprod(const Win &o, double a, double (&pr)[1]) const { pr[0] = a * o; }
};
struct EStore : NoPrefetch { double *out; __device__ __forceinline__ void operator()(int r, const double (&s)[1]) { out[r] = s[0]; } };
struct GVecSplit {           // one concatenated input vector; windowed blocks address its two column segments separately
  const double *x; int split;
  __device__ __forceinline__ void operator()(int c, double a, double (&pr)[1]) const { pr[0] = a * x[c]; }
  using Win = double;
  __device__ __forceinline__ Win stage(int seg, int c) const { return x[seg ? split + c : c]; }
  __device__ __forceinline__ void wprod(const Win &o, double a, double (&pr)[1]) const { pr[0] = a * o; }
};
[[maybe_unused]] __global__ __launch_bounds__(kBlock) void k_test_spmv(DevCsr M, const double *in, double *out) {     // the path the hot kernels take (windowed where the block is)
  __shared__ StreamLdsW<1, double> lds;
  GVecSplit g{in, M.split};
  EStore e{{}, out};
  process_rows<1>(M, g, e, lds);
}
struct EPrec : NoPrefetch { const double *Bval; const int *Bdiag; double *Minv; __device__ __forceinline__ void operator()(int j, const double (&s)[1]) { Minv[j] = 1.0 / (Bval[Bdiag[j]] + s[0]); } };

// slot records (pcg_hip.hip "slot kernels"; the boundary kernels of backend_hip.hip start chunks through them)
enum SlotPhase { P_KB = 0, P_K1, P_K2F, P_K1F, P_KA, P_IDLE, P_F /* a PCG iteration of the F1 form (k_slot1) */ };
enum SlotRec { SR_PHASE = 0, SR_K, SR_ADMM, SR_TARGET, SR_USED, SR_CONV, SR_CAP, SR_SEQ /* slots executed since k_slot_init: every slot adds one */, SR_WORDS = 8 };

struct SlotState { int ph, k, admm, target, used, conv, cap, seq; };
__device__ __forceinline__ SlotState slot_read(const int *r) { return SlotState{r[SR_PHASE], r[SR_K], r[SR_ADMM], r[SR_TARGET], r[SR_USED], r[SR_CONV], r[SR_CAP], r[SR_SEQ]}; }
// The same record through the CONSTANT address space: two scalar loads (lgkmcnt) instead of eight vector loads (vmcnt).  The record was written by
// the PREVIOUS launch (the scalar cache is invalidated at every kernel start) and this launch writes the OTHER record, so the value cannot change
// under the kernel.  Used by the F1 slot kernel, whose head keeps LDS-direct transfers in flight: a wait on a vector load there would wait for
// those transfers as well (the memory counter retires in issue order).
__device__ __forceinline__ SlotState slot_read_scalar(const int *r) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef int __attribute__((ext_vector_type(4))) v4i;
  typedef const v4i __attribute__((address_space(4))) *cptr;
  const v4i a = ((cptr)(unsigned long long)r)[0], b = ((cptr)(unsigned long long)r)[1];
  return SlotState{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#else
  return slot_read(r);
#endif
}
__device__ __forceinline__ void slot_write(int *w, const SlotState &s) {
  if (blockIdx.x == 0 && threadIdx.x == 0) { w[SR_PHASE] = s.ph; w[SR_K] = s.k; w[SR_ADMM] = s.admm; w[SR_TARGET] = s.target; w[SR_USED] = s.used; w[SR_CONV] = s.conv; w[SR_CAP] = s.cap; w[SR_SEQ] = s.seq + 1; }
}

#define LAUNCH(kernel, d, ...) hipLaunchKernelGGL(kernel, dim3(kGrid), dim3(kBlock), 0, st(d), __VA_ARGS__)

}  // namespace

// ---- functions one unit defines and another calls
void wb_factor(Dev &d);                    // woodbury_hip.hip: D0, S, S^-1 for the current rho (called by precond)
void wb_factor_device(Dev &d, int cond);   // woodbury_hip.hip: the small form's re-factorisation as launches only (cond: inside a boundary group)
void dev_publish(Dev &d);                  // pcg_hip.hip: bring the device copy of Dev the F1 slot kernel reads up to date (never inside a stream capture)
void dev_release(Dev &d);                  // pcg_hip.hip: free it (called by destroy)
// dense_hip.hip: the dense fp64 building blocks of the device-factorised Woodbury correction on the matrix cores (strided operands; SPD inverse in place)
void dense_gemm(void *stream, int M, int N, int K, double alpha, const double *A, long as_i, long as_k, const double *B, long bs_k, long bs_j, double beta, double *C, long cs_i, long cs_j);
void dense_spd_inverse(void *stream, double *A, long ld, int n, double *work, double *minpiv);
void wb_release_blas(void *handle);        // woodbury_hip.hip: destroy the rocBLAS handle a Dev's Impl holds (called by destroy)

}  // namespace be
}  // namespace osqp_hip
