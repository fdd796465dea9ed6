// pcg_hip.hip -- the ADMM / PCG hot path of backend.h on gfx950: KB, K1, K2, Kv, the fused pair K2F / K1F, KA, the slot kernels (device-side
// scheduling of the phases), the one-launch PCG iteration (F1 form, k_slot1) and the timing probes.  Design notes: backend_hip.hip (head of the
// file) and DESIGN.md section 4.  Split out of backend_hip.hip in round 4; the shared device helpers are hip_common.h.
#include "hip_common.h"

namespace osqp_hip {
namespace be {

namespace {

// ---------------------------------------------------------------------------------------------- hot-path kernels
struct GKb {
  const double *xg, *v, *t0; int n;      // (xg: the PCG start, Dev::xg; t0 = rho .* (A xg))
  __device__ __forceinline__ void operator()(int c, double a, double (&pr)[2]) const {
    if (c < n) { pr[0] = 0.0; pr[1] = a * xg[c]; }
    else { pr[0] = a * v[c - n]; pr[1] = a * t0[c - n]; }
  }
};
struct EKb {
  const double *x, *q, *Minv; double *r, *uu; double sigma; const double *xg; double *xs; double g = 0, rn = 0, bn = 0; double px = 0, pq = 0, pm = 0, pg = 0; int wt = 0;
  __device__ __forceinline__ void prefetch(int j) { px = x[j]; pq = q[j]; pm = Minv[j]; pg = xg[j]; }
  __device__ __forceinline__ void operator()(int j, const double (&s)[2]) {
    const double rhs = sigma * px - pq + s[0];
    const double rr = rhs - s[1], u = pm * rr;
    stw(wt, r + j, rr); stw(wt, uu + j, u); stw(wt, xs + j, pg);           // x~ restarts from the extrapolated point (nobody gathers xs in this kernel)
    g += rr * u; rn = nanmax(rn, fabs(rr)); bn = nanmax(bn, fabs(rhs));
  }
};
__global__ __launch_bounds__(kBlock) void k_kb(Dev d) {
  __shared__ StreamLds<2> lds;
  GKb g{d.xg, d.v, d.t0, d.n};
  EKb e{d.x, d.q, d.Minv, d.r, d.uu, d.sigma, d.xg, d.xs}; e.wt = d.wt;
  process_rows<2>(d.B, g, e, lds);
  __syncthreads();
  const double G = block_sum(e.g, lds.red);
  double RN = e.rn, BN = e.bn;
  block_max2(RN, BN, lds.red);
  put_partial(d.part, SL_GAMMA0, G); put_partial(d.part, SL_RN0, RN); put_partial(d.part, SL_BN, BN);
  if (blockIdx.x == 0 && threadIdx.x == 0) { d.flags[F_DONE] = 0; d.flags[F_ITERS] = 0; }
}

// K1 ------------------------------------------------------------------------------------------
struct EK1 {
  const double *rho; double *t; double pr = 0;
  __device__ __forceinline__ void prefetch(int i) { pr = rho[i]; }
  __device__ __forceinline__ void operator()(int i, const double (&s)[1]) { t[i] = pr * s[0]; }
};
// PCG stopping test, run by every working workgroup (and workgroup 0) while its first matrix loads are in flight
struct PreK1 {
  [[maybe_unused]] static constexpr int kTraceBase = 0;
  const Dev &d; int i, probe; double *red; int par = -1;      // par: parity of the ADMM iteration (slot form), -1: not recorded
  __device__ __forceinline__ bool operator()() const {
    if (probe == 1) return true;
    // (No early exit on d.flags[F_DONE] here: in the slot form workgroup 0 of THIS launch may set the flag while other workgroups are
    // still arriving, the waves of one workgroup then read different values, one skips the barriers of the reduction below and
    // meets its siblings at the next __syncthreads() instead -- they fold its stale LDS slot.  Seen as run-to-run differences of
    // solves running concurrently on several streams, 1-4 % of them (tools/thread_stress.py); every wave takes the reduction now.)
    const PartRegs prn = partial_load(d.part + (SL_RN0 + (i & 1)) * kGrid), pbn = partial_load(d.part + SL_BN * kGrid);
    double rn = partial_fold_max(prn), bn = partial_fold_max(pbn);
    block_max2(rn, bn, red);
    if (probe) { if (rn < -1.0) d.res[R_COUNT - 1] = bn; return true; }     // probe == 2: pay for the test, ignore it
    const double tol = fmax(d.scal[S_TOL_REL] * bn, d.scal[S_TOL_ABS]);
    if (i == 0 && blockIdx.x == 0 && threadIdx.x == 0) { d.scal[S_TOL_NOW] = tol; d.scal[S_RN0] = rn; if (par >= 0) d.scal[S_RN0H + par] = rn; }   // fused PCG: later tests read the scalar
    if (!(rn > tol)) {            // converged (a NaN residual also stops the inner loop; the ADMM residuals will flag it)
      if (blockIdx.x == 0 && threadIdx.x == 0) { d.flags[F_DONE] = 1; d.flags[F_ITERS] = i; }
      return false;
    }
    return true;
  }
};
__global__ __launch_bounds__(kBlock) void k_k1(Dev d, int i, int probe) {
  __shared__ StreamLdsW<1, double> lds;
  if (!wg_has_rows(d.A) && blockIdx.x != 0) return;               // nothing to do and not the flag owner
  if (!probe && d.flags[F_DONE]) return;                          // PCG already converged: cheapest possible exit
  GVec g{d.uu};
  EK1 e{d.rho, d.t};
  process_rows<1>(d.A, g, e, lds, PreK1{d, i, probe, lds.red});
}

// K2 ------------------------------------------------------------------------------------------
struct GSplit {      // [pn; pm] indexed by a B column
  const double *pn, *pm; int n;
  __device__ __forceinline__ void operator()(int c, double a, double (&pr)[1]) const { pr[0] = a * (c < n ? pn[c] : pm[c - n]); }
};
struct EK2 {
  const double *uu; double *w; double dl = 0, pu = 0;
  __device__ __forceinline__ void prefetch(int j) { pu = uu[j]; }
  __device__ __forceinline__ void operator()(int j, const double (&s)[1]) { w[j] = s[0]; dl += s[0] * pu; }
};
__global__ __launch_bounds__(kBlock) void k_k2(Dev d, int probe) {
  __shared__ StreamLds<1> lds;
  if (!probe && d.flags[F_DONE]) return;
  GSplit g{d.uu, d.t, d.n};
  EK2 e{d.uu, d.w};
  if (!process_rows<1>(d.B, g, e, lds, NoPre())) return;
  __syncthreads();
  const double DL = block_sum(e.dl, lds.red);
  put_partial(d.part, SL_DELTA, DL);
  KT(6);
}

// Kv ------------------------------------------------------------------------------------------
// VEC = 2: one double2 per lane (large n); VEC = 1: one double per lane (keeps more workgroups busy at mid-size n)
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_kv(Dev d, int i, int probe) {
  __shared__ double sred[2 * kWaves];
  const int nv = d.n / VEC;                                       // vector elements (tail handled by workgroup 0)
  // XCD-contiguous chunks of kBlock elements, as in process_rows (each XCD keeps 'its' eighth of the PCG vectors)
  const int nchunk = (nv + kBlock - 1) / kBlock, per = (nchunk + 7) >> 3, slots = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
  const int c0 = xcd * per + slot0;
  const bool active = slot0 < per && c0 < nchunk;
  const int j0 = c0 * kBlock + threadIdx.x;
  if (!active && blockIdx.x != 0) {                                // idle workgroup: its partial slots must still read 0
    if (!probe) { put_partial(d.part, SL_GAMMA0 + ((i + 1) & 1), 0.0); put_partial(d.part, SL_RN0 + ((i + 1) & 1), 0.0); }
    return;
  }
  if (!probe && d.flags[F_DONE]) return;                           // PCG already converged
  const bool first = (i == 0) && !probe;
  typedef typename std::conditional<VEC == 2, double2, double>::type V;
  // fused PCG (final update after the last budgeted iteration): s_i is already complete (k_k2 epilogue), w is not stored,
  // u_i lives in the ping-pong buffer of parity i
  const bool fz = d.fused && !probe;
  const V *uin = reinterpret_cast<const V *>(fz && (i & 1) ? d.uu2 : d.uu);
  V *uout = reinterpret_cast<V *>(fz && !(i & 1) ? d.uu2 : d.uu);
  V *p2 = reinterpret_cast<V *>(d.p), *x2 = reinterpret_cast<V *>(d.xs), *r2 = reinterpret_cast<V *>(d.r), *s2 = reinterpret_cast<V *>(d.s);
  const V *w2 = reinterpret_cast<const V *>(d.w), *m2 = reinterpret_cast<const V *>(d.Minv);
  // issue this lane's first element loads, then fold the partials while they are in flight
  const bool have = active && j0 < nv;
  V u, w, x, r, mi, p, s;
  if (have) { u = uin[j0]; x = x2[j0]; r = r2[j0]; mi = m2[j0]; if (fz) { s = s2[j0]; if (!first) p = p2[j0]; } else { w = w2[j0]; if (!first) { p = p2[j0]; s = s2[j0]; } } }
  double *gam = d.scal + S_HIST, *alp = d.scal + S_HIST + kMaxCg + 1;
  double alpha = 0.0, beta = 0.0;
  if (probe != 1) {
    const PartRegs pg = partial_load(d.part + (SL_GAMMA0 + (i & 1)) * kGrid), pd = partial_load(d.part + SL_DELTA * kGrid);
    double gamma = partial_fold_sum(pg), delta = partial_fold_sum(pd);
    block_sum2(gamma, delta, sred);
    if (probe) { if (gamma == -1.2345e300) d.res[R_COUNT - 1] = delta; }   // probe == 2: pay for the reduction, ignore it
    else {
      if (i == 0) { beta = 0.0; alpha = gamma / delta; }
      else { beta = gamma / gam[i - 1]; alpha = gamma / (delta - beta * gamma / alp[i - 1]); }
      if (blockIdx.x == 0 && threadIdx.x == 0) { gam[i] = gamma; alp[i] = alpha; }
    }
  }
  double g = 0, rn = 0;
  auto upd = [&](double &uu_, double ww_, double &xx_, double &rr_, double mm_, double &pp_, double &ss_) {
    if (fz) { pp_ = first ? uu_ : uu_ + beta * pp_; }                 // ss_ is s_i already
    else if (first) { pp_ = uu_; ss_ = ww_; } else { pp_ = uu_ + beta * pp_; ss_ = ww_ + beta * ss_; }
    xx_ += alpha * pp_; rr_ -= alpha * ss_; uu_ = mm_ * rr_;
    g += rr_ * uu_; rn = nanmax(rn, fabs(rr_));
  };
  for (int sl = slot0; active && sl < per; sl += slots) {
    const int c = xcd * per + sl;
    if (c >= nchunk) break;
    const int j = c * kBlock + threadIdx.x;
    if (j >= nv) break;
    if (sl != slot0) { u = uin[j]; x = x2[j]; r = r2[j]; mi = m2[j]; if (fz) { s = s2[j]; if (!first) p = p2[j]; } else { w = w2[j]; if (!first) { p = p2[j]; s = s2[j]; } } }
    if constexpr (VEC == 2) { upd(u.x, w.x, x.x, r.x, mi.x, p.x, s.x); upd(u.y, w.y, x.y, r.y, mi.y, p.y, s.y); }
    else upd(u, w, x, r, mi, p, s);
    p2[j] = p; if (!fz) s2[j] = s; x2[j] = x; r2[j] = r; uout[j] = u;
  }
  if (VEC == 2 && (d.n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {      // odd tail element
    const int j = d.n - 1;
    const double *uin1 = reinterpret_cast<const double *>(uin); double *uout1 = reinterpret_cast<double *>(uout);
    double uu_ = uin1[j], xx_ = d.xs[j], rr_ = d.r[j], pp_ = first ? 0.0 : d.p[j], ss_ = (first && !fz) ? 0.0 : d.s[j];
    upd(uu_, fz ? 0.0 : d.w[j], xx_, rr_, d.Minv[j], pp_, ss_);
    uout1[j] = uu_; d.xs[j] = xx_; d.r[j] = rr_; d.p[j] = pp_; if (!fz) d.s[j] = ss_;
  }
  block_sum_max(g, rn, sred);
  if (!probe) { put_partial(d.part, SL_GAMMA0 + ((i + 1) & 1), g); put_partial(d.part, SL_RN0 + ((i + 1) & 1), rn); }
}


// Fused PCG (two kernels per iteration) --------------------------------------------------------------------------------
// K2F_k :  stopping test on ||r_k||; beta_k = gamma_k / gamma_{k-1};  w = B [u_k; t_k];  delta_k = <w, u_k>;  and in the row
//          epilogue  s_k = w + beta_k s_{k-1},  ms_k = Minv .* s_k   (w itself is never stored)
// K1F_{k+1}: alpha_k = gamma_k / (delta_k - beta_k gamma_k / alpha_{k-1});
//          (a) on this workgroup's chunk of the n-vectors:  p = u_k + beta_k p ; xs += alpha_k p ; r -= alpha_k s_k ;
//              u_{k+1} = Minv r  (written to the OTHER u buffer) ; partials gamma_{k+1}, ||r_{k+1}||_inf
//          (b) t_{k+1} = rho .* (A u_{k+1})  with u_{k+1}[c] = u_k[c] - alpha_k ms_k[c] recomputed at every gathered column
//              (two gathers), so (b) never waits for (a) of another workgroup.
struct GSplitU {
  const double *pn, *pm; int n;
  using Ops = double;
  __device__ __forceinline__ Ops fetch(int c) const { return c < n ? pn[c] : pm[c - n]; }
  __device__ __forceinline__ void prod(const Ops &o, double a, double (&pr)[1]) const { pr[0] = a * o; }
  __device__ __forceinline__ void operator()(int c, double a, double (&pr)[1]) const { pr[0] = a * fetch(c); }
  using Win = double;
  __device__ __forceinline__ Win stage(int seg, int c) const { return seg ? pm[c] : pn[c]; }
  __device__ __forceinline__ void wprod(const Win &o, double a, double (&pr)[1]) const { pr[0] = a * o; }
};
struct EK2F {
  const double *u, *Minv; double *s, *ms; double beta = 0; int first = 0; double dl = 0, pu = 0, pm = 0, ps = 0; int wt = 0;
  __device__ __forceinline__ void prefetch(int j) { pu = u[j]; pm = Minv[j]; ps = s[j]; }
  __device__ __forceinline__ void operator()(int j, const double (&sm)[1]) {
    const double w = sm[0], sn = first ? w : w + beta * ps;
    dl += w * pu;
    if (KNOCKED(32)) { if (sn == -1.2345e300) s[j] = sn; return; }
    stw(wt, s + j, sn);
    stw(wt, ms + j, pm * sn);                                                       // Minv .* s_k: the vector k_k1f applies A to
  }
};
// block reduction of three quantities (sum, max, sum) behind ONE barrier pair; sred needs 3 * kWaves doubles
__device__ __forceinline__ void block_sum_max_sum(double &a, double &b, double &c, double *sred) {
  a = wave_sum(a); b = wave_max(b); c = wave_sum(c);
  if ((threadIdx.x & 63) == kReduceLane) { sred[threadIdx.x >> 6] = a; sred[kWaves + (threadIdx.x >> 6)] = b; sred[2 * kWaves + (threadIdx.x >> 6)] = c; }
  __syncthreads();
  a = sred_sum(sred); b = sred_max(sred + kWaves); c = sred_sum(sred + 2 * kWaves);
  __syncthreads();
}
// LATE hook of k_k2f (runs between the row sums and the epilogue): folds gamma_k and ||r_k||_inf together with this
// workgroup's share of delta_k = <w, u_k> (one barrier pair for all three), stopping test, beta_k.
struct PreK2F {
  [[maybe_unused]] static constexpr int kTraceBase = 0;
  static constexpr bool kLate = true;
  const Dev &d; int k; EK2F *e; double *red; double *dl_first;
  struct Tok { PartRegs prn, pg; double tol, glast; };
  __device__ __forceinline__ Tok begin() const {
    Tok t;
    if (KNOCKED(1)) { t.tol = 0; t.glast = 1; return t; }
    if (KNOCKED(256)) { for (int q = 0; q < kPart; q++) { t.prn.v[q] = 1.0; t.pg.v[q] = 1.0; } }
    else { t.prn = partial_load(d.part + (SL_RN0 + (k & 1)) * kGrid); t.pg = partial_load(d.part + (SL_GAMMA0 + (k & 1)) * kGrid); }
    t.tol = d.scal[S_TOL_NOW]; t.glast = k == 0 ? 1.0 : d.scal[S_HIST + k - 1];
    return t;
  }
  __device__ __forceinline__ bool finish(const Tok &t, const double (&acc)[1], bool owner) const {
    if (KNOCKED(1)) { e->beta = 0.5; e->first = 0; *dl_first = 0; return true; }
    double gamma = partial_fold_sum(t.pg), rn = partial_fold_max(t.prn);
    const double dl0 = owner ? acc[0] * e->pu : 0.0;
    double dls = dl0;
    block_sum_max_sum(gamma, rn, dls, red);
    double *gam = d.scal + S_HIST, *bet = d.scal + S_HIST + 2 * (kMaxCg + 1);
    if (k > 0 && !(rn > t.tol)) {        // converged after k iterations (k == 0 was tested by k_k1)
      if (blockIdx.x == 0 && threadIdx.x == 0) { d.flags[F_DONE] = 1; d.flags[F_ITERS] = k; }
      return false;
    }
    const double beta = k == 0 ? 0.0 : gamma / t.glast;
    if (blockIdx.x == 0 && threadIdx.x == 0) { gam[k] = gamma; bet[k] = beta; }
    e->beta = beta; e->first = (k == 0);
    e->dl = -dl0;                        // the epilogue adds this row's term again: e->dl then holds only LATER rows' terms
    *dl_first = dls;
    return true;
  }
};
__global__ __launch_bounds__(kBlock) void k_k2f(Dev d, int k) {
  __shared__ StreamLdsW<1, double> lds;
  KT(0);
  const double *u = (k & 1) ? d.uu2 : d.uu;
  GSplitU g{u, d.t, d.n};
  EK2F e{u, d.Minv, d.s, d.ms}; e.wt = d.wt;
  double dl_first = 0.0;
  if (!process_rows<1>(d.B, g, e, lds, PreK2F{d, k, &e, lds.red, &dl_first}, d.flags + F_DONE)) return;
  if (KNOCKED(64)) { if (e.dl == -1.2345e300) put_partial(d.part, SL_DELTA, e.dl); return; }
  double DL = dl_first;
  if (!d.B.single) { __syncthreads(); DL += block_sum(e.dl, lds.red); }    // rows beyond the first pass of the first block
  put_partial(d.part, SL_DELTA, DL);
  KT(6);
}
// K1F_{k+1}: S = A (Minv .* s_k) needs no scalar; alpha_k (from the delta partials) enters only the row epilogue
//   t_{k+1} = t_k - alpha_k rho .* S      ( = rho .* A u_{k+1},  u_{k+1} = u_k - alpha_k Minv .* s_k )
// and this workgroup's slice of the vector update, so the reduction of partials runs LATE, behind the matrix stream.
struct GMs {
  const double *ms;
  using Ops = double;
  __device__ __forceinline__ Ops fetch(int c) const { return ms[c]; }
  __device__ __forceinline__ void prod(const Ops &o, double a, double (&pr)[1]) const { pr[0] = a * o; }
  __device__ __forceinline__ void operator()(int c, double a, double (&pr)[1]) const { pr[0] = a * ms[c]; }
  using Win = double;
  __device__ __forceinline__ Win stage(int, int c) const { return ms[c]; }
  __device__ __forceinline__ void wprod(const Win &o, double a, double (&pr)[1]) const { pr[0] = a * o; }
};
struct EK1F {
  const double *rho; double *t; double alpha = 0, pr = 0, pt = 0; int wt = 0;
  __device__ __forceinline__ void prefetch(int i) { pr = rho[i]; pt = t[i]; }
  __device__ __forceinline__ void operator()(int i, const double (&s)[1]) { stw(wt, t + i, pt - alpha * pr * s[0]); }
};
struct PreK1F {
  [[maybe_unused]] static constexpr int kTraceBase = 8;
  static constexpr bool kLate = true;
  const Dev &d; int k; bool has_vec; EK1F *e; double *red; double *g, *rn;
  struct Tok { PartRegs pd; double gamma, beta, alast; double u0, p0, r0, s0, m0, x0; };
  __device__ __forceinline__ int first_index() const {
    const int nchunk = (d.n + kBlock - 1) / kBlock, per = (nchunk + 7) >> 3;
    return (int)(((blockIdx.x & 7) * per + (blockIdx.x >> 3)) * kBlock + threadIdx.x);
  }
  __device__ __forceinline__ Tok begin() const {
    const double *uin = (k & 1) ? d.uu2 : d.uu;
    const double *gam = d.scal + S_HIST, *alp = d.scal + S_HIST + kMaxCg + 1, *bet = d.scal + S_HIST + 2 * (kMaxCg + 1);
    Tok t;
    if (KNOCKED(512)) { for (int q = 0; q < kPart; q++) t.pd.v[q] = 1.0; } else t.pd = partial_load(d.part + SL_DELTA * kGrid);
    t.gamma = gam[k]; t.beta = bet[k]; t.alast = k == 0 ? 1.0 : alp[k - 1];
    const int j0 = first_index();
    t.u0 = t.p0 = t.r0 = t.s0 = t.m0 = t.x0 = 0.0;
    if (has_vec && j0 < d.n) { t.u0 = uin[j0]; t.p0 = k == 0 ? 0.0 : d.p[j0]; t.r0 = d.r[j0]; t.s0 = d.s[j0]; t.m0 = d.Minv[j0]; t.x0 = d.xs[j0]; }
    return t;
  }
  __device__ __forceinline__ bool finish(const Tok &t, const double (&)[1], bool) const {
    const double *uin = (k & 1) ? d.uu2 : d.uu;
    double *uout = (k & 1) ? d.uu : d.uu2;
    double *alp = d.scal + S_HIST + kMaxCg + 1;
    const PartRegs &pd = t.pd;
    const double gamma = t.gamma, beta = t.beta, alast = t.alast;
    const int nchunk = (d.n + kBlock - 1) / kBlock, per = (nchunk + 7) >> 3, slots = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
    const int j0 = first_index();
    const bool live0 = has_vec && j0 < d.n;
    const double u0 = t.u0, p0 = t.p0, r0 = t.r0, s0 = t.s0, m0 = t.m0, x0 = t.x0;
    const double delta = block_sum(partial_fold_sum(pd), red);
    const double alpha = k == 0 ? gamma / delta : gamma / (delta - beta * gamma / alast);
    if (blockIdx.x == 0 && threadIdx.x == 0) alp[k] = alpha;
    e->alpha = alpha;
    double gg = 0, rr = 0;
    if (live0) {
      const double pp_ = k == 0 ? u0 : u0 + beta * p0;
      const double rr_ = r0 - alpha * s0, un = m0 * rr_;
      stw(d.wt, d.p + j0, pp_); stw(d.wt, d.xs + j0, x0 + alpha * pp_); stw(d.wt, d.r + j0, rr_); stw(d.wt, uout + j0, un);
      gg += rr_ * un; rr = nanmax(rr, fabs(rr_));
    }
    if (has_vec) {
      for (int sl = slot0 + slots; sl < per; sl += slots) {
        const int c = xcd * per + sl;
        if (c >= nchunk) break;
        const int j = c * kBlock + threadIdx.x;
        if (j >= d.n) break;
        const double u = uin[j];
        const double pp_ = k == 0 ? u : u + beta * d.p[j];
        const double rr_ = d.r[j] - alpha * d.s[j], un = d.Minv[j] * rr_;
        stw(d.wt, d.p + j, pp_); stw(d.wt, d.xs + j, d.xs[j] + alpha * pp_); stw(d.wt, d.r + j, rr_); stw(d.wt, uout + j, un);
        gg += rr_ * un; rr = nanmax(rr, fabs(rr_));
      }
    }
    *g = gg; *rn = rr;
    return true;
  }
};
__global__ __launch_bounds__(kBlock) void k_k1f(Dev d, int i) {          // i >= 1; performs the vector update of k = i - 1
  __shared__ StreamLdsW<1, double> lds;
  KT(8);
  const int k = i - 1;
  const bool has_rows = wg_has_rows(d.A);
  const int nchunk = (d.n + kBlock - 1) / kBlock, per = (nchunk + 7) >> 3;
  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
  const bool has_vec = slot0 < per && xcd * per + slot0 < nchunk;
  if (!has_rows && !has_vec && blockIdx.x != 0) {     // (partials of a finished PCG are never read: no flag test needed)
    put_partial(d.part, SL_GAMMA0 + (i & 1), 0.0); put_partial(d.part, SL_RN0 + (i & 1), 0.0);
    return;
  }
  double g = 0, rn = 0;
  GMs gr{d.ms};
  EK1F e{d.rho, d.t}; e.wt = d.wt;
  if (!process_rows<1>(d.A, gr, e, lds, PreK1F{d, k, has_vec, &e, lds.red, &g, &rn}, d.flags + F_DONE)) return;
  __syncthreads();
  block_sum_max(g, rn, lds.red);
  put_partial(d.part, SL_GAMMA0 + (i & 1), g); put_partial(d.part, SL_RN0 + (i & 1), rn);
  KT(14);
}

// KA ------------------------------------------------------------------------------------------
struct EKa {
  const double *l, *u, *rho, *rho_inv; double *z, *y, *zt, *t0, *v, *dy; double alpha; double *ztg; double theta; int wt = 0;
  double pl = 0, pu = 0, prho = 0, prinv = 0, pz = 0, py = 0, pzt = 0;
  __device__ __forceinline__ void prefetch(int i) { pl = l[i]; pu = u[i]; prho = rho[i]; prinv = rho_inv[i]; pz = z[i]; py = y[i]; pzt = zt[i]; }
  __device__ __forceinline__ void operator()(int i, const double (&s)[1]) {
    const double ztil = s[0];
    const double zr = alpha * ztil + (1.0 - alpha) * pz;                    // _osqp.py:686-690
    const double zn = fmin(fmax(zr + prinv * py, pl), pu);                   // :674
    const double dyi = prho * (zr - zn), yn = py + dyi;                      // :698-703
    const double zg = ztil + theta * (ztil - pzt);                           // A xg (Dev::ztg)
    stw(wt, y + i, yn); stw(wt, dy + i, dyi); stw(wt, z + i, zn); stw(wt, zt + i, ztil); stw(wt, v + i, prho * zn - yn); stw(wt, ztg + i, zg); stw(wt, t0 + i, prho * zg);
  }
};
// The extrapolated PCG start is used after a solve that REACHED its tolerance -- and after a cut-off one only while the start
// residuals keep falling (slot form, below).  A cut-off solve leaves an error that the few
// iterations it was given barely touched in the slow modes of K; extrapolating along a step that contains it feeds that error, times
// (1 + theta), to the next cut-off solve, and through z and y back into the next right-hand side: observed as iterates growing to
// 1e12 within 25 ADMM iterations after a rho update had left five-iteration budgets on an ill-conditioned system (then 500
// iterations of recovery; unstructured config 2 with cg_tol_fraction 0.1).  Limiting theta by the measured residual reduction of
// the cut-off solve did not prevent it (the residual norm says little about the slow modes); starting from x~ itself does.
// Every workgroup takes the same branch (conv / done come from an earlier launch); rn, bn of the last iterate are folded for
// workgroup 0's statistics.
__device__ __forceinline__ double cutoff_theta(const Dev &d, int slot, double *red, double &rn, double &bn, int admm = -1) {
  rn = partial_fold_max(partial_load(gptr(d.part) + (SL_RN0 + slot) * kGrid)); bn = partial_fold_max(partial_load(gptr(d.part) + SL_BN * kGrid));
  block_max2(rn, bn, red);
  if (admm < 1) return 0.0;
  // slot form: the start residuals of this and of the previous ADMM iteration are on record (written by earlier launches).  While they
  // FALL the cut-off solves are keeping up and the extrapolation stays (config 2: budget-limited chunks are part of normal operation,
  // 53 vs 65 ms); once the start residual grows, the next solve starts from x~ itself.
  const double now = gptr(d.scal)[S_RN0H + (admm & 1)], prev = gptr(d.scal)[S_RN0H + ((admm + 1) & 1)];
  return (now < prev) ? d.theta : 0.0;
}
__global__ __launch_bounds__(kBlock) void k_ka(Dev d, int budget) {
  __shared__ StreamLdsW<1, double> lds;
  int done = d.flags[F_DONE];                       // (set by an earlier launch: the same value in every wave)
  double theta = d.theta, rn_last = 0.0, bn_last = 0.0;
  if (!done && budget > 0) { theta = cutoff_theta(d, budget & 1, lds.red, rn_last, bn_last); __syncthreads(); }
  GVec g{d.xs};
  EKa e{d.l, d.u, d.rho, d.rho_inv, d.z, d.y, d.zt, d.t0, d.v, d.dy, d.alpha, d.ztg, theta}; e.wt = d.wt;
  process_rows<1>(d.A, g, e, lds);
  const int stride = gridDim.x * kBlock;
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < d.n; j += stride) {    // _osqp.py:664-668
    const double xt = d.xs[j], xo = d.x[j], xn = d.alpha * xt + (1.0 - d.alpha) * xo;
    stw(d.wt, d.dx + j, xn - xo); stw(d.wt, d.x + j, xn);
    stw(d.wt, d.xg + j, xt + theta * (xt - d.xsp[j])); stw(d.wt, d.xsp + j, xt);                   // next PCG start (Dev::xg)
  }
  if (blockIdx.x == 0) {                                                     // PCG statistics of this ADMM iteration
    if (!done && budget > 0) {            // did the last budgeted iteration reach the tolerance? (no K1 ran after it)
      const double rn = rn_last, bn = bn_last;
      done = !(rn > fmax(d.scal[S_TOL_REL] * bn, d.scal[S_TOL_ABS])) ? 2 : 0;
      if (!done && threadIdx.x == 0 && rn > 0.1 * d.scal[S_RN0]) d.flags[F_STAT_STAG] += 1;
    }
    if (threadIdx.x == 0) {
      const int used = done == 1 ? d.flags[F_ITERS] : budget;
      d.flags[F_STAT_SUM] += used; d.flags[F_STAT_SUMSQ] += used * used; d.flags[F_STAT_N] += 1;
      if (used > d.flags[F_STAT_MAX]) d.flags[F_STAT_MAX] = used;
      if (!done) d.flags[F_STAT_UNCONV] += 1;
    }
  }
}

// ---------------------------------------------------------------------------------------------- slot kernels
// Device-side scheduling of the ADMM / PCG phases (no launch is wasted on a converged PCG).  A chunk of ADMM iterations is a fixed
// string of launches  B A B A ...  ("slots"): a B slot streams B = [P + sigma I | A'] and runs whichever B-phase is due (KB, or
// the K2F of the current PCG iteration), an A slot streams A and runs the A-phase that is due (the first K1, a K1F, or KA).  Which
// phase is due is a small record in device memory: every slot reads the record its predecessor wrote (kernel boundary = ordering),
// its workgroup 0 writes the successor's -- two records, so that no workgroup of a launch can observe its own launch's update.
// The PCG of ADMM iteration j therefore takes exactly as many slot pairs as it has iterations (plus the pair that detects
// convergence and runs KA), whatever the neighbouring iterations needed; only the few slots left over at the END of a chunk idle.
__global__ void k_slot_init(int *slot, int target, int cap, int epoch) {      // (cap: PCG iterations per solve; in the record, not a kernel argument, so that captured strings of slots serve every chunk)
  slot[SR_PHASE] = P_KB; slot[SR_K] = 0; slot[SR_ADMM] = 0; slot[SR_TARGET] = target; slot[SR_USED] = 0; slot[SR_CONV] = 0; slot[SR_CAP] = cap; slot[SR_SEQ] = 0;
  slot[SR_WORDS + SR_SEQ] = 0; slot[SR_WORDS + SR_ADMM] = 0;        // (record B still holds the previous chunk's last state: slot_poll takes the newer record)
  slot[2 * SR_WORDS] = epoch;                                        // which chunk the records belong to (slot_poll)
}

__global__ __launch_bounds__(kBlock) void k_slot_b(Dev d) {
  __shared__ union { StreamLds<2> kb; StreamLdsW<1, double> k2f; } lds;
  const FirstDesc fd = first_desc<true>(d.B);          // (both B phases start from the same descriptors: requested before the branch)
  SlotState st = slot_read(d.slot);                    // written by the previous A slot (or k_slot_init)
  int *W = d.slot + SR_WORDS;
  if (st.ph == P_KB) {
    if (st.admm >= st.target) { st.ph = P_IDLE; slot_write(W, st); return; }
    GKb g{d.xg, d.v, d.t0, d.n};
    EKb e{d.x, d.q, d.Minv, d.r, d.uu, d.sigma, d.xg, d.xs}; e.wt = d.wt;
    process_rows_fd<2>(d.B, g, e, lds.kb, NoPre(), fd);
    __syncthreads();
    const double G = block_sum(e.g, lds.kb.red);
    double RN = e.rn, BN = e.bn;
    block_max2(RN, BN, lds.kb.red);
    put_partial(d.part, SL_GAMMA0, G); put_partial(d.part, SL_RN0, RN); put_partial(d.part, SL_BN, BN);
    if (blockIdx.x == 0 && threadIdx.x == 0) { d.flags[F_DONE] = 0; d.flags[F_ITERS] = 0; }
    st.ph = P_K1; st.k = 0;
  } else if (st.ph == P_K2F) {
    const int k = st.k;
    const double *u = (k & 1) ? d.uu2 : d.uu;
    GSplitU g{u, d.t, d.n};
    EK2F e{u, d.Minv, d.s, d.ms}; e.wt = d.wt;
    double dl_first = 0.0;
    if (!process_rows_fd<1>(d.B, g, e, lds.k2f, PreK2F{d, k, &e, lds.k2f.red, &dl_first}, fd)) {   // converged after k iterations
      st.ph = P_KA; st.used = k; st.conv = 1;
      slot_write(W, st);
      return;
    }
    double DL = dl_first;
    if (!d.B.single) { __syncthreads(); DL += block_sum(e.dl, lds.k2f.red); }
    put_partial(d.part, SL_DELTA, DL);
    st.ph = P_K1F;
  }
  slot_write(W, st);                                   // (P_KA pending after a PCG that hit the cap, P_IDLE: passed through)
}

// KA with the PCG statistics taken from the slot record (used iterations; conv = 0: the PCG stopped at the cap -- did its last
// iterate reach the tolerance anyway?)
template <class L>
__device__ __forceinline__ void slot_ka(const Dev &d, L &lds, int used, int conv, const FirstDesc &fd, int admm, int target, int seq, int rn_slot = -1) {
  double theta = d.theta, rn_last = 0.0, bn_last = 0.0;
  // (rn_slot: which of the two ||r|| partial buffers the last PCG launch wrote -- the parity of `used` in the two-kernel form, of the LAUNCH in the F1 form)
  if (!conv) { theta = cutoff_theta(d, rn_slot >= 0 ? rn_slot : (used & 1), lds.red, rn_last, bn_last, admm); __syncthreads(); }      // (conv comes from the slot record: uniform)
  GVec g{d.xs};
  EKa e{d.l, d.u, d.rho, d.rho_inv, d.z, d.y, d.zt, d.t0, d.v, d.dy, d.alpha, d.ztg, theta}; e.wt = d.wt;
  process_rows_fd<1>(d.A, g, e, lds, NoPre(), fd);
  const int stride = gridDim.x * kBlock;
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < d.n; j += stride) {    // _osqp.py:664-668
    const double xt = d.xs[j], xo = d.x[j], xn = d.alpha * xt + (1.0 - d.alpha) * xo;
    stw(d.wt, d.dx + j, xn - xo); stw(d.wt, d.x + j, xn);
    stw(d.wt, d.xg + j, xt + theta * (xt - d.xsp[j])); stw(d.wt, d.xsp + j, xt);                   // next PCG start (Dev::xg)
  }
  if (blockIdx.x == 0) {
    if (!conv) {
      const double rn = rn_last, bn = bn_last;
      conv = !(rn > fmax(d.scal[S_TOL_REL] * bn, d.scal[S_TOL_ABS])) ? 2 : 0;
      if (!conv && threadIdx.x == 0 && rn > 0.1 * d.scal[S_RN0]) d.flags[F_STAT_STAG] += 1;
    }
    if (threadIdx.x == 0) {
      d.flags[F_STAT_SUM] += used; d.flags[F_STAT_SUMSQ] += used * used; d.flags[F_STAT_N] += 1;
      if (used > d.flags[F_STAT_MAX]) d.flags[F_STAT_MAX] = used;
      if (!conv) d.flags[F_STAT_UNCONV] += 1;
      if (d.ctl && admm + 1 >= target) { d.ctl->seq_end = seq + 1; d.ctl->chunk_done = 1; }      // device-driven boundaries: the chunk's last ADMM iteration (read by LATER launches)
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_slot_a(Dev d) {
  __shared__ union { StreamLdsW<1, double> k1; StreamLdsW<1, double> k1f; } lds;
  const FirstDesc fd = first_desc<true>(d.A);
  SlotState st = slot_read(d.slot + SR_WORDS);         // written by the previous B slot
  int *W = d.slot;
  if (st.ph == P_K1) {                                 // first A-apply of this ADMM iteration's PCG: stopping test on r_0, t_0 = rho .* (A u_0)
    GVec g{d.uu};
    EK1 e{d.rho, d.t};
    if (process_rows_fd<1>(d.A, g, e, lds.k1, PreK1{d, 0, 0, lds.k1.red, st.admm & 1}, fd)) { st.ph = P_K2F; st.k = 0; }
    else {                                             // the warm start already meets the tolerance: no PCG iteration, KA right here
      __syncthreads();
      slot_ka(d, lds.k1, 0, 1, fd, st.admm, st.target, st.seq);
      st.ph = P_KB; st.admm += 1;
    }
  } else if (st.ph == P_K1F) {
    const int k = st.k, i = k + 1;
    const int nchunk = (d.n + kBlock - 1) / kBlock, per = (nchunk + 7) >> 3;
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
    const bool has_vec = slot0 < per && xcd * per + slot0 < nchunk;
    if (!wg_has_rows(d.A) && !has_vec && blockIdx.x != 0) {
      put_partial(d.part, SL_GAMMA0 + (i & 1), 0.0); put_partial(d.part, SL_RN0 + (i & 1), 0.0);
    } else {
      double g = 0, rn = 0;
      GMs gr{d.ms};
      EK1F e{d.rho, d.t}; e.wt = d.wt;
      process_rows_fd<1>(d.A, gr, e, lds.k1f, PreK1F{d, k, has_vec, &e, lds.k1f.red, &g, &rn}, fd);
      __syncthreads();
      block_sum_max(g, rn, lds.k1f.red);
      put_partial(d.part, SL_GAMMA0 + (i & 1), g); put_partial(d.part, SL_RN0 + (i & 1), rn);
    }
    if (i >= st.cap) { st.ph = P_KA; st.used = i; st.conv = 0; }      // the PCG stops at the cap; the next A slot runs KA
    else { st.ph = P_K2F; st.k = i; }
  } else if (st.ph == P_KA) {
    slot_ka(d, lds.k1, st.used, st.conv, fd, st.admm, st.target, st.seq);
    st.ph = P_KB; st.admm += 1;
  }
  slot_write(W, st);
}


// ---------------------------------------------------------------------------------------------- one launch per PCG iteration (F1)
// backend.h DevF1.  Launch F_k of the PCG of one ADMM iteration (k = 0 .. iterations):
//   scalars   k = 0:  ||r_0||, ||rhs|| (KB's partials) -> tolerance, stopping test
//             k >= 1: gamma_{k-1}, delta_{k-1}, ||r_{k-1}|| (partials of F_{k-1}) -> stopping test (k >= 2), beta_{k-1}, alpha_{k-1}
//   window    u_k[c] = Minv (r_{k-1} - alpha (w_{k-1} + beta s_{k-2}))[c],  w_{k-1} = sum_d rep_d = K u_{k-1}   (k = 0: Minv r_0)
//             for every column c of the block's GATHER window (the columns its rows of A and its own rows of P touch) -- recomputed
//             by every workgroup that gathers c, with the same instruction sequence as the owner's update (f1_upd): all copies
//             are bit-identical.  The lane whose window column is one of the block's OWN columns also performs that column's
//             vector update:  s_{k-1}, r_k, p_{k-1}, x~ += alpha p_{k-1}  stored;  partials gamma_k = <r_k, Minv r_k>, ||r_k||_inf
//   SpMV      t = rho .* (A_g u_k)  (rows of the block, products staged in LDS, one lane per row),
//             pu_k = (P + sigma I) u_k  on the own columns,  rep_{g mod D} = A_g' t (+ pu_k on the own columns)  per column of the block's
//             SCATTER window (the columns of its rows of A, which include its own columns; second, column-ordered pass over the
//             entries still held in registers),  partial delta_k = <t, A u_k> + <u_k, pu_k>_own = <u_k, K u_k>
// r, s, rep are double-buffered by the parity of k: a workgroup reads what the PREVIOUS launch wrote while its neighbours
// write this launch's values.  Four workgroup barriers per block, no global synchronisation inside the launch.
// Template: D = replicas, FIRST = the launch F_0, MIX = per-block mixing (straight-line code: no run-time branch on any of them).  MIX (backend.h
// DevF1::mix): the block's far columns -- columns of its rows outside its window -- take the LAST kF1MaxFar gather slots (lanes kBlock / 2 .. kBlock - 1
// in their second window element: the same 4 + D loads at column fcol[.]) and nfc more segments of the column-ordered pass, whose sums go to the
// block's spill slots; every reconstruction of a column is  f1_w(replicas) + f1_spill_sum(column's slots in index order).
struct F1Lds {
  double win[kF1Win];            // u_k on the block's gather window (MIX: far columns in the last kF1MaxFar slots)
  double prod[kF1Chunk];         // A products in row-major entry order, then val * t[row] in column-major order
  double tvec[kF1MaxRows];       // t of the block's rows
  double uown[kF1MaxOwn];        // u_k on the own columns
  double puown[kF1MaxOwn];       // (P + sigma I) u_k on the own columns: added to the block's own slice of A' t (the own columns lie inside its scatter window)
  double pprod[kF1PChunk];       // (P + sigma I) products of the own rows
  double red[3 * kWaves];
};
// A block's matrix stream in LDS: the image of its slice of DevF1::stream, filled by LDS-direct loads (no VGPR destination).
struct F1Stream { double val[kF1Chunk]; unsigned int ent[kF1Chunk]; };
static_assert(sizeof(F1Stream) == kF1StreamBytes, "LDS image and global layout of a block's stream must agree");
static_assert(kF1StreamBytes % (kWaves * 1024) == 0, "every wave issues whole 1 KB pieces");
// One 1 KB piece per wave instruction: lane l moves the 16 bytes at gsrc (its own address) to  LDS[lds_base + 16 l]  (lds_base wave-uniform, in
// M0).  Inline asm: hipcc's builtin would make every later barrier drain the transfer (it counts it against vmcnt and waits before
// __syncthreads); as asm the transfer is invisible to the compiler's wait insertion -- its own waits can only become stricter, never too weak
// (it assumes fewer operations in flight than there are) -- and the consumer waits with f1_stream_wait.  M0 is saved and restored.
// NT: the load carries the non-temporal bit -- the matrix stream is read once per launch and, beyond the Infinity Cache (n = 1M: 126 of the 320 MB a launch
// moves), only evicts the window vectors its neighbours are about to gather: 96 -> 85 us per launch there; at n = 100k (everything cache-resident) it costs
// 0.15 us, so the bit follows the write-through switch (WT = false: the HBM-resident regime).
template <bool NT>
__device__ __forceinline__ void lds_dma16(const void *gsrc, unsigned lds_base) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned keep;
  if constexpr (NT) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
  else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
#else
  (void)gsrc; (void)lds_base;
#endif
}
__device__ __forceinline__ unsigned lds_offset_of(const void *p) {       // byte offset of a __shared__ object inside the workgroup's LDS allocation
#if defined(__HIP_DEVICE_COMPILE__)
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void *)p;
#else
  (void)p; return 0u;
#endif
}
// request block b's stream: kF1StreamBytes / (kWaves * 1024) pieces per wave, no register holds anything afterwards
template <bool NT>
__device__ __forceinline__ void f1_stream_issue(const unsigned char *stream, int b, F1Stream &S) {
  constexpr int kPieces = kF1StreamBytes / (kWaves * 1024);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned char *g = stream + (size_t)b * kF1StreamBytes + (size_t)wave * (kPieces * 1024) + (threadIdx.x & 63) * 16;
  const unsigned l0 = lds_offset_of(&S) + (unsigned)wave * (kPieces * 1024);
#pragma unroll
  for (int q = 0; q < kPieces; q++) lds_dma16<NT>(g + q * 1024, __builtin_amdgcn_readfirstlane(l0 + q * 1024));
}
// everything this wave has requested has landed (its own pieces of the stream included); the workgroup barrier that follows publishes them
__device__ __forceinline__ void f1_stream_wait() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
struct F1Scal { double alpha, beta; int general; };       // general = 0: the first update (s_0 = w_0, p_0 = u_0: s_{-1}, p_{-1} are not used)
template <int D>
__device__ __forceinline__ double f1_w(const double (&rp)[D]) {     // w_{k-1} = K u_{k-1}: the replicas in index order (deterministic)
  double w = rp[0];
#pragma unroll
  for (int q = 1; q < D; q++) w += rp[q];
  return w;
}
__device__ __forceinline__ void f1_upd(const F1Scal &sc, double minv, double r, double w, double sp, double &sn, double &rn, double &un) {
  sn = fma(sc.beta, sc.general ? sp : 0.0, w);           // (first update: beta = 0 and the stale s is masked, so s_0 = w_0 exactly)
  rn = fma(-sc.alpha, sn, r);
  un = minv * rn;
}
// sum of seg[a .. z): the first kB entries with independent LDS reads (as process_rows)
template <int kB>
__device__ __forceinline__ double f1_segsum(const double *seg, int a, int z) {
  double v[kB];
#pragma unroll
  for (int q = 0; q < kB; q++) v[q] = a + q < z ? seg[a + q] : 0.0;
  double acc = 0.0;
#pragma unroll
  for (int q = 0; q < kB; q++) acc += v[q];
  for (int k = a + kB; k < z; k++) acc += seg[k];
  return acc;
}
// The scalar part of launch F_k.  The per-workgroup partials (gamma, delta, ||r||) are double-buffered by the parity of the LAUNCH, not of k:
// a launch reads what the previous launch of the string wrote -- KB: gamma_0, ||r_0||, ||rhs|| (in delta's slot); F_k: gamma_k, delta_k, ||r_k|| --
// so the three loads depend on nothing but the kernel's `par` argument and leave at the very head of the launch (f1_fold_issue), next to
// the phase record instead of behind it.  f1_fold_finish returns false when the PCG had already converged (the caller runs KA in this launch).
struct F1Fold { PartRegs a, b, c; };
__device__ __forceinline__ F1Fold f1_fold_issue(const double *part, const int par, const int probe) {
  F1Fold f;
#pragma unroll
  for (int q = 0; q < kPart; q++) { f.a.v[q] = 0.0; f.b.v[q] = 0.0; f.c.v[q] = 0.0; }
  if (probe == 1) return f;                                 // (probe == 2 pays for the fold like a solve's launch, then uses the fixed scalars)
  const int prev = par ^ 1;
  f.a = partial_load(part + (SL_GAMMA0 + prev) * kGrid); f.b = partial_load(part + (SL_DELTA + prev) * kGrid); f.c = partial_load(part + (SL_RN0 + prev) * kGrid);
  return f;
}
__device__ __forceinline__ bool f1_fold_finish(const Dev &d, const int k, const int admm_par, const int probe, const F1Fold &f, double *red, F1Scal &sc) {
  double *gam = gptr(d.scal) + S_HIST, *alp = gptr(d.scal) + S_HIST + kMaxCg + 1;
  const int tid = threadIdx.x;
  sc = F1Scal{0.0, 0.0, k >= 2};
  if (probe == 1) { sc.alpha = 1e-3; sc.beta = k >= 2 ? 0.5 : 0.0; return true; }
  if (k == 0) return true;                                  // F_0 builds r_0 itself (from the slices KA / the chunk's first launch left): nothing to fold yet
  double tol = gptr(d.scal)[S_TOL_NOW];
  const double glast = k >= 2 ? gam[k - 2] : 1.0, alast = k >= 2 ? alp[k - 2] : 1.0;
  double gamma = partial_fold_sum(f.a), rn = partial_fold_max(f.c), delta = partial_fold_sum(f.b);
  block_sum_max_sum(gamma, rn, delta, red);
  if (k == 1 && probe == 0) {
    // the tolerance of this ADMM iteration's PCG, from ||rhs|| (F_0's partials) -- and the test of the START: r_0 may already meet it
    const double bn = block_max(partial_fold_max(partial_load(gptr(d.part) + SL_BN * kGrid)), red);
    tol = fmax(gptr(d.scal)[S_TOL_REL] * bn, gptr(d.scal)[S_TOL_ABS]);
    if (blockIdx.x == 0 && tid == 0) { gptr(d.scal)[S_TOL_NOW] = tol; gptr(d.scal)[S_RN0] = rn; gptr(d.scal)[S_RN0H + admm_par] = rn; }
    if (!(rn > tol)) return false;                          // no PCG iteration (a NaN also ends the inner loop): the caller runs KA in this launch, x~ = the start
  }
  if (probe) {                                              // timing probe: the fold above was paid for; bounded, repeatable scalars instead of its result
    if (rn < -1.0) gptr(d.res)[R_COUNT - 1] = gamma + delta + tol + glast + alast;      // (never true: keeps the fold alive)
    sc.alpha = 1e-3; sc.beta = k >= 2 ? 0.5 : 0.0;
    return true;
  }
  if (k >= 2 && !(rn > tol)) return false;                  // converged after k - 1 iterations
  sc.beta = k >= 2 ? gamma / glast : 0.0;
  sc.alpha = k >= 2 ? gamma / (delta - sc.beta * gamma / alast) : gamma / delta;
  if (blockIdx.x == 0 && tid == 0) { gam[k - 1] = gamma; alp[k - 1] = sc.alpha; }
  return true;
}
// base[idx] as int4 through the CONSTANT address space: with a wave-uniform index the compiler emits scalar loads
__device__ __forceinline__ int4 sload_int4(const int *base, size_t idx) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef int __attribute__((ext_vector_type(4))) v4i;
  typedef const v4i __attribute__((address_space(4))) *cptr;
  const v4i v = ((cptr)(unsigned long long)base)[idx];
  return make_int4(v.x, v.y, v.z, v.w);
#else
  return reinterpret_cast<const int4 *>(base)[idx];
#endif
}
struct F1Rec { int4 ds, fa, fb, fc; };      // a block's record (DevF1::blk)
__device__ __forceinline__ F1Rec f1_record(const DevF1 &f, int b) {
  return F1Rec{sload_int4(f.blk, 4 * (size_t)b), sload_int4(f.blk, 4 * (size_t)b + 1), sload_int4(f.blk, 4 * (size_t)b + 2), sload_int4(f.blk, 4 * (size_t)b + 3)};
}
// per-block mixing (backend.h DevF1::mix): what the far columns' spill slots hold for one column, summed in index order.  EVERY reconstruction of a
// column is  f1_w(replicas) + f1_spill_sum(...)  -- the same expression in every workgroup that needs the column: all copies are bit-identical
// A column's spill slots arrive as ONE packed word (DevF1::spk: first slot << 6 | count); the first two slots are requested together, behind the
// packed word alone -- a loop would put one memory round trip per slot on the
// block's critical path.  f1_spill_take: those two loads; f1_spill_sum: the sum.
struct F1Spill { double v0, v1; };
__device__ __forceinline__ F1Spill f1_spill_take(const double *sp, int word) {     // (lanes without slots request nothing)
  const int q0 = word >> 6, cnt = word & 63;
  F1Spill t{0.0, 0.0};
  if (cnt > 0) t.v0 = sp[q0];
  if (cnt > 1) t.v1 = sp[q0 + 1];
  return t;
}
__device__ __forceinline__ double f1_spill_sum(const double *sp, int word, const F1Spill &t) {
  const int q0 = word >> 6, cnt = word & 63;               // (the plan refuses columns with more than 63 slots)
  double a = 0.0;
  if (cnt > 0) a += t.v0;
  if (cnt > 1) a += t.v1;
  for (int q = q0 + 2; q < q0 + cnt; q++) a += sp[q];
  return a;
}
__device__ __forceinline__ double f1_spill_sum(const double *sp, const int *spk, int c) {     // (columns off the fast path)
  const int word = spk[c];
  return f1_spill_sum(sp, word, f1_spill_take(sp, word));
}
// (Tried: the scalar fold INSIDE the body, behind the first block's vector requests, so that the two round trips overlap -- the fold's 24 partial
//  registers on top of the window's cost 36 - 128 bytes of scratch per lane and the launch time did not move: 24.97 vs 24.66 us per pair.  In two steps --
//  the lane's share of the partials reduced to three doubles BEFORE the requests, the block reduction behind them -- no scratch, but the solve got
//  slower: 44.3 -> 46.4 ms, 12.42 -> 13.03 us per F launch: the requests then wait for the partials instead of the other way round.
//  Also tried: the NEXT block's record and the first window element of every lane (3 + D values) requested during the current block, next to the next
//  block's stream (n >= 1M, ten blocks per workgroup): 123 VGPRs, no scratch -- n = 1M unchanged (98.3 us per launch), n = 100k 12.5 -> 13.0 us.
//  And: everything the F body reads of Dev (24 pointers / sizes) as ONE block of three s_load_dwordx16 behind one wait, instead of the compiler's fetches
//  at first use (nine s_load -> wait -> vector-load rounds in the ISA of the load phase): at the head of the launch 12.4 -> 14.2 us per F launch, at the
//  start of the F phase 13.7 us -- the just-in-time fetches overlap the issue of the vector loads, the block does not; 104 spilled SGPRs, 126 VGPRs.)
// Stores of a launch's results.  What a launch writes is read by the NEXT launch on other XCDs, i.e. through memory, never from this XCD's L2 -- and the
// end of a kernel waits until its dirty lines have been written back.  WT: relaxed agent-scope stores (global_store ... sc1: written through the L2 as they are
// produced), so that the kernel's end finds nothing to write back: KA launch 12.8 -> 10.9 us, F launch 12.7 -> 12.4 us, configs[1] 44.5 -> 43.0 ms per cold solve.
// Only while the working set sits in the Infinity Cache (DevF1::wt, Engine::upload_f1): at n = 1M, where the launch is bound by HBM bandwidth, the same
// stores cost 4 - 6 us per launch (95 -> 100 us).  (Non-temporal stores -- the nt bit -- changed nothing: 12.55 us.)
template <bool WT>
__device__ __forceinline__ void gst_(double *p, double v) {
  if constexpr (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
#define gst gst_<WT>
template <int D, bool FIRST, bool MIX, bool WT>
__device__ __forceinline__ void f1_body(const Dev &d, const int k, const bool vec_only, const F1Scal sc, F1Lds &L, F1Stream &S, const F1Rec &rec0, const int par) {
  const DevF1 &f = d.f1;
  const int n = d.n, tid = threadIdx.x;
  const int cur = (k + 1) & 1, nxt = k & 1;               // parity of k - 1 / of k
  KT(1);
  // every n-vector of the iteration lives in ONE arena (DevF1::va, stride ns): the addresses derive from one base pointer by scalar
  // adds instead of a kernel-argument load per vector
  const double *va = gptr(f.va); const size_t ns = f.ns;
  const double *Minv = va, *xs_r = va + ns, *p_r = va + 2 * ns;
  double *xs_w = gptr(f.va) + ns, *p_w = gptr(f.va) + 2 * ns;
  const double *rread = va + (3 + (cur == 0 ? 0 : 1)) * ns;                 // r_{k-1}
  double *rnxt = gptr(f.va) + (3 + nxt) * ns;                     // r_k
  const double *sprev = va + (5 + cur) * ns;                // s_{k-2}: stored next to r_{k-1}
  double *snew = gptr(f.va) + (5 + nxt) * ns;                     // s_{k-1}: stored next to r_k
  const double *repcur = va + (7 + (size_t)cur * D) * ns;   // K u_{k-1} in D partial vectors  (F_0: the slices of r_0 = rhs - K x_g that f1_ka_body left)
  const double *repV = va + (7 + (size_t)2 * D) * ns;       // F_0: the slices of rhs = sigma x - q + A' v alone (||rhs||_inf: read on the own columns only)
  double *repnxt = gptr(f.va) + (7 + (size_t)nxt * D) * ns;
  // per-block mixing: the spill sets that go with the three replica sets, the spill slots by column
  [[maybe_unused]] const double *spcur = MIX ? gptr(f.spill) + (size_t)cur * (f.nsp + 2) : nullptr, *spV = MIX ? gptr(f.spill) + 2 * (f.nsp + 2) : nullptr;
  [[maybe_unused]] double *spnxt = MIX ? gptr(f.spill) + (size_t)nxt * (f.nsp + 2) : nullptr;
  [[maybe_unused]] const int *spk = MIX ? gptr(f.spk) : nullptr;
  double g_acc = 0.0, rn_acc = 0.0, dl_acc = 0.0, bn_acc = 0.0;
  // the vector update of one own column (operands in registers): stores s_{k-1}, r_k, p_{k-1}, x~; returns u_k
  auto own_update = [&](int j, double mi, double r, double w, double sp, double pp, double x) -> double {
    double sn, rn, un;
    f1_upd(sc, mi, r, w, sp, sn, rn, un);
    const double pn = fma(sc.beta, sc.general ? pp : 0.0, mi * r);
    gst(xs_w + j, fma(sc.alpha, pn, x)); gst(p_w + j, pn); gst(snew + j, sn); gst(rnxt + j, rn);
    g_acc += rn * un; rn_acc = nanmax(rn_acc, fabs(rn));
    return un;
  };
  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int per = (d.A.nblk + 7) >> 3;
  constexpr int CW = kF1Win / kBlock, CE = kF1Chunk / kBlock;
  constexpr int kFB = kF1Win - kF1MaxFar;                   // first far slot
  static_assert(kFB >= (CW - 1) * kBlock, "the far slots belong to the lanes' last window element");
  static_assert(kF1PChunk == kBlock, "one (P + sigma I) entry per lane");
  for (int sl = slot0; sl < per; sl += slots) {
    const int b = __builtin_amdgcn_readfirstlane(xcd * per + sl);
    if (b >= d.A.nblk) break;
    // MIX: the far slots this lane serves -- {column, its packed spill word} for the gather, the spill slot for the column-ordered pass -- sit at a
    // fixed stride per block: requested before the record (the window loads of those lanes wait for them, not for a second round trip)
    [[maybe_unused]] int2 fcl = make_int2(0, 0); [[maybe_unused]] int fql = 0;
    if constexpr (MIX) { if (tid >= kFB - (CW - 1) * kBlock) { const size_t i = (size_t)b * kF1MaxFar + (tid - (kFB - (CW - 1) * kBlock)); fcl = reinterpret_cast<const int2 *>(gptr(f.fcol))[i]; fql = gptr(f.fq)[i]; } }
    // the block's record: 16 words, one scalar load
    // (read through the constant address space: the index is wave-uniform, so the four int4 become scalar loads behind ONE wait --
    //  as generic-pointer loads inside this loop they were four vector loads, each waited for before the next was issued)
    const F1Rec rec = sl == slot0 ? rec0 : f1_record(f, b);      // (the first block's record was requested before the scalar fold)
    const int4 ds = rec.ds, fa = rec.fa, fb = rec.fb, fc = rec.fc;
    const int r0 = ds.x, nrows = ds.y - ds.x, k0 = ds.z, cnt = ds.w - ds.z;
    const int cov0 = fa.x, cov1 = fa.y, cs0 = fa.z, nown = fa.w - fa.z;
    const int cpo = fb.x, pk0 = fb.y, pcnt = fb.z - fb.y;
    const int g0 = fc.x, gl = vec_only ? 0 : fc.y, a0 = fc.z, wl = fc.w;      // gather window [g0, g0 + gl), scatter window [a0, a0 + wl)
    // MIX: the block's far columns sit in the LAST kF1MaxFar slots of the gather list and of the column-ordered pass (slots kFB ..): which lanes
    // serve them does not depend on the record (their loads were requested before it arrived)
    const int nfc = MIX && !vec_only ? fb.w : 0;
    const int nw = nfc ? CW : (gl + kBlock - 1) / kBlock, nu = (cnt + kBlock - 1) / kBlock, ns2 = nfc ? CW : (wl + kBlock - 1) / kBlock;
    KT(2);
    // ---- loads.  First the (P + sigma I) entry of this lane: its column decides whether the operand comes from the window or has
    //      to be recomputed from its parts (columns outside the window), and those loads should leave with the window's, not after it
    // (every request of this stage is unconditional and its destination is written by nothing else: a default value assigned first makes the
    //  compiler guard the register at the loop's head with `s_waitcnt vmcnt(0)` -- which also sits out the next block's stream and the previous
    //  block's stores.  The last budgeted update (vec_only) requests a few values it does not use.)
    const bool hasp = !vec_only && tid < pcnt;
    const int pe = min(pk0 + max(0, min(tid, pcnt - 1)), f.pnnz - 1);
    const double pv = gptr(f.pval)[pe]; const int pc = gptr(f.pcol)[pe];
    // ---- window parts (+ p, x~ where the window column is one of the block's own)
    double wm[CW], wr[CW], wsv[CW], wq[CW][D], wpp[CW], wx[CW];
    bool wown[CW];
    [[maybe_unused]] int wsw[CW]; [[maybe_unused]] F1Spill wst[CW];      // MIX: the column's packed spill word, its first two slots
#pragma unroll
    for (int u = 0; u < CW; u++) {
      wown[u] = false;
      if (u < nw) {
        const int e = tid + u * kBlock;
        int c = g0 + min(e, gl - 1);
        if constexpr (MIX) { const bool isfar = u == CW - 1 && e >= kFB && e - kFB < nfc; if (isfar) c = fcl.x; wsw[u] = isfar ? fcl.y : spk[c]; }
        wown[u] = e < gl && c >= cs0 && c - cs0 < nown;
        wm[u] = Minv[c];
#pragma unroll
        for (int q = 0; q < D; q++) wq[u][q] = repcur[q * ns + c];
        const int co = wown[u] ? c : g0;                    // (other lanes re-read one valid element: no branch around the loads)
        if (!FIRST) {
          wr[u] = rread[c];
          wsv[u] = sprev[c];
          wx[u] = xs_r[co]; wpp[u] = p_r[co];
        } else { wx[u] = xs_r[co]; wpp[u] = gptr(d.xg)[co]; }      // F_0: the own lane also moves x~ on (x~_prev <- x~, x~ <- x_g)
      }
    }
    if constexpr (MIX) {                                     // the spill slots: behind the packed words alone (the first of this lane's requests to return)
#pragma unroll
      for (int u = 0; u < CW; u++) { if (u < nw) wst[u] = f1_spill_take(spcur, wsw[u]); }
    }
    // ---- row / column pointers (the matrix entries themselves arrive in LDS: S, requested one block ahead)
    int cp0[CW], cp1[CW];
    const int prow = r0 + min(tid, nrows - 1);
    const int rp0 = gptr(d.A.rowptr)[prow], rp1 = gptr(d.A.rowptr)[prow + 1]; const double rrho = gptr(d.rho)[prow];
#pragma unroll
    for (int u = 0; u < CW; u++) {
      if (u < ns2) {
        int c = min(tid + u * kBlock, wl - 1);
        if constexpr (MIX) { const int cf = tid + u * kBlock - kFB; if (u == CW - 1 && cf >= 0 && cf < nfc) c = wl + cf; }      // a far segment: behind the window's
        cp0[u] = gptr(f.cptr)[cpo + c]; cp1[u] = gptr(f.cptr)[cpo + c + 1];
      }
    }
    const int pj = min(cs0 + max(0, min(tid, nown - 1)), n - 1);
    const int pp0 = gptr(f.prp)[pj], pp1 = gptr(f.prp)[pj + 1];
    // ---- a (P + sigma I) entry whose column lies outside the window: its operand is recomputed from its parts in the product phase below (requested
    //      THERE: held from here they would cost 7 + 2 D registers across the block's register peak for a case banded problems never meet)
    const int pcl = pc - g0;
    const bool esc = hasp && !(pcl >= 0 && pcl < gl);
    KT(3);
    // ---- u_k on the window -> LDS; the lane of an own column also performs that column's vector update
#pragma unroll
    for (int u = 0; u < CW; u++) {
      if (u < nw) {
        int e = min(tid + u * kBlock, gl - 1);
        if constexpr (MIX) { const int ef = tid + u * kBlock; if (u == CW - 1 && ef >= kFB && ef - kFB < nfc) e = ef; }
        double un;
        double w = f1_w<D>(wq[u]);
        if constexpr (MIX) w += f1_spill_sum(spcur, wsw[u], wst[u]);
        if constexpr (FIRST) {
          const double r0 = w;                              // rhs - K x_g: the slices' sum, the same expression in every workgroup that holds the column
          un = wm[u] * r0;
          if (wown[u]) {
            const int j = g0 + e;
            gst(rnxt + j, r0); gst(gptr(d.xsp) + j, wx[u]); gst(xs_w + j, wpp[u]);
            g_acc += r0 * un; rn_acc = nanmax(rn_acc, fabs(r0));
          }
        } else {
          if (wown[u]) un = own_update(g0 + e, wm[u], wr[u], w, wsv[u], wpp[u], wx[u]);
          else { double sn, rn; f1_upd(sc, wm[u], wr[u], w, wsv[u], sn, rn, un); }
        }
        if (wown[u]) L.uown[g0 + e - cs0] = un;
        L.win[e] = un;                                      // (clamped lanes store the same value)
      }
    }
    // ---- own columns outside the gather window (none on banded problems; all of them in the last budgeted update)
    for (int jj = tid; jj < nown; jj += kBlock) {
      const int j = cs0 + jj;
      if (j >= g0 && j - g0 < gl) continue;
      const double mi = Minv[j];
      double rp[D];
#pragma unroll
      for (int q = 0; q < D; q++) rp[q] = repcur[q * ns + j];
      double un;
      double wj = f1_w<D>(rp);
      if constexpr (MIX) wj += f1_spill_sum(spcur, spk, j);
      if constexpr (FIRST) {
        const double r0 = wj;
        un = mi * r0;
        gst(rnxt + j, r0); gst(gptr(d.xsp) + j, xs_r[j]); gst(xs_w + j, gptr(d.xg)[j]);
        g_acc += r0 * un; rn_acc = nanmax(rn_acc, fabs(r0));
      } else un = own_update(j, mi, rread[j], wj, sprev[j], p_r[j], xs_r[j]);
      if (!vec_only) L.uown[jj] = un;
    }
    if (vec_only) continue;
    KT(4);
    // ---- the block's stream has landed (this wave's pieces: the wait; everybody's: the barrier).  Every load this lane has requested so far is
    //      complete as well -- pinned into registers HERE, so that the compiler places no wait of its own behind the request of the next block's
    //      stream below (it cannot see that transfer: a later `s_waitcnt vmcnt(small)` for one of these values would wait for the stream too)
    f1_stream_wait();
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" :: "v"(rp0), "v"(rp1), "v"(rrho), "v"(pp0), "v"(pp1), "v"(pv), "v"(pc));
#pragma unroll
    for (int u = 0; u < CW; u++) { if (u < ns2) asm volatile("" :: "v"(cp0[u]), "v"(cp1[u])); }
    if constexpr (MIX) asm volatile("" :: "v"(fql));
#endif
    __syncthreads();
    // F_0: ||rhs||_inf of the own columns from the slices of rhs alone (requested now that the window's registers are free, folded at the block's end)
    [[maybe_unused]] double bv[D];
    [[maybe_unused]] int bsw = 0; [[maybe_unused]] F1Spill bst{0.0, 0.0};
    if constexpr (FIRST) {
      const int j = cs0 + max(0, min(tid, nown - 1));
#pragma unroll
      for (int q = 0; q < D; q++) bv[q] = repV[q * ns + j];
      if constexpr (MIX) bsw = spk[j];
    }
    // ---- products: A entries against the window; the (P + sigma I) entry against the window or its recomputed operand
    double vw[CE]; unsigned int en[CE];
#pragma unroll
    for (int u = 0; u < CE; u++) { if (u < nu) { const int e = min(tid + u * kBlock, cnt - 1); vw[u] = S.val[e]; en[u] = S.ent[e]; } }
#pragma unroll
    for (int u = 0; u < CE; u++) { if (u < nu) L.prod[tid + u * kBlock] = vw[u] * L.win[en[u] & 0x1ffu]; }
    if (hasp) {
      double uv;
      if (!esc) uv = L.win[pcl];
      else {
        double eq[D];
#pragma unroll
        for (int q = 0; q < D; q++) eq[q] = repcur[q * ns + pc];
        const double em = Minv[pc];
        double ew = f1_w<D>(eq);
        if constexpr (MIX) ew += f1_spill_sum(spcur, spk, pc);
        if constexpr (FIRST) uv = em * ew;
        else { double sn, rn; f1_upd(sc, em, rread[pc], ew, sprev[pc], sn, rn, uv); }
      }
      L.pprod[tid] = pv * uv;
    }
    KT(5);
    __syncthreads();
    // ---- every wave holds its entries in registers: the stream buffer is free -- the NEXT block's stream goes out now and lands during this
    //      block's remaining LDS phases, its stores, and the next block's record / window loads
    if (sl + slots < per) {
      const int bn = __builtin_amdgcn_readfirstlane(xcd * per + sl + slots);
      if (bn < d.A.nblk) f1_stream_issue<!WT>(f.stream, bn, S);
    }
    // ---- row sums: t = rho .* (A u) -> LDS
    // (first pass peeled: its operands are in registers -- a shared loop body would carry the later passes' load waits, and any `s_waitcnt vmcnt`
    //  executed here waits for the stream requested above as well)
    if (tid < nrows) {
      const double au = f1_segsum<6>(L.prod, rp0 - k0, rp1 - k0), t = rrho * au;
      L.tvec[tid] = t; dl_acc += t * au;
    }
    for (int row = tid + kBlock; row < nrows; row += kBlock) {
      const int q0 = gptr(d.A.rowptr)[r0 + row], q1 = gptr(d.A.rowptr)[r0 + row + 1];
      const double au = f1_segsum<6>(L.prod, q0 - k0, q1 - k0), t = gptr(d.rho)[r0 + row] * au;
      L.tvec[row] = t; dl_acc += t * au;
    }
    KT(6);
    __syncthreads();
    // ---- A_g' t: val * t[row] scattered to column-major order;  pu = (P + sigma I) u on the own columns -> global
#pragma unroll
    for (int u = 0; u < CE; u++) { if (u < nu) L.prod[en[u] >> 18] = vw[u] * L.tvec[(en[u] >> 9) & 0x1ffu]; }     // (clamped lanes repeat the last entry's store)
    if (tid < nown) {
      const double pu = f1_segsum<4>(L.pprod, pp0 - pk0, pp1 - pk0);
      L.puown[tid] = pu; dl_acc += L.uown[tid] * pu;
    }
    for (int jj = tid + kBlock; jj < nown; jj += kBlock) {
      const int q0 = gptr(f.prp)[cs0 + jj], q1 = gptr(f.prp)[cs0 + jj + 1];
      const double pu = f1_segsum<4>(L.pprod, q0 - pk0, q1 - pk0);
      L.puown[jj] = pu; dl_acc += L.uown[jj] * pu;
    }
    KT(7);
    __syncthreads();
    // ---- one lane per column of the scatter window
    double *rout = repnxt + (size_t)(b % D) * ns;
#pragma unroll
    for (int u = 0; u < CW; u++) {
      if (u < ns2) {
        const int c = tid + u * kBlock;
        if (c < wl) {
          double v = f1_segsum<8>(L.prod, cp0[u], cp1[u]);
          const int jo = a0 + c - cs0;
          if (jo >= 0 && jo < nown) v += L.puown[jo];         // this block owns the column: + (P + sigma I) u
          gst(rout + a0 + c, v);
        } else if constexpr (MIX) {
          if (u == CW - 1 && c >= kFB && c - kFB < nfc) gst(spnxt + fql, f1_segsum<8>(L.prod, cp0[u], cp1[u]));      // a far column's sum: the block's spill slot for it
        }
      }
    }
    for (int j = cov0 + tid; j < a0; j += kBlock) gst(rout + j, 0.0);             // the replica's gap up to the next window of this replica
    for (int j = a0 + wl + tid; j < cov1; j += kBlock) gst(rout + j, 0.0);
    if constexpr (FIRST) {
      if (tid < nown) bn_acc = nanmax(bn_acc, fabs(MIX ? f1_w<D>(bv) + f1_spill_sum(spV, bsw, f1_spill_take(spV, bsw)) : f1_w<D>(bv)));
      for (int jj = tid + kBlock; jj < nown; jj += kBlock) {
        double rv[D];
#pragma unroll
        for (int q = 0; q < D; q++) rv[q] = repV[q * ns + cs0 + jj];
        bn_acc = nanmax(bn_acc, fabs(MIX ? f1_w<D>(rv) + f1_spill_sum(spV, spk, cs0 + jj) : f1_w<D>(rv)));
      }
    }
    KT(8);
    if (sl + slots < per) __syncthreads();                  // (another block follows: the LDS arrays are reused)
  }
  if (vec_only) f1_stream_wait();                           // (a launch that applies no operator never consumed the stream its head requested: nothing may be in flight when the
                                                            //  workgroup's LDS is released.  Not otherwise: the wait would also sit out every store of the body)
  __syncthreads();
  block_sum_max_sum(g_acc, rn_acc, dl_acc, L.red);
  put_partial(gptr(d.part), SL_GAMMA0 + par, g_acc); put_partial(gptr(d.part), SL_RN0 + par, rn_acc);
  if constexpr (FIRST) {                                    // F_0 also owns ||rhs||_inf (the PCG tolerance of this ADMM iteration: f1_fold_finish, k = 1; KA of a capped PCG)
    bn_acc = block_max(bn_acc, L.red);
    put_partial(gptr(d.part), SL_BN, bn_acc);
  }
  if (!vec_only) put_partial(gptr(d.part), SL_DELTA + par, dl_acc);
  KT(9);
}
// returns false when the PCG had already converged (nothing done: the caller runs KA in this launch)
// the record of the workgroup's first row block
__device__ __forceinline__ F1Rec f1_first_record(const int *blk, int nblk) {
  const int b0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x & 7) * ((nblk + 7) >> 3) + (int)(blockIdx.x >> 3));
  const size_t b = (size_t)min(b0, nblk - 1);
  return F1Rec{sload_int4(blk, 4 * b), sload_int4(blk, 4 * b + 1), sload_int4(blk, 4 * b + 2), sload_int4(blk, 4 * b + 3)};
}
// D (the number of replica vectors, DevF1::D) is a TEMPLATE parameter of the kernels: a slot kernel that carries the bodies of all four
// values pays for the three it never runs in every launch (the head of a launch is as long as the kernel's register / code footprint
// makes it)
template <int D, bool MIX, bool WT>
__device__ __forceinline__ bool f1_iteration(const Dev &d, const int k, const int cap, const int admm_par, const int probe, F1Lds &L, F1Stream &S, const F1Rec &rec0, const F1Fold &fold, const int par) {
  KT(0);
  F1Scal sc;
  if (!f1_fold_finish(d, k, admm_par, probe, fold, L.red, sc)) return false;
  const bool vec_only = !probe && k >= cap;                 // the last budgeted update: no operator apply follows
  if (k == 0) f1_body<D, true, MIX, WT>(d, k, vec_only, sc, L, S, rec0, par);
  else f1_body<D, false, MIX, WT>(d, k, vec_only, sc, L, S, rec0, par);
  return true;
}
// the stream of the workgroup's FIRST row block: its address depends on nothing but the block index, so it leaves at the head of the launch,
// before the phase record, the block record or the partials have arrived
template <bool NT>
__device__ __forceinline__ void f1_stream_first(const unsigned char *stream, int nblk, F1Stream &S) {
  const int per = (nblk + 7) >> 3, slot0 = (int)(blockIdx.x >> 3);
  const int b0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x & 7) * per + slot0);
  if (slot0 < per && b0 < nblk) f1_stream_issue<NT>(stream, b0, S);
}

// ---------------------------------------------------------------------------------------------- KA in the F1 form (no KB launch)
// One launch does what KA and the following KB did in two ("k + 2 launches"):
//   rows of block g     z~ = A_g x~ (window of x~ staged in LDS, products through the block's stream, one lane per row), the z / y update
//                       (_osqp.py:682-703), v = rho z - y, z~, A x_g = z~ + theta (z~ - z~_prev), t0 = rho A x_g           -- what KA did
//   own columns         x = alpha x~ + (1 - alpha) x, dx (_osqp.py:660-668), the next PCG start x_g = x~ + theta (x~ - x~_prev)  -- what KA did
//   transposed passes   A_g' v and A_g' t0 per column of the block's scatter window (the same column-ordered pass as F's A_g' t, twice),
//                       + sigma x - q resp. (P + sigma I) x_g on the own columns, written as the block's slices of two replica sets:
//                         sum_d repR_d = r_0 = rhs - K x_g ,   sum_d repV_d = rhs = sigma x - q + A' v     (K x_g = (P + sigma I) x_g + A' t0)   -- what KB did
// F_0 then forms r_0 column by column from the first set (f1_body<D, true>) and ||rhs|| on its own columns from the second, with gamma_0, ||r_0||.
// SCATTER_ONLY: the first launch of a chunk -- v, t0, x, x_g are in memory (a rho update, a warm start or the previous chunk left them): only
// the transposed passes and the own-column terms run.  x~_prev is NOT advanced here (F_0's own lanes do that): this launch reads it for the
// operands of P at columns other workgroups own.
template <int D, bool SCATTER_ONLY, bool MIX, bool WT>
__device__ __forceinline__ void f1_ka_body(const Dev &d, F1Lds &L, F1Stream &S, const F1Rec &rec0, const double theta) {
  const DevF1 &f = d.f1;
  const int tid = threadIdx.x;
  const double *va = gptr(f.va); const size_t ns = f.ns;
  const double *xs_r = va + ns;
  double *repR = gptr(f.va) + (7 + (size_t)1 * D) * ns, *repV = gptr(f.va) + (7 + (size_t)2 * D) * ns;      // the parity-1 set (what F_0 reads as K u_{-1}'s place), the third set
  [[maybe_unused]] double *spR = MIX ? gptr(f.spill) + (f.nsp + 2) : nullptr, *spVw = MIX ? gptr(f.spill) + 2 * (f.nsp + 2) : nullptr;      // per-block mixing: the spill sets that go with them
  const double alpha = d.alpha, sigma = d.sigma;
  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int per = (d.A.nblk + 7) >> 3;
  constexpr int CW = kF1Win / kBlock, CE = kF1Chunk / kBlock;
  constexpr int kFB = kF1Win - kF1MaxFar;
  for (int sl = slot0; sl < per; sl += slots) {
    const int b = __builtin_amdgcn_readfirstlane(xcd * per + sl);
    if (b >= d.A.nblk) break;
    [[maybe_unused]] int fcl = 0, fql = 0;                   // MIX: the far slot of this lane (f1_body): column, spill slot
    if constexpr (MIX) { if (tid >= kFB - (CW - 1) * kBlock) { const size_t i = (size_t)b * kF1MaxFar + (tid - (kFB - (CW - 1) * kBlock)); fcl = gptr(f.fcol)[2 * i]; fql = gptr(f.fq)[i]; } }
    const F1Rec rec = sl == slot0 ? rec0 : f1_record(f, b);
    const int4 ds = rec.ds, fa = rec.fa, fb = rec.fb, fc = rec.fc;
    const int r0 = ds.x, nrows = ds.y - ds.x, k0 = ds.z, cnt = ds.w - ds.z;
    const int cov0 = fa.x, cov1 = fa.y, cs0 = fa.z, nown = fa.w - fa.z;
    const int cpo = fb.x, pk0 = fb.y, pcnt = fb.z - fb.y;
    const int g0 = fc.x, gl = fc.y, a0 = fc.z, wl = fc.w;
    const int nfc = MIX ? fb.w : 0;                          // MIX: far columns in the last kF1MaxFar slots (f1_body)
    const int nw = nfc ? CW : (gl + kBlock - 1) / kBlock, nu = (cnt + kBlock - 1) / kBlock, ns2 = nfc ? CW : (wl + kBlock - 1) / kBlock;
    // ---- loads: the (P + sigma I) entry of this lane and its operand x_g[column]
    double pv = 0.0; int pc = g0;
    const bool hasp = tid < pcnt;
    { const int e = pk0 + max(0, min(tid, pcnt - 1)); pv = gptr(f.pval)[e]; pc = gptr(f.pcol)[e]; }
    const int pcl = pc - g0;
    const bool pin = pcl >= 0 && pcl < gl;
    double pxs = 0.0, pxp = 0.0;                             // x~[pc] (from the window when it is inside), x~_prev[pc]  -- or x_g[pc] itself (SCATTER_ONLY)
    if (SCATTER_ONLY) pxs = gptr(d.xg)[pc];
    else { pxp = gptr(d.xsp)[pc]; if (!pin) pxs = xs_r[pc]; }
    // ---- window of x~ (rows' products) and the own columns' operands
    double wx[CW], wxo[CW], wxp[CW], wq[CW];
    bool wown[CW];
#pragma unroll
    for (int u = 0; u < CW; u++) {
      wown[u] = false;
      if (u < nw) {
        const int e = tid + u * kBlock;
        int c = g0 + min(e, gl - 1);
        if constexpr (MIX) { if (u == CW - 1 && e >= kFB && e - kFB < nfc) c = fcl; }
        wown[u] = e < gl && c >= cs0 && c - cs0 < nown;
        const int co = wown[u] ? c : g0;
        if (!SCATTER_ONLY) { wx[u] = xs_r[c]; wxp[u] = gptr(d.xsp)[co]; }
        wxo[u] = gptr(d.x)[co]; wq[u] = gptr(d.q)[co];
      }
    }
    // ---- rows
    int rp0 = 0, rp1 = 0;
    double el = 0, eu = 0, erho = 0, erinv = 0, ez = 0, ey = 0, ezt = 0;      // (SCATTER_ONLY: ez = v, ey = t0 of the row)
    { const int row = r0 + min(tid, nrows - 1);
      if (SCATTER_ONLY) { ez = gptr(d.v)[row]; ey = gptr(d.t0)[row]; }
      else { rp0 = gptr(d.A.rowptr)[row]; rp1 = gptr(d.A.rowptr)[row + 1]; el = gptr(d.l)[row]; eu = gptr(d.u)[row]; erho = gptr(d.rho)[row]; erinv = gptr(d.rho_inv)[row]; ez = gptr(d.z)[row]; ey = gptr(d.y)[row]; ezt = gptr(d.zt)[row]; } }
    int cp0[CW], cp1[CW];
#pragma unroll
    for (int u = 0; u < CW; u++) {
      if (u < ns2) {
        int c = min(tid + u * kBlock, wl - 1);
        if constexpr (MIX) { const int cf = tid + u * kBlock - kFB; if (u == CW - 1 && cf >= 0 && cf < nfc) c = wl + cf; }
        cp0[u] = gptr(f.cptr)[cpo + c]; cp1[u] = gptr(f.cptr)[cpo + c + 1];
      }
    }
    int pp0 = 0, pp1 = 0;
    { const int j = min(cs0 + max(0, min(tid, nown - 1)), d.n - 1); pp0 = gptr(f.prp)[j]; pp1 = gptr(f.prp)[j + 1]; }
    // ---- window -> LDS; the own lane updates x and leaves  sigma x - q  for its column
#pragma unroll
    for (int u = 0; u < CW; u++) {
      if (u < nw) {
        int e = min(tid + u * kBlock, gl - 1);
        if constexpr (MIX) { const int ef = tid + u * kBlock; if (u == CW - 1 && ef >= kFB && ef - kFB < nfc) e = ef; }
        if (!SCATTER_ONLY) L.win[e] = wx[u];
        if (wown[u]) {
          const int j = g0 + e;
          double xn = wxo[u];
          if (!SCATTER_ONLY) {
            const double xt = wx[u];
            xn = alpha * xt + (1.0 - alpha) * wxo[u];                        // _osqp.py:664-668
            gst(gptr(d.dx) + j, xn - wxo[u]); gst(gptr(d.x) + j, xn);
            gst(gptr(d.xg) + j, xt + theta * (xt - wxp[u]));                            // next PCG start (Dev::xg); x~_prev moves on in F_0
          }
          L.uown[j - cs0] = sigma * xn - wq[u];
        }
      }
    }
    f1_stream_wait();
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" :: "v"(rp0), "v"(rp1), "v"(el), "v"(eu), "v"(erho), "v"(erinv), "v"(ez), "v"(ey), "v"(ezt), "v"(pp0), "v"(pp1), "v"(pv), "v"(pc), "v"(pxs), "v"(pxp));
#pragma unroll
    for (int u = 0; u < CW; u++) { if (u < ns2) asm volatile("" :: "v"(cp0[u]), "v"(cp1[u])); }
    if constexpr (MIX) asm volatile("" :: "v"(fql));
#endif
    __syncthreads();
    double vw[CE]; unsigned int en[CE];
#pragma unroll
    for (int u = 0; u < CE; u++) { if (u < nu) { const int e = min(tid + u * kBlock, cnt - 1); vw[u] = S.val[e]; en[u] = S.ent[e]; } }
    if (!SCATTER_ONLY) {
#pragma unroll
      for (int u = 0; u < CE; u++) { if (u < nu) L.prod[tid + u * kBlock] = vw[u] * L.win[en[u] & 0x1ffu]; }
    }
    if (hasp) {
      double xg;
      if (SCATTER_ONLY) xg = pxs;
      else { const double xt = pin ? L.win[pcl] : pxs; xg = xt + theta * (xt - pxp); }
      L.pprod[tid] = pv * xg;
    }
    __syncthreads();
    if (sl + slots < per) {                                 // (every wave holds its entries in registers: the next block's stream goes out)
      const int bn = __builtin_amdgcn_readfirstlane(xcd * per + sl + slots);
      if (bn < d.A.nblk) f1_stream_issue<!WT>(f.stream, bn, S);
    }
    // ---- rows: z~, the z / y update, v and t0 of the row (v -> LDS now, t0 kept for the second pass)
    double t0r = 0.0;
    if (tid < nrows) {
      double vi;
      if (SCATTER_ONLY) { vi = ez; t0r = ey; }
      else {
        const int i = r0 + tid;
        const double ztil = f1_segsum<6>(L.prod, rp0 - k0, rp1 - k0);
        const double zr = alpha * ztil + (1.0 - alpha) * ez;                 // _osqp.py:686-690
        const double zn = fmin(fmax(zr + erinv * ey, el), eu);               // :674
        const double dyi = erho * (zr - zn), yn = ey + dyi;                  // :698-703
        const double zg = ztil + theta * (ztil - ezt);                       // A x_g
        vi = erho * zn - yn; t0r = erho * zg;
        gst(gptr(d.y) + i, yn); gst(gptr(d.dy) + i, dyi); gst(gptr(d.z) + i, zn); gst(gptr(d.zt) + i, ztil); gst(gptr(d.v) + i, vi); gst(gptr(d.ztg) + i, zg); gst(gptr(d.t0) + i, t0r);
      }
      L.tvec[tid] = vi;
    }
    for (int row = tid + kBlock; row < nrows; row += kBlock) {              // (blocks of more than kBlock rows; t0 of these rows is re-read from memory below)
      const int i = r0 + row;
      double vi;
      if (SCATTER_ONLY) vi = gptr(d.v)[i];
      else {
        const int q0 = gptr(d.A.rowptr)[i], q1 = gptr(d.A.rowptr)[i + 1];
        const double rho = gptr(d.rho)[i], zo = gptr(d.z)[i], yo = gptr(d.y)[i];
        const double ztil = f1_segsum<6>(L.prod, q0 - k0, q1 - k0);
        const double zr = alpha * ztil + (1.0 - alpha) * zo;
        const double zn = fmin(fmax(zr + gptr(d.rho_inv)[i] * yo, gptr(d.l)[i]), gptr(d.u)[i]);
        const double dyi = rho * (zr - zn), yn = yo + dyi;
        const double zg = ztil + theta * (ztil - gptr(d.zt)[i]);
        vi = rho * zn - yn;
        gst(gptr(d.y) + i, yn); gst(gptr(d.dy) + i, dyi); gst(gptr(d.z) + i, zn); gst(gptr(d.zt) + i, ztil); gst(gptr(d.v) + i, vi); gst(gptr(d.ztg) + i, zg); gst(gptr(d.t0) + i, rho * zg);
      }
      L.tvec[row] = vi;
    }
    // ---- (P + sigma I) x_g on the own columns
    if (tid < nown) L.puown[tid] = f1_segsum<4>(L.pprod, pp0 - pk0, pp1 - pk0);
    for (int jj = tid + kBlock; jj < nown; jj += kBlock) { const int q0 = gptr(f.prp)[cs0 + jj], q1 = gptr(f.prp)[cs0 + jj + 1]; L.puown[jj] = f1_segsum<4>(L.pprod, q0 - pk0, q1 - pk0); }
    __syncthreads();
    // ---- first transposed pass: A_g' v
#pragma unroll
    for (int u = 0; u < CE; u++) { if (u < nu) L.prod[en[u] >> 18] = vw[u] * L.tvec[(en[u] >> 9) & 0x1ffu]; }
    __syncthreads();
    double cv[CW];
#pragma unroll
    for (int u = 0; u < CW; u++) {
      cv[u] = 0.0;
      if (u < ns2) { const int c = tid + u * kBlock; if (c < wl || (MIX && u == CW - 1 && c >= kFB && c - kFB < nfc)) cv[u] = f1_segsum<8>(L.prod, cp0[u], cp1[u]); }
    }
    __syncthreads();
    // ---- second pass: A_g' t0
    if (tid < nrows) L.tvec[tid] = t0r;
    for (int row = tid + kBlock; row < nrows; row += kBlock) L.tvec[row] = gptr(d.t0)[r0 + row];      // (written by this thread above, or by an earlier launch)
    __syncthreads();
#pragma unroll
    for (int u = 0; u < CE; u++) { if (u < nu) L.prod[en[u] >> 18] = vw[u] * L.tvec[(en[u] >> 9) & 0x1ffu]; }
    __syncthreads();
    double *routR = repR + (size_t)(b % D) * ns, *routV = repV + (size_t)(b % D) * ns;
#pragma unroll
    for (int u = 0; u < CW; u++) {
      if (u < ns2) {
        const int c = tid + u * kBlock;
        if (c < wl) {
          double tv = cv[u], tt = f1_segsum<8>(L.prod, cp0[u], cp1[u]);
          const int jo = a0 + c - cs0;
          if (jo >= 0 && jo < nown) { tv += L.uown[jo]; tt += L.puown[jo]; }      // this block owns the column: + sigma x - q  resp.  + (P + sigma I) x_g
          gst(routV + a0 + c, tv); gst(routR + a0 + c, tv - tt);                             // slices of rhs, and of r_0 = rhs - K x_g
        } else if constexpr (MIX) {
          if (u == CW - 1 && c >= kFB && c - kFB < nfc) { const double tv = cv[u], tt = f1_segsum<8>(L.prod, cp0[u], cp1[u]); gst(spVw + fql, tv); gst(spR + fql, tv - tt); }      // a far column: the block's spill slots
        }
      }
    }
    for (int j = cov0 + tid; j < a0; j += kBlock) { gst(routV + j, 0.0); gst(routR + j, 0.0); }      // the replicas' gaps up to the next window of this replica
    for (int j = a0 + wl + tid; j < cov1; j += kBlock) { gst(routV + j, 0.0); gst(routR + j, 0.0); }
    if (sl + slots < per) __syncthreads();
  }
}
// KA of the slot machine in the F1 form: extrapolation weight, the body above, PCG statistics of the ADMM iteration that ends (as slot_ka)
template <int D, bool MIX, bool WT>
__device__ __forceinline__ void f1_slot_ka(const Dev &d, F1Lds &L, F1Stream &S, const F1Rec &rec0, int used, int conv, int admm, int target, int rn_slot, int seq) {
  double theta = d.theta, rn_last = 0.0, bn_last = 0.0;
  if (!conv) { theta = cutoff_theta(d, rn_slot, L.red, rn_last, bn_last, admm); __syncthreads(); }      // (conv comes from the slot record: uniform)
  f1_ka_body<D, false, MIX, WT>(d, L, S, rec0, theta);
  if (blockIdx.x == 0) {
    if (!conv) {
      const double rn = rn_last, bn = bn_last;
      conv = !(rn > fmax(gptr(d.scal)[S_TOL_REL] * bn, gptr(d.scal)[S_TOL_ABS])) ? 2 : 0;
      if (!conv && threadIdx.x == 0 && rn > 0.1 * gptr(d.scal)[S_RN0]) gptr(d.flags)[F_STAT_STAG] += 1;
    }
    if (threadIdx.x == 0) {
      gptr(d.flags)[F_STAT_SUM] += used; gptr(d.flags)[F_STAT_SUMSQ] += used * used; gptr(d.flags)[F_STAT_N] += 1;
      if (used > gptr(d.flags)[F_STAT_MAX]) gptr(d.flags)[F_STAT_MAX] = used;
      if (!conv) gptr(d.flags)[F_STAT_UNCONV] += 1;
      if (d.ctl && admm + 1 >= target) { gptr(d.ctl)->seq_end = seq + 1; gptr(d.ctl)->chunk_done = 1; }      // device-driven boundaries: the chunk's last ADMM iteration (read by LATER launches)
    }
  }
}
__global__ __launch_bounds__(kBlock) void k_f1_refresh(Dev d) {
  for (int k = blockIdx.x * kBlock + threadIdx.x; k < d.f1.pnnz; k += gridDim.x * kBlock) d.f1.pval[k] = d.B.val[d.f1.psrc[k]];
  // the value half of every block's stream <- A.val (entry e of block b sits at A.val[first entry of b + e])
  for (int b = blockIdx.x; b < d.A.nblk; b += gridDim.x) {
    const int k0 = d.f1.blk[16 * (size_t)b + 2], cnt = d.f1.blk[16 * (size_t)b + 3] - k0;
    double *dst = reinterpret_cast<double *>(d.f1.stream + (size_t)b * kF1StreamBytes);
    for (int e = threadIdx.x; e < cnt; e += kBlock) dst[e] = d.A.val[k0 + e];
  }
}
// timing probe: one F launch as a solve runs it -- the scalar fold of the previous launch's partials included -- with fixed alpha, beta
// and no stopping test (mode 2; mode 1 skips the fold: what the launch costs without it)
template <int D, bool MIX, bool WT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_f1_probe(Dev d, int k, int mode) {
  __shared__ F1Lds lds;
  __shared__ F1Stream sbuf;
  const int par = k & 1;
  f1_stream_first<!WT>(d.f1.stream, d.A.nblk, sbuf);
  const F1Fold fold = f1_fold_issue(gptr(d.part), par, mode);
  f1_iteration<D, MIX, WT>(d, k, 1 << 30, 0, mode, lds, sbuf, f1_first_record(d.f1.blk, d.A.nblk), fold, par);      // (the first block's record goes out ahead of the partials of the scalar fold: both latencies overlap)
}

// timing probes of the KA body above (time_kernel 17 / 18)
template <int D, bool MIX, bool WT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_f1_ka_probe(Dev d, int scatter_only) {
  __shared__ F1Lds lds;
  __shared__ F1Stream sbuf;
  f1_stream_first<!WT>(d.f1.stream, d.A.nblk, sbuf);
  const F1Rec rec0 = f1_first_record(d.f1.blk, d.A.nblk);
  if (scatter_only) f1_ka_body<D, true, MIX, WT>(d, lds, sbuf, rec0, d.theta); else f1_ka_body<D, false, MIX, WT>(d, lds, sbuf, rec0, d.theta);
}

// The slot kernel of the F1 form: every launch of a chunk's string is this kernel (par: which of the two phase records it reads).  Every phase is
// built on the row blocks of A and their stream (no pass over B = [P + sigma I | A'] exists in this form):
//   P_KB   the first launch of a chunk: the slices of rhs and K x_g from the vectors in memory (f1_ka_body, SCATTER_ONLY)
//   P_F    PCG launch F_k; F_0 forms r_0 from the slices; the launch whose fold finds the PCG converged runs KA right there
//   P_KA   KA after a PCG that stopped at its cap
// An ADMM iteration with k PCG iterations is F_0 .. F_k + the launch that detects convergence and runs KA: k + 2 launches.
// Dev reaches this kernel through MEMORY, not as a by-value argument: the compiler lowers a by-value struct into loads of every used field at
// the kernel's entry -- some sixty here -- and, short of scalar registers, spills each batch to vector lanes before it fetches the next: ten
// dependent load-wait-spill rounds in front of the first request of every launch (measured: +2 us per launch against the probe kernel that
// holds one phase).  From memory the fields are fetched where the phase that runs needs them; the head's own six values sit in front of the
// copy as one 48-byte block.  be::dev_publish keeps the copy current (slot_begin / ctl_begin: once per chunk, never inside a capture).
struct F1Head { const unsigned char *stream; const int *blk; const double *part; const int *slot; int nblk, pad; };
struct F1DevBlock { F1Head h; Dev d; };
// (Per-block mixing, tried: the far lanes of the workgroup's first row block touching their lines at the head of the launch -- both parities, as LDS-direct
//  loads into a scratch KB, so that the body's requests find them in this XCD's L2 instead of waiting for memory: the solve got SLOWER, 80 -> 95 ms on
//  `bench.py --config mixed` (tools/mix_ab.py); the extra requests ahead of the fold's partials cost more than the late far lines do.)
template <int D, bool MIX, bool WT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_slot1(const F1DevBlock *__restrict__ blk, int par) {
  __shared__ F1Lds lds;
  __shared__ F1Stream sbuf;
  static_assert(sizeof(F1Lds) + sizeof(F1Stream) <= 160 * 1024 / 4, "four workgroups per CU");
  const F1Head h = blk->h;
  const Dev &d = blk->d;
  const int *R = h.slot + (par ? SR_WORDS : 0);
  int *W = gptr(d.slot) + (par ? 0 : SR_WORDS);
  // every phase but the idle one consumes the first block's stream: requested before anything else has arrived
  f1_stream_first<!WT>(h.stream, h.nblk, sbuf);
  const F1Fold fold = f1_fold_issue(gptr(h.part), par, 0);    // the previous launch's partials: their address depends on `par` alone
  const F1Rec rec0 = f1_first_record(h.blk, h.nblk);
  SlotState st = slot_read_scalar(R);
  if (st.ph == P_KB) {
    if (st.admm >= st.target) { st.ph = P_IDLE; f1_stream_wait(); slot_write(W, st); return; }
    f1_ka_body<D, true, MIX, WT>(d, lds, sbuf, rec0, 0.0);
    if (blockIdx.x == 0 && threadIdx.x == 0) { gptr(d.flags)[F_DONE] = 0; gptr(d.flags)[F_ITERS] = 0; }
    st.ph = P_F; st.k = 0;
  } else if (st.ph == P_F) {
    if (f1_iteration<D, MIX, WT>(d, st.k, st.cap, st.admm & 1, 0, lds, sbuf, rec0, fold, par)) {
      if (st.k >= st.cap) { st.ph = P_KA; st.used = st.k; st.conv = 0; }      // stopped at the cap: the next launch runs KA
      else st.k += 1;
    } else {                                             // converged after k - 1 iterations (k = 1: the start met the tolerance): KA right here
      __syncthreads();
      f1_slot_ka<D, MIX, WT>(d, lds, sbuf, rec0, st.k - 1, 1, st.admm, st.target, 0, st.seq);
      st.admm += 1; st.k = 0; st.ph = st.admm >= st.target ? P_IDLE : P_F;
    }
  } else if (st.ph == P_KA) {
    f1_slot_ka<D, MIX, WT>(d, lds, sbuf, rec0, st.used, st.conv, st.admm, st.target, par ^ 1, st.seq);
    st.admm += 1; st.k = 0; st.ph = st.admm >= st.target ? P_IDLE : P_F;
  } else f1_stream_wait();                               // (idle: nothing may be in flight when the workgroup's LDS is released)
  slot_write(W, st);
}


#undef gst
// ---------------------------------------------------------------------------------------------- one launch per PCG iteration on the explicit K (K form)
// backend.h DevKf.  Launch F_k of the PCG of one ADMM iteration (k = 0 .. iterations), exactly the F1 form's recurrences (above) with the operator
// applied as ONE CSR product over K's row blocks (process_rows: any sparsity, no windows, no replicas):
//   scalars   the previous launch's partials (f1_fold_issue / f1_fold_finish: stopping test, alpha_{k-1}, beta_{k-1}; k = 0: alpha = beta = 0)
//   gather    u_k[c] = Minv (r_{k-1} - alpha (w_{k-1} + beta s_{k-2}))[c]  from column c's 32-byte record of parity (k + 1) & 1 -- two 16-byte loads
//             of one aligned 32-byte line per entry, requested as the column indices arrive (GKf::fetch), evaluated behind the fold (GKf::prod)
//   own rows  the same update from the row's own record (f1_upd: the owner's and every gatherer's u_k[j] are bit-identical), p_{k-1}, x~ += alpha p_{k-1},
//             w_k[j] = (K u_k)_j  = the row sum, the record of parity k & 1 = {Minv, r_k, w_k, s_{k-1}}, partials gamma_k, ||r_k||, delta_k = <u_k, w_k>
// KB (one pass over B = [P + sigma I | A'], as the two-kernel form's) leaves  {Minv, r_0, 0, 0}  in parity 1 and ||rhs|| in its partial slot, so F_0 is
// the general launch with alpha = beta = 0.  An ADMM iteration with k PCG iterations = KB, F_0 .. F_k, [detect + KA] = k + 3 launches.
struct KfOps { double2 a, b; };                      // a column's record: {Minv, r}, {w, s}
struct GKf {
  const double *rec; const F1Scal *sc;
  using Ops = KfOps;
  __device__ __forceinline__ Ops fetch(int c) const { const double2 *p = reinterpret_cast<const double2 *>(rec + 4 * (size_t)c); Ops o; o.a = p[0]; o.b = p[1]; return o; }
  __device__ __forceinline__ void prod(const Ops &o, double a, double (&pr)[1]) const { double sn, rn, un; f1_upd(*sc, o.a.x, o.a.y, o.b.x, o.b.y, sn, rn, un); pr[0] = a * un; }
  __device__ __forceinline__ void operator()(int c, double a, double (&pr)[1]) const { prod(fetch(c), a, pr); }
};
struct EKf {
  const double *recr; double *recw, *p, *xs; const F1Scal *sc;
  double g = 0, rn = 0, dl = 0;
  double2 oa = make_double2(0, 0), ob = make_double2(0, 0); double pp = 0, px = 0;
  __device__ __forceinline__ void prefetch(int j) { const double2 *q = reinterpret_cast<const double2 *>(recr + 4 * (size_t)j); oa = q[0]; ob = q[1]; pp = p[j]; px = xs[j]; }
  // w: (K u_k)_j; with_w = false: the last budgeted update (no operator apply follows: w is not needed by anybody)
  __device__ __forceinline__ void update(int j, double w, bool with_w) {
    double sn, rnew, un;
    f1_upd(*sc, oa.x, oa.y, ob.x, ob.y, sn, rnew, un);
    const double pn = fma(sc->beta, sc->general ? pp : 0.0, oa.x * oa.y);
    xs[j] = fma(sc->alpha, pn, px); p[j] = pn;
    double2 *o = reinterpret_cast<double2 *>(recw + 4 * (size_t)j);
    o[0] = make_double2(oa.x, rnew); o[1] = make_double2(w, sn);
    g += rnew * un; rn = nanmax(rn, fabs(rnew)); if (with_w) dl += un * w;
  }
  __device__ __forceinline__ void operator()(int j, const double (&s)[1]) { update(j, s[0], true); }
};
struct PreKf {
  [[maybe_unused]] static constexpr int kTraceBase = 0;
  const Dev &d; int k, admm_par, probe, par; double *red; F1Scal *sc;
  using Tok = F1Fold;
  __device__ __forceinline__ Tok begin() const { return f1_fold_issue(d.part, par, probe); }
  __device__ __forceinline__ bool finish(const Tok &t) const { return f1_fold_finish(d, k, admm_par, probe, t, red, *sc); }
};
// returns false when the PCG had already converged (nothing done: the caller runs KA in this launch)
template <class L>
__device__ __forceinline__ bool kf_iteration(const Dev &d, const int k, const int cap, const int admm_par, const int probe, L &lds, const int par) {
  F1Scal sc{0.0, 0.0, 0};
  const int cur = (k + 1) & 1, nxt = k & 1;
  const double *recr = d.kf.rec + (size_t)cur * 4 * (size_t)d.n;
  double *recw = d.kf.rec + (size_t)nxt * 4 * (size_t)d.n;
  const bool vec_only = !probe && k >= cap;                 // the last budgeted update: no operator apply follows
  EKf e{recr, recw, d.p, d.xs, &sc};
  if (vec_only) {
    const F1Fold fold = f1_fold_issue(d.part, par, probe);
    if (!f1_fold_finish(d, k, admm_par, probe, fold, lds.red, sc)) return false;
    // the rows of this workgroup's row blocks of K (the same ownership as the operator launches)
    const DevCsr &M = d.kf.K;
    const int4 *desc = reinterpret_cast<const int4 *>(M.blkdesc);
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, slots = gridDim.x >> 3, per = (M.nblk + 7) >> 3;
    for (int sl = slot0; sl < per; sl += slots) {
      const int b = xcd * per + sl;
      if (b >= M.nblk) break;
      const int4 ds = desc[b];
      const int r1 = ds.y < 0 ? ds.x + 1 : ds.y;
      for (int j = ds.x + (int)threadIdx.x; j < r1; j += kBlock) { e.prefetch(j); e.update(j, 0.0, false); }
    }
  } else {
    GKf g{recr, &sc};
    if (!process_rows<1>(d.kf.K, g, e, lds, PreKf{d, k, admm_par, probe, par, lds.red, &sc})) return false;
  }
  __syncthreads();
  block_sum_max_sum(e.g, e.rn, e.dl, lds.red);
  put_partial(d.part, SL_GAMMA0 + par, e.g); put_partial(d.part, SL_RN0 + par, e.rn);
  if (!vec_only) put_partial(d.part, SL_DELTA + par, e.dl);
  return true;
}
struct EKbK {       // KB of the K form: rhs and r_0 as EKb; the start record {Minv, r_0, 0, 0} of parity 1 instead of r and u_0
  const double *x, *q, *Minv; double *rec1; double sigma; const double *xg; double *xs; double bn = 0; double px = 0, pq = 0, pm = 0, pg = 0;
  __device__ __forceinline__ void prefetch(int j) { px = x[j]; pq = q[j]; pm = Minv[j]; pg = xg[j]; }
  __device__ __forceinline__ void operator()(int j, const double (&s)[2]) {
    const double rhs = sigma * px - pq + s[0];
    const double rr = rhs - s[1];
    double2 *o = reinterpret_cast<double2 *>(rec1 + 4 * (size_t)j);
    o[0] = make_double2(pm, rr); o[1] = make_double2(0.0, 0.0);
    xs[j] = pg;                                              // x~ restarts from the extrapolated point (nobody gathers xs in this kernel)
    bn = nanmax(bn, fabs(rhs));
  }
};
// K.val <- the term lists (backend.h DevKf): copies of B.val entries (P + sigma I) and products rho_i A_ia A_ib, summed in list order
__global__ __launch_bounds__(kBlock) void k_kf_values(Dev d, int cond) {
  if (cond && !d.ctl->rho_flag) return;
  const DevKf &f = d.kf;
  const int stride = gridDim.x * kBlock;
  for (int e = blockIdx.x * kBlock + threadIdx.x; e < f.K.nnz; e += stride) {
    const int t0 = f.tptr[e], t1 = f.tptr[e + 1];
    double v = 0.0;
    for (int t = t0; t < t1; t++) { const int i = f.trow[t]; v += i < 0 ? d.B.val[f.ta[t]] : d.rho[i] * (d.A.val[f.ta[t]] * d.A.val[f.tb[t]]); }
    f.K.val[e] = v;
  }
}
// The slot kernel of the K form: every launch of a chunk's string is this kernel (par: which of the two phase records it reads).
//   P_KB   rhs, r_0 and the start record from one pass over B (what the two-kernel form's KB does)
//   P_F    PCG launch F_k on the explicit K; the launch whose fold finds the PCG converged runs KA right there
//   P_KA   KA after a PCG that stopped at its cap
__global__ __launch_bounds__(kBlock) void k_slotk(Dev d, int par) {
  __shared__ union { StreamLds<2> kb; StreamLds<1> f; StreamLdsW<1, double> ka; } lds;
  const int *R = d.slot + (par ? SR_WORDS : 0);
  int *W = d.slot + (par ? 0 : SR_WORDS);
  const FirstDesc fdA = first_desc<true>(d.A);          // (KA's first descriptors: requested with the phase record, as k_slot_a does)
  SlotState st = slot_read(R);
  if (st.ph == P_KB) {
    if (st.admm >= st.target) { st.ph = P_IDLE; slot_write(W, st); return; }
    GKb g{d.xg, d.v, d.t0, d.n};
    EKbK e{d.x, d.q, d.Minv, d.kf.rec + 4 * (size_t)d.n, d.sigma, d.xg, d.xs};
    process_rows<2>(d.B, g, e, lds.kb);
    __syncthreads();
    const double BN = block_max(e.bn, lds.kb.red);
    put_partial(d.part, SL_BN, BN);
    if (blockIdx.x == 0 && threadIdx.x == 0) { d.flags[F_DONE] = 0; d.flags[F_ITERS] = 0; }
    st.ph = P_F; st.k = 0;
  } else if (st.ph == P_F) {
    // (st.conv == 2 while the phase is P_F: the timing probe's records -- k_slot_probe_f -- the fold is paid for, then fixed scalars: a CG left running for
    //  hundreds of iterations past convergence ends in 0 / 0, and a NaN residual reads as "converged")
    if (kf_iteration(d, st.k, st.cap, st.admm & 1, st.conv == 2 ? 2 : 0, lds.f, par)) {
      if (st.k >= st.cap) { st.ph = P_KA; st.used = st.k; st.conv = 0; }      // stopped at the cap: the next launch runs KA
      else st.k += 1;
    } else {                                             // converged after k - 1 iterations (k = 1: the start met the tolerance): KA right here
      __syncthreads();
      slot_ka(d, lds.ka, st.k - 1, 1, fdA, st.admm, st.target, st.seq);
      st.admm += 1; st.k = 0; st.ph = P_KB;
    }
  } else if (st.ph == P_KA) {
    slot_ka(d, lds.ka, st.used, st.conv, fdA, st.admm, st.target, st.seq, par ^ 1);
    st.admm += 1; st.k = 0; st.ph = P_KB;
  }
  slot_write(W, st);
}


}  // namespace

void kb_rhs(Dev &d) { LAUNCH(k_kb, d, d); }
bool pcg_fused(const Dev &d) { return d.fused != 0; }
void k1(Dev &d, int i) { if (d.fused && i > 0) LAUNCH(k_k1f, d, d, i); else LAUNCH(k_k1, d, d, i, 0); }
void k2(Dev &d, int i) { if (d.fused) LAUNCH(k_k2f, d, d, i); else LAUNCH(k_k2, d, d, 0); }
void kv(Dev &d, int i) { if (d.n >= 2 * kGrid * kBlock) LAUNCH(k_kv<2>, d, d, i, 0); else LAUNCH(k_kv<1>, d, d, i, 0); }
void ka(Dev &d, int budget) { LAUNCH(k_ka, d, d, budget); }
bool slots_supported(const Dev &d) { return (d.fused != 0 || wbx_slots(d)) && d.slot != nullptr; }
// The F1 slot kernel reads Dev from device memory (k_slot1): upload {head, Dev} when it differs from what was uploaded last.  Called where a chunk
// begins (slot_begin, ctl_begin) and by the timing probe -- eager calls, never inside a stream capture (a copy node captured into a string of slots
// would be replayed with every replay).  Stream-ordered: the launches that follow on d.stream see the new copy.
void dev_publish(Dev &d) {
  if (!d.f1.on) return;
  Impl &p = im(d);
  if (!p.dev_block) {
    HIP_CHECK(hipMalloc(&p.dev_block, sizeof(F1DevBlock)));
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&p.pin_block), sizeof(F1DevBlock), hipHostMallocDefault));
    p.shadow_block = static_cast<unsigned char *>(std::calloc(1, sizeof(F1DevBlock)));
    std::memset(p.pin_block, 0, sizeof(F1DevBlock));
    p.shadow_block[0] = 1;                               // (differs from any real block: the first call uploads)
  }
  F1DevBlock nb;
  std::memset(static_cast<void *>(&nb), 0, sizeof(nb));
  nb.h = F1Head{d.f1.stream, d.f1.blk, d.part, d.slot, d.A.nblk, 0};
  std::memcpy(static_cast<void *>(&nb.d), static_cast<const void *>(&d), sizeof(Dev));
  if (std::memcmp(&nb, p.shadow_block, sizeof(nb)) == 0) return;
  // (the pinned staging block may still be the source of an upload in flight: drain the stream before rewriting it -- rare: a setting changed)
  HIP_CHECK(hipStreamSynchronize(st(d)));
  std::memcpy(p.shadow_block, &nb, sizeof(nb));
  std::memcpy(p.pin_block, &nb, sizeof(nb));
  HIP_CHECK(hipMemcpyAsync(p.dev_block, p.pin_block, sizeof(nb), hipMemcpyHostToDevice, st(d)));
}
void dev_release(Dev &d) {
  Impl &p = im(d);
  if (p.dev_block) { (void)hipFree(p.dev_block); p.dev_block = nullptr; }
  if (p.pin_block) { (void)hipHostFree(p.pin_block); p.pin_block = nullptr; }
  if (p.shadow_block) { std::free(p.shadow_block); p.shadow_block = nullptr; }
}
// the F1 kernels are templates on D (replica vectors) and MIX (per-block mixing: far columns / spill slots, backend.h DevF1::mix)
template <class F>
static void f1_dispatch(const Dev &d, F &&f) {
  auto with_m = [&](auto Dc, auto Mc) { if (d.f1.wt) f(Dc, Mc, std::true_type{}); else f(Dc, Mc, std::false_type{}); };
  auto with_d = [&](auto Dc) { if (d.f1.mix) with_m(Dc, std::true_type{}); else with_m(Dc, std::false_type{}); };
  switch (d.f1.D) {
    case 1: with_d(std::integral_constant<int, 1>{}); break;
    case 2: with_d(std::integral_constant<int, 2>{}); break;
    case 3: with_d(std::integral_constant<int, 3>{}); break;
    default: with_d(std::integral_constant<int, 4>{}); break;
  }
}
void slot_begin(Dev &d, int target, int cap) { HIP_CHECK(hipSetDevice(d.device)); dev_publish(d); hipLaunchKernelGGL(k_slot_init, dim3(1), dim3(1), 0, st(d), d.slot, target, cap, ++im(d).epoch); }
void slot_pair(Dev &d) {
  if (wbx_slots(d)) { wbx_slot_pair(d); return; }      // Woodbury direct mode: X, Y (wbdirect_hip.hip)
  if (d.f1.on) {
    const F1DevBlock *db = static_cast<const F1DevBlock *>(im(d).dev_block);      // (current as of the chunk's slot_begin / ctl_begin: dev_publish)
    f1_dispatch(d, [&](auto Dc, auto Mc, auto Wc) {
      constexpr int DD = decltype(Dc)::value; constexpr bool MM = decltype(Mc)::value, WW = decltype(Wc)::value;
      hipLaunchKernelGGL((k_slot1<DD, MM, WW>), dim3(kGrid), dim3(kBlock), 0, st(d), db, 0);
      hipLaunchKernelGGL((k_slot1<DD, MM, WW>), dim3(kGrid), dim3(kBlock), 0, st(d), db, 1);
    });
  }
  else if (d.kf.on) { LAUNCH(k_slotk, d, d, 0); LAUNCH(k_slotk, d, d, 1); }
  else { LAUNCH(k_slot_b, d, d); LAUNCH(k_slot_a, d, d); }
}
bool kf_supported() { return true; }
void kf_values(Dev &d, int cond) { if (d.kf.on) { HIP_CHECK(hipSetDevice(d.device)); LAUNCH(k_kf_values, d, d, cond); } }
void f1_refresh(Dev &d) { if (d.f1.on) { HIP_CHECK(hipSetDevice(d.device)); LAUNCH(k_f1_refresh, d, d); } }
int slot_seq(Dev &d) { return (im(d).pin_flags + F_COUNT)[SR_SEQ]; }      // slots executed since slot_begin, as of the last fetch (record A)
int slot_done(Dev &d) {        // ADMM iterations completed by the chunk, as of the last fetch_flags / fetch_res_flags (record A: written by the last A slot)
  const int *rec = im(d).pin_flags + F_COUNT;
  return rec[SR_ADMM];
}

// Progress of the running chunk, read on a side stream WITHOUT waiting for d.stream: slots executed and ADMM iterations completed
// according to the newer of the two records (a record is eight words written by one thread -- a read may mix two states of it, but
// both counters only grow, so neither is ever ahead of the truth).  Scheduling information only: Engine::exec_chunk tops the chunk's
// string of slot launches up before it runs dry; what the slots compute does not depend on how many of them are enqueued.
void slot_poll(Dev &d, int *seq, int *done) {
  HIP_CHECK(hipSetDevice(d.device));
  Impl &p = im(d);
  HIP_CHECK(hipMemcpyAsync(p.pin_poll, d.slot, sizeof(int) * kSlotInts, hipMemcpyDeviceToHost, p.side));
  HIP_CHECK(hipStreamSynchronize(p.side));
  if (p.pin_poll[2 * SR_WORDS] != p.epoch) { *seq = 0; *done = 0; return; }      // the chunk's first launch (k_slot_init) has not run yet
  const int *ra = p.pin_poll, *rb = p.pin_poll + SR_WORDS;
  const int *nw = ra[SR_SEQ] >= rb[SR_SEQ] ? ra : rb;
  *seq = nw[SR_SEQ]; *done = nw[SR_ADMM];
}

// Mean duration of one launch of a hot-path kernel, measured with a hipEvent pair on the solver's stream.
// Kernels run in probe mode (no convergence logic; Kv with alpha = beta = 0) on the solver's live buffers; the
// iterate state that KB/KA/Kv overwrite is saved and restored around the measurement.
// probe 16: the phase records say "PCG iteration k0 of a chunk that never ends", the tolerance can never be met: the launches that follow
// are the slot kernel's own F launches -- scalars from the fold, stopping test, record hand-over -- exactly as a solve runs them
__global__ void k_slot_probe_f(int *slot, double *scal, int k0, int conv) {
  for (int rec = 0; rec < 2; rec++) {
    int *r = slot + rec * SR_WORDS;
    r[SR_PHASE] = P_F; r[SR_K] = k0; r[SR_ADMM] = 0; r[SR_TARGET] = 1; r[SR_USED] = 0; r[SR_CONV] = conv; r[SR_CAP] = 1 << 20; r[SR_SEQ] = rec ? -1 : 0;
  }
  scal[S_TOL_NOW] = -1.0;
}
static void f1_probe_pair(Dev &d, int mode) {           // two consecutive F launches of the probe kernel (the double-buffered vectors alternate)
  f1_dispatch(d, [&](auto Dc, auto Mc, auto Wc) {
    constexpr int DD = decltype(Dc)::value; constexpr bool MM = decltype(Mc)::value, WW = decltype(Wc)::value;
    hipLaunchKernelGGL((k_f1_probe<DD, MM, WW>), dim3(kGrid), dim3(kBlock), 0, st(d), d, 2, mode);
    hipLaunchKernelGGL((k_f1_probe<DD, MM, WW>), dim3(kGrid), dim3(kBlock), 0, st(d), d, 3, mode);
  });
}
float time_kernel(Dev &d, int which, int reps) {
  HIP_CHECK(hipSetDevice(d.device));
  Impl &p = im(d);
  struct Save { double *ptr; size_t cnt; double *bak; };
  const size_t n = d.n, m = d.m;
  if (which >= 14 && which <= 18 && !d.f1.on && !(which == 16 && d.kf.on)) return 0.f;
  if (which == 20 && !wbx_active(d)) return 0.f;
  if (which == 21 && !(d.wb.on && d.wb.exact)) return 0.f;
  if (which == 23 && !wbf_active(d)) return 0.f;
  const bool wbx = which == 20;
  const size_t r3 = wbx ? 3 * (size_t)d.wb.r : 0, gp_ = wbx ? (size_t)d.wb.x.G * kWbMaxRows : 0;
  Save sv[] = {{d.x, n, nullptr}, {d.z, m, nullptr}, {d.y, m, nullptr}, {d.xs, n, nullptr}, {d.zt, m, nullptr}, {d.t0, m, nullptr},
               {d.v, m, nullptr}, {d.dx, n, nullptr}, {d.dy, m, nullptr}, {d.r, n, nullptr}, {d.uu, n, nullptr}, {d.p, n, nullptr},
               {d.s, n, nullptr}, {d.w, n, nullptr}, {d.t, m, nullptr}, {d.uu2, n, nullptr}, {d.ms, 2 * n, nullptr},
               {d.xg, n, nullptr}, {d.xsp, n, nullptr}, {d.ztg, m, nullptr}, {d.kf.rec, d.kf.on ? 8 * n : 0, nullptr},
               {d.wb.x.ls0, r3, nullptr}, {d.wb.x.ls1, r3, nullptr}, {d.wb.x.partG, gp_, nullptr}, {d.wb.x.partZ, gp_, nullptr}};
  int flags_bak[F_COUNT];
  HIP_CHECK(hipStreamSynchronize(st(d)));
  HIP_CHECK(hipMemcpy(flags_bak, d.flags, sizeof(flags_bak), hipMemcpyDeviceToHost));
  for (auto &s : sv) {
    if (!s.cnt) continue;
    HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&s.bak), s.cnt * sizeof(double)));
    HIP_CHECK(hipMemcpy(s.bak, s.ptr, s.cnt * sizeof(double), hipMemcpyDeviceToDevice));
  }
  auto K1 = [&](int pr) { LAUNCH(k_k1, d, d, 1, pr); };
  auto K2 = [&]() { LAUNCH(k_k2, d, d, 1); };
  auto KV = [&](int pr) { if (d.n >= 2 * kGrid * kBlock) LAUNCH(k_kv<2>, d, d, 1, pr); else LAUNCH(k_kv<1>, d, d, 1, pr); };
  auto launch = [&]() {
    switch (which) {
      case 0: LAUNCH(k_k1, d, d, 1, 1); break;
      case 1: LAUNCH(k_k2, d, d, 1); break;
      case 2: if (d.n >= 2 * kGrid * kBlock) LAUNCH(k_kv<2>, d, d, 1, 1); else LAUNCH(k_kv<1>, d, d, 1, 1); break;
      case 3: LAUNCH(k_kb, d, d); break;
      case 4: LAUNCH(k_ka, d, d, 0); break;
      case 5: K1(1); K2(); KV(1); break;      // one PCG iteration, reductions of partials skipped
      case 6: K1(2); K2(); KV(2); break;      // one PCG iteration as a solve executes it
      case 7: K1(2); KV(2); break;            // ... without K2   (6 minus 7 = K2's time inside the sequence, L2-cold like in a solve)
      case 8: K2(); KV(2); break;             // ... without K1
      case 9: K1(2); K2(); break;             // ... without Kv
      case 11: LAUNCH(k_k1f, d, d, 1); break;  // fused SpMV-A + vector update alone (alpha fixed by the stored history; drifts linearly, bounded)
      case 12: LAUNCH(k_k2f, d, d, 0); break;  // fused SpMV-B alone
      case 13: LAUNCH(k_k2f, d, d, 2); LAUNCH(k_k1f, d, d, 3); break;   // the same pair with the done flag set: what an early-exit pair costs
      case 14: f1_probe_pair(d, 1); break;   // F1 form without the scalar fold at the head of the launch (two consecutive iterations)
      case 16: slot_pair(d); break;           // two F launches of the slot kernel itself (records set up by k_slot_probe_f below): what a launch costs inside a solve
      case 17: case 18:                      // KA of the F1 form (17) / the chunk's first launch, transposed passes only (18)
        f1_dispatch(d, [&](auto Dc, auto Mc, auto Wc) {
          constexpr int DD = decltype(Dc)::value; constexpr bool MM = decltype(Mc)::value, WW = decltype(Wc)::value;
          hipLaunchKernelGGL((k_f1_ka_probe<DD, MM, WW>), dim3(kGrid), dim3(kBlock), 0, st(d), d, (int)(which == 18));
        });
        break;
      case 21: HIP_CHECK(hipMemsetAsync(d.flags + F_DONE, 0, sizeof(int), st(d))); wb_apply(d, 0, 1); break;      // (the last kernel marks the solve as converged: cleared per repetition)     // Woodbury direct mode, device-factorised form: the three kernels of M^-1 = K^-1 (long rows, S^-1 product, transposed long rows + x~)
      case 23: wbf_iteration(d); break;       // one ADMM iteration of the fused column-space direct mode (seven launches; the iterates move on: state saved and restored around the measurement)
      case 15: f1_probe_pair(d, 2); break;   // F1 form: one PCG iteration = one launch; two consecutive iterations as a solve runs them (buffers alternate, fold included)
      default: LAUNCH(k_k2f, d, d, 0); LAUNCH(k_k1f, d, d, 1); break;   // one FUSED PCG iteration (two kernels): repeated exact line-search steps, bounded
    }
  };
  if (which == 20) {                              // the Woodbury direct mode in two launches: `reps` ADMM iterations as one chunk (X, {Y, X} x (reps - 1), Y, X); ms per ITERATION
    wbx_chunk(d, 3);
    HIP_CHECK(hipEventRecord(p.ev0, st(d)));
    wbx_chunk(d, reps);
    HIP_CHECK(hipEventRecord(p.ev1, st(d)));
    HIP_CHECK(hipEventSynchronize(p.ev1));
    float msx = 0.f;
    HIP_CHECK(hipEventElapsedTime(&msx, p.ev0, p.ev1));
    for (auto &sx : sv) { if (!sx.cnt) continue; HIP_CHECK(hipMemcpy(sx.ptr, sx.bak, sx.cnt * sizeof(double), hipMemcpyDeviceToDevice)); HIP_CHECK(hipFree(sx.bak)); }
    HIP_CHECK(hipMemcpy(d.flags, flags_bak, sizeof(flags_bak), hipMemcpyHostToDevice));
    return msx / reps;
  }
  if (which == 16) {
    dev_publish(d);
    if (reps > 400) reps = 400;                 // (k advances by two per repetition; the alpha / gamma history holds kMaxCg entries)
    hipLaunchKernelGGL(k_slot_probe_f, dim3(1), dim3(1), 0, st(d), d.slot, d.scal, 2, d.kf.on ? 2 : 0);
  }
  if (which >= 10) HIP_CHECK(hipMemsetAsync(d.flags + F_DONE, which == 13 ? 1 : 0, sizeof(int), st(d)));   // (byte pattern 1 -> nonzero flag)
  for (int w = 0; w < 5; w++) launch();
  HIP_CHECK(hipEventRecord(p.ev0, st(d)));
  for (int r = 0; r < reps; r++) launch();
  HIP_CHECK(hipEventRecord(p.ev1, st(d)));
  HIP_CHECK(hipEventSynchronize(p.ev1));
  float ms = 0.f;
  HIP_CHECK(hipEventElapsedTime(&ms, p.ev0, p.ev1));
  for (auto &s : sv) {
    if (!s.cnt) continue;
    HIP_CHECK(hipMemcpy(s.ptr, s.bak, s.cnt * sizeof(double), hipMemcpyDeviceToDevice));
    HIP_CHECK(hipFree(s.bak));
  }
  HIP_CHECK(hipMemcpy(d.flags, flags_bak, sizeof(flags_bak), hipMemcpyHostToDevice));
  return ms / reps;
}

// Diagnostic: workgroup phase stamps of the last launches (count <= kGrid * 16); false when built without OSQP_HIP_KTRACE.
bool ktrace_read(Dev &d, unsigned long long *out, int count) {
#ifdef OSQP_HIP_KTRACE
  HIP_CHECK(hipSetDevice(d.device));
  HIP_CHECK(hipStreamSynchronize(st(d)));
  if (count > kGrid * kTraceSlots) count = kGrid * kTraceSlots;
  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ktrace), sizeof(unsigned long long) * count));
  return true;
#else
  (void)d; (void)out; (void)count;
  return false;
#endif
}


}  // namespace be
}  // namespace osqp_hip
