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

// Block reductions; all threads get the result.  NW = waves per workgroup.  With ONE wave per problem (NW = 1) a
// reduction is six __shfl_xor steps: no LDS, no barrier.
template <int NW>
struct Red {
  double *s;   // >= 16 doubles of LDS (NW > 1 only)
  __device__ __forceinline__ double sum(double v) const {
    if constexpr (NW == 1) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      return v;
    } else {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
      __syncthreads();
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < NW; w++) t += s[w];
      __syncthreads();
      return t;
    }
  }
  __device__ __forceinline__ double max(double v) const {
    if constexpr (NW == 1) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v = nmax(v, __shfl_xor(v, o, 64));
      return v;
    } else {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v = nmax(v, __shfl_down(v, o, 64));
      if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
      __syncthreads();
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < NW; w++) t = nmax(t, s[w]);
      __syncthreads();
      return t;
    }
  }
  __device__ __forceinline__ void sum_max(double &a, double &b) const {   // a: sum, b: max
    if constexpr (NW == 1) { a = sum(a); b = max(b); }
    else {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b = nmax(b, __shfl_down(b, o, 64)); }
      if ((threadIdx.x & 63) == 0) { s[threadIdx.x >> 6] = a; s[8 + (threadIdx.x >> 6)] = b; }
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

// kBB = threads per problem: 64 (one wave: barriers are free, reductions are pure shuffles; small problems) or 256.
// EA / EB > 0: every lane keeps EA entries of A and EB entries of B (value + column) in registers for the whole solve
// (entry k belongs to lane k % kBB); an SpMV is then  prod[k] = val * v[col]  for the lane's own entries (LDS only),
// a barrier, and one lane per row summing its segment of prod -- no matrix traffic inside the ADMM / PCG loops and a
// balanced first phase (the MPC rows have 1..13 entries).  EA = EB = 0: generic row loops reading the matrices from
// global memory (L1/L2), for patterns with more than 8 * kBB entries per matrix.
template <int kBB, int EA, int EB>
__global__ __launch_bounds__(kBB) void k_batch_admm(BatchParams P) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = P.n, m = P.m, tid = threadIdx.x, b = blockIdx.x;
  if (b >= P.nbatch) return;
  // ---- LDS carve ----
  double *x = sm, *xs = x + n, *r = xs + n, *zv = r + n, *p = zv + n, *Kp = p + n, *q = Kp + n, *Minv = q + n, *dx = Minv + n, *tn = dx + n;
  double *z = tn + n, *y = z + m, *t = y + m, *l = t + m, *u = l + m, *rho = u + m, *zt = rho + m, *dy = zt + m;
  Red<kBB / 64> red{dy + m};
  double *prod = dy + m + 16;                       // max(nnzA, nnzB) products (register path only)
  const DevCsr &A = P.A, &B = P.B;
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
      for (int i = tid; i < m; i += kBB) { double a = 0.0; for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) a += prod[k]; f(i, a); }
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
      for (int j = tid; j < n; j += kBB) { double a = 0.0; for (int k = B.rowptr[j]; k < B.rowptr[j + 1]; k++) a += prod[k]; f(j, a); }
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
  const double eqf = (n_ineq == 0.0) ? 1e3 : P.eq_factor;  // engine.cpp classify_constraints()
  double rho_bar = P.rho0;
  auto set_rho = [&](double rb) {
    for (int i = tid; i < m; i += kBB) {
      const double li = l[i], ui = u[i];
      int ty = (li < -OSQP_INFTY * 1e-4 && ui > OSQP_INFTY * 1e-4) ? -1 : ((ui - li < 1e-4) ? 1 : 0);
      if (!P.rho_is_vec) ty = 0;
      rho[i] = ty == -1 ? 1e-6 : (ty == 1 ? eqf * rb : rb);                                       // _osqp.py:520-522
    }
    __syncthreads();
    for (int j = tid; j < n; j += kBB) {                   // Jacobi preconditioner = 1/diag(K)
      double sacc = 0.0, dg = 0.0;
      for (int k = B.rowptr[j]; k < B.rowptr[j + 1]; k++) { const int c = B.col[k]; const double a = B.val[k]; if (c == j) dg = a; if (c >= n) sacc += rho[c - n] * a * a; }
      Minv[j] = P.precond ? 1.0 / (dg + sacc) : 1.0;
    }
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
  for (int i = tid; i < m; i += kBB) zt[i] = z[i];
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
    pri_u = red.max(a1); ax_u = red.max(a2); z_u = red.max(a3); pri_s = red.max(a4); ax_s = red.max(a5); z_s = red.max(a6);
    dy_u = red.max(a7); dy_s = red.max(a8); pinf_lhs = red.sum(s1);
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
    dua_u = red.max(b1); px_u = red.max(b2); aty_u = red.max(b3); dua_s = red.max(b4); px_s = red.max(b5); aty_s = red.max(b6);
    dxn_u = red.max(b7); dxn_s = red.max(b8); qn_s = red.max(b9); qn_u = red.max(b10);
    xpx = red.sum(t1); qx = red.sum(t2); qdx = red.sum(t3);
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
    double rz_l = 0, rn_l = 0, bn_l = 0;
    for (int i = tid; i < m; i += kBB) t[i] = rho[i] * z[i] - y[i];
    __syncthreads();
    applyB(nullptr, t, [&](int j, double sA) { Kp[j] = P.sigma * x[j] - q[j] + sA; });          // Kp holds rhs for a moment
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
    // ---- z~ = A xs; x, z, y update (_osqp.py:660-703) ----
    applyA(xs, [&](int i, double a) {
      const double rh = rho[i], yi = y[i];
      const double zr = P.alpha * a + (1.0 - P.alpha) * z[i];
      const double zn = fmin(fmax(zr + yi / rh, l[i]), u[i]);
      const double dyi = rh * (zr - zn);
      y[i] = yi + dyi; dy[i] = dyi; z[i] = zn; zt[i] = a;
    });
    for (int j = tid; j < n; j += kBB) { const double xo = x[j], xn = P.alpha * xs[j] + (1.0 - P.alpha) * xo; dx[j] = xn - xo; x[j] = xn; }
    __syncthreads();

    const bool at_check = (P.check > 0 && iter % P.check == 0) || iter >= P.max_iter;
    const bool at_rho = P.rho_interval > 0 && iter % P.rho_interval == 0;
    if (!at_check && !at_rho) continue;
    residuals();
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
  // ---- store: x = D x, y = cinv E y (_osqp.py:1110-1112); certificates in place of x / y for infeasible problems ----
  const bool pinf = status == OSQP_PRIMAL_INFEASIBLE || status == OSQP_PRIMAL_INFEASIBLE_INACCURATE;
  const bool dinf = status == OSQP_DUAL_INFEASIBLE || status == OSQP_DUAL_INFEASIBLE_INACCURATE;
  for (int j = tid; j < n; j += kBB) P.x[(size_t)b * n + j] = dinf ? (unsc ? P.D[j] * dx[j] : dx[j]) : (pinf ? NAN : (P.scaling ? P.D[j] * x[j] : x[j]));
  for (int i = tid; i < m; i += kBB) P.y[(size_t)b * m + i] = pinf ? (unsc ? P.E[i] * dy[i] : dy[i]) : (dinf ? NAN : (P.scaling ? P.cinv * P.E[i] * y[i] : y[i]));
  if (tid == 0) {
    double *rc = P.rec + (size_t)b * 8;
    rc[0] = status; rc[1] = iter; rc[2] = obj; rc[3] = prim_res; rc[4] = dual_res; rc[5] = rho_bar; rc[6] = rho_updates; rc[7] = (double)pcg_total;
  }
}

}  // namespace

// LDS needed per problem (bytes); 0 if the problem does not fit one workgroup's LDS.  nnz > 0 adds the product buffer of
// the register-resident path.
size_t batch_lds_bytes_nnz(int n, int m, int nnz) {
  const size_t b = sizeof(double) * ((size_t)10 * n + (size_t)8 * m + 16 + (size_t)nnz);
  return b <= 64 * 1024 ? b : 0;
}
size_t batch_lds_bytes(int n, int m) { return batch_lds_bytes_nnz(n, m, 0); }

int batch_solve(Dev &d, const BatchParams &p) {
  if (hipSetDevice(d.device) != hipSuccess) return OSQP_ALGEBRA_LOAD_ERROR;
  hipStream_t st = static_cast<hipStream_t>(d.stream);
  const int mx = p.A.nnz > p.B.nnz ? p.A.nnz : p.B.nnz;
  const size_t lds_reg = batch_lds_bytes_nnz(p.n, p.m, mx), lds_gen = batch_lds_bytes(p.n, p.m);
  const char *force = std::getenv("OSQP_HIP_BATCH_VARIANT");      // debugging: "w64", "w256", "generic"
  const int e64 = (mx + 63) / 64, e256 = (mx + 255) / 256;
  const bool can64 = lds_reg && e64 <= 24 && p.n <= 1024 && p.m <= 2048, can256 = lds_reg && e256 <= 8;
  const bool use64 = force ? !std::strcmp(force, "w64") && can64 : can64;
  const bool use256 = !use64 && (force ? !std::strcmp(force, "w256") && can256 : can256);
#define BATCH_LAUNCH(TB, E, LDS) hipLaunchKernelGGL((k_batch_admm<TB, E, E>), dim3(p.nbatch), dim3(TB), LDS, st, p)
  if (use64) {
    if (e64 <= 8) BATCH_LAUNCH(64, 8, lds_reg); else if (e64 <= 16) BATCH_LAUNCH(64, 16, lds_reg); else BATCH_LAUNCH(64, 24, lds_reg);
  } else if (use256) {
    if (e256 <= 2) BATCH_LAUNCH(256, 2, lds_reg); else if (e256 <= 4) BATCH_LAUNCH(256, 4, lds_reg); else BATCH_LAUNCH(256, 8, lds_reg);
  } else if (lds_gen) {
    BATCH_LAUNCH(256, 0, lds_gen);
  } else {
    return OSQP_FUNC_NOT_IMPLEMENTED;
  }
#undef BATCH_LAUNCH
  hipError_t e = hipStreamSynchronize(st);
  if (e != hipSuccess) throw DeviceError(std::string("osqp_hip: batch kernel failed: ") + hipGetErrorString(e));
  return OSQP_NO_ERROR;
}

}  // namespace be
}  // namespace osqp_hip
