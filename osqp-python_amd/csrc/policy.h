// policy.h -- the chunk-boundary rules of the ADMM driver as plain functions over one state block (Ctl), compiled TWICE from this
// one text: by the host driver (engine.cpp: launch-per-iteration form, the test-only host simulator, verbose solves, and the rare
// boundaries the device hands back) and by the device (backend_hip.hip k_decide: the default -- a solve runs without a host round
// trip per termination check).  Everything here is arithmetic on a few dozen scalars; nothing touches a vector.
//
// What the rules are (each cites the reference where it has a counterpart; DESIGN.md sections 2, 2.1 for the measurements behind them):
//   ctl_next_chunk      where the next chunk of ADMM iterations ends: the next termination check (_osqp.py:1254-1262), the next rho
//                       adaptation point (:1229-1245), or the start of the tight-tolerance window in front of it
//   ctl_stage1          residuals -> info fields, tolerances, the termination test proper (:998-1077) and the FIRST stage of the
//                       infeasibility tests (:796-878) -- the second stage needs two more SpMVs and is run by the host on demand
//   ctl_rho_rule        adaptive rho (:880-930) with this engine's square-root tolerance and persistence test
//   ctl_tol_rule        PCG tolerance for the next chunk (a fraction of the scaled dual residual, never loosening; dropped while the
//                       iterates run away)
//   ctl_budget_rule     PCG iteration limit of the next chunk of the same kind (mean + 3 sigma of what the last one needed) and the
//                       cap escalation while the inner solver stagnates
#pragma once
#include <math.h>

#include "../../include/osqp_hip.h"
#include "backend.h"

#if defined(__HIPCC__)
#define OSQP_HD __host__ __device__
#else
#define OSQP_HD
#endif

namespace osqp_hip {

constexpr double kPolRhoMin = 1e-6, kPolRhoMax = 1e6;            // _osqp.py:25-26
constexpr double kPolCgTolAbsMin = 1e-13;

enum CtlStatus { CTL_RUNNING = 0, CTL_DONE = 1, CTL_NEED_HOST = 2 };
// why a boundary was handed to the host
enum CtlNeed { NEED_NONE = 0, NEED_PINF = 1, NEED_DINF = 2, NEED_MAXITER = 4, NEED_REFACTOR = 8 /* Woodbury direct mode: the device-side inversion failed its check */ };

struct Ctl {
  // ---- settings snapshot (constant during a solve)
  int ct, ari, max_iter, tightW, m, scaling, scaled_termination, check_dualgap, has_quad, persist, esc_on, stall_on, full_budget, cap_max;
  double tightF, tol_exp, cg_tol_fraction, cg_tol_reduction, rho_tolerance, eps_abs, eps_rel, eps_pinf, eps_dinf, c, cinv;
  double budget_tolerate, budget_sigma; int budget_slack, budget_min;      // budget_min: smallest PCG limit per solve (2; 1 with the Woodbury preconditioner, whose solves take one iteration)
  // ---- state
  int iter, cap, budget[2], tight_seen, last_side, stalled_checks, rho_updates, status, need, osqp_status;
  double tol_rel, tol_abs, eps_cg_prev, stall, best_dua, prev_aobj, rho_bar;
  // the chunk in flight
  int ch_next, ch_tight, ch_kind, ch_at_check;
  int chunk_done;                 // device: set by the slot kernel that completes the chunk's last ADMM iteration
  int rho_flag;                   // device: rho_bar changed at this boundary -- the conditional set_rho / precond kernels act
  int stage2;                     // CtlNeed bits: second stage of the infeasibility tests pending (its kernels run, then ctl_boundary_stage2)
  double inf_nd_p, inf_nd_d;      // ||dy||, ||dx|| of the pending tests (_osqp.py:806, :836); inf_thr_d = eps_dual_inf ||dx|| (threshold of the A dx test)
  double inf_thr_d; int inf_unscaled;
  int boundaries;                 // boundaries processed (progress, polled by the host)
  int seq_begin;                  // slot launches executed when the chunk in flight began (the host's rate estimate: launches of THIS chunk per finished iteration)
  int last_flags[F_COUNT];        // PCG statistics of the chunk the last boundary closed (the host needs them when it finishes a boundary)
  // ---- statistics of the solve
  double pcg_total, pcg_max, pcg_unconv; int escalations;
  double kind_sum[3], kind_n[3];  // PCG iterations / solves per chunk kind (sizes the strings of slot launches)
  // ---- results of the last check
  double obj_val, prim_res, dual_res, dual_obj_val, duality_gap, rel_kkt_error, rho_estimate;
  double res[R_COUNT];            // the residual block the last check saw
  // ---- verbose output: the checks whose iteration count is a multiple of kCtlPrintInterval (_osqp.py:32, :1230-1231), as a ring the host prints
  // from -- the lines of a device-driven solve are then the same lines, whenever the host gets to read them
  int nlog, log_pad;
  double log[16][5];              // {iter, obj_val, prim_res, dual_res, rho_bar at the check}
  // ---- how many slot launches every chunk of this solve consumed (device-driven solves; index = boundaries processed when the chunk began):
  // the next solve of the handle is fed from it (Engine::run_device_driven) instead of from polled progress
  int seq_end;                    // device: value of the slot counter when the chunk in flight completed its last ADMM iteration (written with chunk_done)
  int hist_pad;
  int hist[64];
};
constexpr int kCtlLog = 16, kCtlPrintInterval = 200, kCtlHist = 64;

OSQP_HD inline int pol_imin(int a, int b) { return a < b ? a : b; }
OSQP_HD inline int pol_imax(int a, int b) { return a > b ? a : b; }
OSQP_HD inline double pol_clamp_rho(double r) { return fmin(fmax(r, kPolRhoMin), kPolRhoMax); }

// (_osqp.py:880-908, scaled quantities)
OSQP_HD inline double pol_rho_estimate(double rho_bar, const double *res) {
  const double pri = res[R_PRI_S] / (fmax(res[R_AX_S], res[R_Z_S]) + 1e-10);
  const double dua = res[R_DUA_S] / (fmax(fmax(res[R_ATY_S], res[R_PX_S]), res[R_QN_S]) + 1e-10);
  return pol_clamp_rho(rho_bar * sqrt(pri / (dua + 1e-10)));
}

// PCG tolerance of the first chunk from the residuals of the starting point: a warm start near the optimum must not be perturbed by a
// loose first-chunk solve (warm_start_test.py:52-57 expects < 10 iterations)
OSQP_HD inline void ctl_init_tol(Ctl &c, const double *r0) {
  c.tol_rel = 1e-14; c.tol_abs = kPolCgTolAbsMin;
  const double eps0 = c.cg_tol_fraction * r0[R_DUA_S];
  if (isfinite(eps0) && eps0 > kPolCgTolAbsMin) c.tol_abs = eps0;                          // absolute, like every later chunk
  else { c.tol_rel = 1.0 / c.cg_tol_reduction; c.tol_abs = kPolCgTolAbsMin; }              // dual-feasible start (e.g. q = 0): relative
  c.eps_cg_prev = INFINITY;
}

// The chunk that starts at c.iter: fills ch_*; also the one-time initialisation of the tight window's budget.
OSQP_HD inline void ctl_next_chunk(Ctl &c) {
  int next = c.max_iter;
  if (c.ct > 0) next = pol_imin(next, (c.iter / c.ct + 1) * c.ct);
  if (c.ari > 0) next = pol_imin(next, (c.iter / c.ari + 1) * c.ari);
  int tight = 0;
  if (c.tightW > 0) {
    const int ts = (c.iter / c.ari + 1) * c.ari - c.tightW;       // start of the tight window before the next adaptation point
    if (c.iter >= ts) tight = 1;
    else if (ts < next) next = ts;
  }
  if (c.full_budget) c.budget[0] = c.budget[1] = c.cap;
  if (tight && !c.tight_seen) { c.budget[1] = pol_imin(c.cap, 3 * c.budget[0] + 2); c.tight_seen = 1; }
  c.ch_next = next; c.ch_tight = tight;
  c.ch_at_check = ((c.ct > 0 && next % c.ct == 0) || next >= c.max_iter || (c.ari > 0 && next % c.ari == 0)) ? 1 : 0;
  c.ch_kind = tight ? 2 : ((c.ari > 0 && c.tightW > 0 && c.iter % c.ari == 0) ? 0 : 1);
}
// PCG tolerance (absolute part) the chunk in flight runs with
OSQP_HD inline double ctl_chunk_tol_abs(const Ctl &c) { return c.ch_tight ? fmax(c.tightF * c.tol_abs, kPolCgTolAbsMin) : c.tol_abs; }

// cg_max_iter escalation: when most solves of a chunk ran into the cap having reduced their residual by less than 10x the inner solver
// is STAGNATING (unbounded LP / rank-deficient QP with n > m: DESIGN.md section 5): the cap doubles (up to cap_max).
OSQP_HD inline void ctl_escalate(Ctl &c, const int *flags) {
  const int t = c.ch_tight;
  if (c.esc_on && c.budget[t] >= c.cap && c.cap < c.cap_max && flags[F_STAT_STAG] * 2 > pol_imax(1, flags[F_STAT_N])) {
    c.cap = pol_imin(c.cap_max, 2 * c.cap);
    c.escalations += 1;
  }
}
// PCG limit per solve for the next chunk of this kind: mean + 3 sigma of the PCG counts of the last one (+1), never above its max + 1;
// +1 / doubled when solves ran out
OSQP_HD inline int ctl_next_budget(const Ctl &c, int cur, const int *flags) {
  if (flags[F_STAT_UNCONV] * 4 > pol_imax(1, flags[F_STAT_N]) && flags[F_STAT_UNCONV] > c.budget_tolerate * flags[F_STAT_N]) return pol_imin(c.cap, pol_imax(cur + 2, 2 * cur));
  if (flags[F_STAT_UNCONV] > c.budget_tolerate * flags[F_STAT_N]) return pol_imin(c.cap, cur + 1);
  const double cnt = pol_imax(1, flags[F_STAT_N]), mean = flags[F_STAT_SUM] / cnt;
  const double var = fmax(0.0, flags[F_STAT_SUMSQ] / cnt - mean * mean);
  const int q3 = (int)ceil(mean + c.budget_sigma * sqrt(var));
  return pol_imin(c.cap, pol_imax(c.budget_min, pol_imin(flags[F_STAT_MAX], q3)) + c.budget_slack);
}
OSQP_HD inline void ctl_budget_rule(Ctl &c, const int *flags) {
  ctl_escalate(c, flags);
  c.budget[c.ch_tight] = ctl_next_budget(c, c.budget[c.ch_tight], flags);
}
// statistics of a finished chunk (every boundary)
OSQP_HD inline void ctl_account(Ctl &c, const int *flags) {
  c.pcg_total += flags[F_STAT_SUM];
  c.pcg_max = fmax(c.pcg_max, (double)flags[F_STAT_MAX]);
  c.pcg_unconv += flags[F_STAT_UNCONV];
  if (flags[F_STAT_N] > 0) { c.kind_sum[c.ch_kind] = flags[F_STAT_SUM]; c.kind_n[c.ch_kind] = flags[F_STAT_N]; }
  c.iter = c.ch_next;
}

// info fields of a check (_osqp.py:705-764) + the v1 gap fields (engine.cpp update_gap_info; [UPSTREAM-UNVERIFIED] formulas)
OSQP_HD inline void ctl_info(Ctl &c, const double *res) {
  const bool unsc = c.scaling && !c.scaled_termination;
  const double ci = c.scaling ? c.cinv : 1.0;
  c.obj_val = (0.5 * res[R_XPX] + res[R_QX]) * ci;
  c.prim_res = (c.m == 0) ? 0.0 : (unsc ? res[R_PRI_U] : res[R_PRI_S]);
  c.dual_res = unsc ? c.cinv * res[R_DUA_U] : res[R_DUA_S];
  c.dual_obj_val = (-0.5 * res[R_XPX] - res[R_SUPP]) * ci;
  c.duality_gap = c.obj_val - c.dual_obj_val;
  const double pn = unsc ? fmax(res[R_AX_U], res[R_Z_U]) : fmax(res[R_AX_S], res[R_Z_S]);
  const double dn = unsc ? c.cinv * fmax(fmax(res[R_ATY_U], res[R_PX_U]), res[R_QN_U]) : fmax(fmax(res[R_ATY_S], res[R_PX_S]), res[R_QN_S]);
  const double gn = fmax(fabs(c.obj_val), fabs(c.dual_obj_val));
  const double tiny = 1e-10;
  c.rel_kkt_error = fmax(fmax(c.m == 0 ? 0.0 : c.prim_res / (pn + tiny), c.dual_res / (dn + tiny)), fabs(c.duality_gap) / (gn + tiny));
  for (int q = 0; q < R_COUNT; q++) c.res[q] = res[q];
}

// First stage of check_termination (_osqp.py:998-1077).  Returns an osqp_status_type value when the status is decided without
// the infeasibility tests' second stage, 0 when the solve goes on, and -1 with c.need set when the host has to run the second stage
// (is_primal_infeasible :815-818 / is_dual_infeasible :846-872 need A' dy, P dx, A dx).
OSQP_HD inline int ctl_stage1(Ctl &c, const double *res, bool approximate, bool *pri_ok_out, bool *dua_ok_out) {
  double ea = c.eps_abs, er = c.eps_rel, epi = c.eps_pinf, edi = c.eps_dinf;
  if (approximate) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
  const bool unsc = c.scaling && !c.scaled_termination;
  if (c.prim_res > OSQP_INFTY || c.dual_res > OSQP_INFTY || c.prim_res != c.prim_res || c.dual_res != c.dual_res) return OSQP_NON_CVX;   // :1025-1028
  bool pri_ok = false, dua_ok = false;
  c.need = NEED_NONE;
  c.inf_unscaled = unsc ? 1 : 0;
  if (c.m == 0) pri_ok = true;
  else {
    const double eps_pri = ea + er * (unsc ? fmax(res[R_AX_U], res[R_Z_U]) : fmax(res[R_AX_S], res[R_Z_S]));   // :728-751
    if (c.prim_res < eps_pri) pri_ok = true;
    else {
      const double nd = unsc ? res[R_DY_U] : res[R_DY_S];
      if (nd > epi && res[R_PINF_LHS] < -epi * nd) { c.need |= NEED_PINF; c.inf_nd_p = nd; }
    }
  }
  const double mx = unsc ? c.cinv * fmax(fmax(res[R_ATY_U], res[R_PX_U]), res[R_QN_U]) : fmax(fmax(res[R_ATY_S], res[R_PX_S]), res[R_QN_S]);   // :766-794
  if (c.dual_res < ea + er * mx) dua_ok = true;
  else {
    const double nd = unsc ? res[R_DX_U] : res[R_DX_S], sc = unsc ? c.c : 1.0;
    if (nd > edi && res[R_QDX] < -sc * edi * nd) { c.need |= NEED_DINF; c.inf_nd_d = nd; c.inf_thr_d = edi * nd; }
  }
  const bool gap_ok = !c.check_dualgap || fabs(c.duality_gap) < ea + er * fmax(fabs(c.obj_val), fabs(c.dual_obj_val));
  if (pri_ok_out) *pri_ok_out = pri_ok;
  if (dua_ok_out) *dua_ok_out = dua_ok;
  if (pri_ok && dua_ok && gap_ok) return approximate ? OSQP_SOLVED_INACCURATE : OSQP_SOLVED;
  return c.need ? -1 : 0;
}

// Second stage (r2: the block with R_ATDY_*, R_PDX_*, R_ADX_VIOL filled by the second-stage kernels for the tests in c.need):
// an osqp_status_type value when an infeasibility certificate holds (:815-818, :846-872), else 0.
OSQP_HD inline int ctl_stage2(const Ctl &c, const double *r2, bool approximate) {
  double epi = c.eps_pinf, edi = c.eps_dinf;
  if (approximate) { epi *= 10; edi *= 10; }
  const bool unsc = c.inf_unscaled != 0;
  if ((c.need & NEED_PINF) && (unsc ? r2[R_ATDY_U] : r2[R_ATDY_S]) < epi * c.inf_nd_p)
    return approximate ? OSQP_PRIMAL_INFEASIBLE_INACCURATE : OSQP_PRIMAL_INFEASIBLE;
  const double sc = unsc ? c.c : 1.0;
  if ((c.need & NEED_DINF) && (unsc ? r2[R_PDX_U] : r2[R_PDX_S]) < sc * edi * c.inf_nd_d && r2[R_ADX_VIOL] == 0.0)
    return approximate ? OSQP_DUAL_INFEASIBLE_INACCURATE : OSQP_DUAL_INFEASIBLE;
  return 0;
}

// adapt_rho (_osqp.py:910-930) on the indirect path.  Returns true when rho_bar changed.
// An update costs this path two small kernels, not a refactorisation -- the reason for the reference's factor-5 guard -- so the tolerance
// is spent on a square-root scale (the default 5 fires at 2.24; LPs keep the literal value), and an estimate that falls on the same side
// of rho by more than sqrt(tolerance) at two CONSECUTIVE adaptation points is applied as well.
OSQP_HD inline bool ctl_rho_rule(Ctl &c, const double *res) {
  const double rn = pol_rho_estimate(c.rho_bar, res), tol = pow(c.rho_tolerance, c.has_quad ? c.tol_exp : 1.0);
  c.rho_estimate = rn;
  const double st = sqrt(tol);
  const int side = rn > st * c.rho_bar ? 1 : (rn < c.rho_bar / st ? -1 : 0);
  const bool big = rn > tol * c.rho_bar || rn < c.rho_bar / tol;
  const bool persistent = c.persist && side != 0 && side == c.last_side;
  c.last_side = side;
  if (big || persistent) { c.rho_bar = rn; c.rho_updates++; c.last_side = 0; return true; }
  return false;
}

// inner tolerance for the next chunks: a fraction of the current SCALED dual residual, never loosening; while the dual residual does
// not improve on its best value AND |objective| keeps growing from check to check (an unbounded problem) it drops by 10x per check
OSQP_HD inline void ctl_tol_rule(Ctl &c, const double *res) {
  const double aobj = fabs(c.obj_val);
  const bool growing = aobj > 1.0 && aobj > 1.02 * c.prev_aobj;
  c.prev_aobj = aobj;
  if (!c.stall_on) c.stall = 1.0;
  else if (res[R_DUA_S] > 0.9 * c.best_dua) { if (growing && ++c.stalled_checks >= 2) c.stall = fmax(1e-8, 0.1 * c.stall); }
  else { c.stalled_checks = 0; c.stall = fmin(1.0, 10.0 * c.stall); }
  c.best_dua = fmin(c.best_dua, res[R_DUA_S]);
  double eps = c.cg_tol_fraction * res[R_DUA_S];
  eps = fmax(fmin(eps, c.eps_cg_prev), kPolCgTolAbsMin);
  if (isfinite(eps)) { c.eps_cg_prev = eps; c.tol_rel = 1e-14; c.tol_abs = fmax(eps * c.stall, kPolCgTolAbsMin); }
}

// the rest of a boundary that did not terminate: rho adaptation, tolerance, budget, next chunk.  Returns true when rho changed.
OSQP_HD inline bool ctl_boundary_continue(Ctl &c, const double *res, const int *flags) {
  bool rho_changed = false;
  if (c.ari > 0 && c.iter % c.ari == 0) rho_changed = ctl_rho_rule(c, res);
  c.rho_flag = rho_changed ? 1 : 0;
  ctl_tol_rule(c, res);
  ctl_budget_rule(c, flags);
  ctl_next_chunk(c);
  return rho_changed;
}
// A finished chunk, first part.  Returns
//   CTL_RUNNING    the next chunk is set up in ch_* (rho_flag: rho_bar changed) -- or, with c.stage2 != 0, the second stage of the
//                  infeasibility tests is pending: its kernels have to run, then ctl_boundary_stage2 finishes the boundary;
//   CTL_DONE       osqp_status set;
//   CTL_NEED_HOST  max_iter reached without convergence: the approximate-tolerance pass (:1264-1266) is the host's.
OSQP_HD inline int ctl_boundary(Ctl &c, const double *res, const int *flags) {
  ctl_account(c, flags);
  c.rho_flag = 0; c.need = NEED_NONE; c.stage2 = 0;
  c.boundaries++;
  if (!c.ch_at_check) { ctl_budget_rule(c, flags); ctl_next_chunk(c); return CTL_RUNNING; }
  ctl_info(c, res);
  if (c.iter % kCtlPrintInterval == 0) {
    double *e = c.log[c.nlog % kCtlLog];
    e[0] = c.iter; e[1] = c.obj_val; e[2] = c.prim_res; e[3] = c.dual_res; e[4] = c.rho_bar;
    c.nlog++;
  }
  const bool do_check = (c.ct > 0 && c.iter % c.ct == 0) || c.iter == c.max_iter;
  if (do_check) {
    const int st = ctl_stage1(c, res, false, nullptr, nullptr);
    if (st > 0) { c.osqp_status = st; return CTL_DONE; }
    if (st < 0) { c.stage2 = c.need; return CTL_RUNNING; }
  }
  if (c.iter >= c.max_iter) { c.need = NEED_MAXITER; return CTL_NEED_HOST; }
  ctl_boundary_continue(c, res, flags);
  return CTL_RUNNING;
}
// ... second part, after the second-stage kernels (res: the residual block with their results)
OSQP_HD inline int ctl_boundary_stage2(Ctl &c, const double *res, const int *flags) {
  const int st = ctl_stage2(c, res, false);
  c.stage2 = 0;
  if (st > 0) { c.osqp_status = st; return CTL_DONE; }
  if (c.iter >= c.max_iter) { c.need = NEED_MAXITER; return CTL_NEED_HOST; }
  ctl_boundary_continue(c, c.res, flags);
  return CTL_RUNNING;
}

}  // namespace osqp_hip
