// TEST INFRASTRUCTURE: the rules of osqp-python_amd/csrc/policy.h (one text, compiled for the host driver and for the device's k_decide)
// behind a C ABI, so that the CPU tier can exercise each rule on its own (tests/test_policy_rules.py).  Never part of the product.
#include <cstring>
#include <string>

#include "../../osqp-python_amd/csrc/policy.h"

using namespace osqp_hip;

namespace {
struct Field { const char *name; int kind; size_t off; };      // kind 0: int, 1: double
#define FI(f) {#f, 0, offsetof(Ctl, f)}
#define FD(f) {#f, 1, offsetof(Ctl, f)}
const Field kFields[] = {
  FI(ct), FI(ari), FI(max_iter), FI(tightW), FI(has_quad), FI(persist), FI(esc_on), FI(stall_on), FI(full_budget), FI(cap_max), FI(budget_slack), FI(budget_min),
  FD(tightF), FD(tol_exp), FD(cg_tol_fraction), FD(cg_tol_reduction), FD(rho_tolerance), FD(budget_tolerate), FD(budget_sigma),
  FI(iter), FI(cap), FI(tight_seen), FI(last_side), FI(stalled_checks), FI(rho_updates), FI(escalations),
  FD(tol_rel), FD(tol_abs), FD(eps_cg_prev), FD(stall), FD(best_dua), FD(prev_aobj), FD(rho_bar), FD(obj_val), FD(rho_estimate),
  FI(ch_next), FI(ch_tight), FI(ch_kind), FI(ch_at_check),
};
const Field *find(const char *name) { for (const Field &f : kFields) if (!std::strcmp(f.name, name)) return &f; return nullptr; }
}  // namespace

extern "C" {
void *pp_new() { Ctl *c = new Ctl(); std::memset(c, 0, sizeof(Ctl)); c->stall = 1.0; c->best_dua = INFINITY; c->eps_cg_prev = INFINITY; return c; }
void pp_free(void *p) { delete static_cast<Ctl *>(p); }
int pp_set(void *p, const char *name, double v) {
  Ctl *c = static_cast<Ctl *>(p);
  if (!std::strcmp(name, "budget0")) { c->budget[0] = (int)v; return 0; }
  if (!std::strcmp(name, "budget1")) { c->budget[1] = (int)v; return 0; }
  const Field *f = find(name);
  if (!f) return -1;
  char *b = reinterpret_cast<char *>(c) + f->off;
  if (f->kind) *reinterpret_cast<double *>(b) = v; else *reinterpret_cast<int *>(b) = (int)v;
  return 0;
}
double pp_get(void *p, const char *name) {
  Ctl *c = static_cast<Ctl *>(p);
  if (!std::strcmp(name, "budget0")) return c->budget[0];
  if (!std::strcmp(name, "budget1")) return c->budget[1];
  const Field *f = find(name);
  if (!f) return NAN;
  const char *b = reinterpret_cast<const char *>(c) + f->off;
  return f->kind ? *reinterpret_cast<const double *>(b) : (double)*reinterpret_cast<const int *>(b);
}
void pp_next_chunk(void *p) { ctl_next_chunk(*static_cast<Ctl *>(p)); }
double pp_chunk_tol_abs(void *p) { return ctl_chunk_tol_abs(*static_cast<Ctl *>(p)); }
static void fill(int *fl, int sum, int sumsq, int n, int mx, int unconv, int stag) {
  std::memset(fl, 0, sizeof(int) * F_COUNT);
  fl[F_STAT_SUM] = sum; fl[F_STAT_SUMSQ] = sumsq; fl[F_STAT_N] = n; fl[F_STAT_MAX] = mx; fl[F_STAT_UNCONV] = unconv; fl[F_STAT_STAG] = stag;
}
int pp_next_budget(void *p, int cur, int sum, int sumsq, int n, int mx, int unconv, int stag) {
  int fl[F_COUNT]; fill(fl, sum, sumsq, n, mx, unconv, stag);
  return ctl_next_budget(*static_cast<Ctl *>(p), cur, fl);
}
void pp_budget_rule(void *p, int sum, int sumsq, int n, int mx, int unconv, int stag) {
  int fl[F_COUNT]; fill(fl, sum, sumsq, n, mx, unconv, stag);
  ctl_budget_rule(*static_cast<Ctl *>(p), fl);
}
// residual block reduced to what the rules read: scaled primal / dual residuals and their normalisations
static void res_block(double *res, double pri, double nrm_p, double dua, double nrm_d) {
  for (int q = 0; q < R_COUNT; q++) res[q] = 0.0;
  res[R_PRI_S] = pri; res[R_AX_S] = nrm_p; res[R_Z_S] = nrm_p; res[R_DUA_S] = dua; res[R_ATY_S] = nrm_d; res[R_PX_S] = nrm_d; res[R_QN_S] = nrm_d;
}
int pp_rho_rule(void *p, double pri, double nrm_p, double dua, double nrm_d) {
  double res[R_COUNT]; res_block(res, pri, nrm_p, dua, nrm_d);
  return ctl_rho_rule(*static_cast<Ctl *>(p), res) ? 1 : 0;
}
void pp_tol_rule(void *p, double dua) {
  double res[R_COUNT]; res_block(res, 0.0, 1.0, dua, 1.0);
  ctl_tol_rule(*static_cast<Ctl *>(p), res);
}
void pp_init_tol(void *p, double dua0) {
  double res[R_COUNT]; res_block(res, 0.0, 1.0, dua0, 1.0);
  ctl_init_tol(*static_cast<Ctl *>(p), res);
}
}
