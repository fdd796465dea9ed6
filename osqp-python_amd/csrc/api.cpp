// api.cpp -- extern "C" surface of libosqp_hip.so (include/osqp_hip.h).  Thin: every call forwards to Engine.
#include <cstdio>
#include <memory>
#include <new>

#include "engine.hpp"

using osqp_hip::Engine;

namespace {
Engine *eng(OSQPSolver *s) { return s ? reinterpret_cast<Engine *>(s->work) : nullptr; }

// Every entry point runs its body under this guard: a failed HIP call (osqp_hip::DeviceError) or an allocation failure
// comes back to the caller as an osqp_error_type value, never as an exception across the C ABI and never as abort().
template <class F>
OSQPInt guarded(OSQPSolver *s, F &&f) {
  Engine *e = eng(s);
  if (!e) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  try { return f(*e); }
  catch (const std::bad_alloc &) { return OSQP_MEM_ALLOC_ERROR; }
  catch (const osqp_hip::DeviceError &err) { std::fprintf(stderr, "osqp_hip: device error: %s\n", err.what()); return OSQP_ALGEBRA_LOAD_ERROR; }
  catch (...) { return OSQP_LINSYS_SOLVER_INIT_ERROR; }
}
}

extern "C" {

const char *osqp_version(void) { return "1.0.0-hip.r1"; }
void osqp_hip_set_default_print(osqp_hip_print_fn fn, void *user) { Engine::set_default_print(fn, user); }
OSQPInt osqp_hip_set_print(OSQPSolver *s, osqp_hip_print_fn fn, void *user) { return guarded(s, [&](Engine &e) { e.set_print(fn, user); return (int)OSQP_NO_ERROR; }); }
const char *osqp_hip_backend(void) { return osqp_hip::be::name(); }

OSQPInt osqp_capabilities(void) { return OSQP_CAPABILITY_INDIRECT_SOLVER | OSQP_CAPABILITY_UPDATE_MATRICES; }

// Defaults: values of the v1.0.0 C core as recalled in SURVEY.md Appendix C [UPSTREAM-UNVERIFIED]; the reference's
// tests pass the settings that matter explicitly (basic_test.py:20-31).  linsys_solver defaults to the only solver
// this algebra has (as the reference pairs 'cuda' only with 'indirect', tests/conftest.py:26-29).
void osqp_set_default_settings(OSQPSettings *s) {
  if (!s) return;
  s->device = 0; s->linsys_solver = OSQP_INDIRECT_SOLVER; s->verbose = 1; s->warm_starting = 1; s->scaling = 10; s->polishing = 0;
  s->rho = 0.1; s->rho_is_vec = 1; s->sigma = 1e-6; s->alpha = 1.6;
  s->cg_max_iter = 50; s->cg_tol_reduction = 10; s->cg_tol_fraction = 0.15; s->cg_precond = OSQP_DIAGONAL_PRECONDITIONER;
  s->adaptive_rho = 1; s->adaptive_rho_interval = 0; s->adaptive_rho_fraction = 0.4; s->adaptive_rho_tolerance = 5.0;
  s->max_iter = 4000; s->eps_abs = 1e-3; s->eps_rel = 1e-3; s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4;
  s->scaled_termination = 0; s->check_termination = 25; s->check_dualgap = 0; s->time_limit = 1e10;
  s->delta = 1e-6; s->polish_refine_iter = 3;
}

OSQPInt osqp_setup(OSQPSolver **solverp, const OSQPCscMatrix *P, const OSQPFloat *q, const OSQPCscMatrix *A,
                   const OSQPFloat *l, const OSQPFloat *u, OSQPInt m, OSQPInt n, const OSQPSettings *settings) {
  if (!solverp) return OSQP_DATA_VALIDATION_ERROR;
  *solverp = nullptr;
  Engine *e = new (std::nothrow) Engine();
  if (!e) return OSQP_MEM_ALLOC_ERROR;
  int err;
  try { err = e->setup(P, q, A, l, u, m, n, settings); }
  catch (const std::bad_alloc &) { err = OSQP_MEM_ALLOC_ERROR; }
  catch (const osqp_hip::DeviceError &) { err = OSQP_ALGEBRA_LOAD_ERROR; }
  catch (...) { err = OSQP_LINSYS_SOLVER_INIT_ERROR; }
  if (err) { delete e; return err; }
  *solverp = &e->pub;
  return OSQP_NO_ERROR;
}

OSQPInt osqp_solve(OSQPSolver *s) { return guarded(s, [&](Engine &e) { return e.solve(); }); }
OSQPInt osqp_cleanup(OSQPSolver *s) { Engine *e = eng(s); if (e) delete e; return OSQP_NO_ERROR; }
OSQPInt osqp_warm_start(OSQPSolver *s, const OSQPFloat *x, const OSQPFloat *y) { return guarded(s, [&](Engine &e) { return e.warm_start(x, y); }); }
OSQPInt osqp_cold_start(OSQPSolver *s) { return guarded(s, [&](Engine &e) { return e.cold_start(); }); }
OSQPInt osqp_update_data_vec(OSQPSolver *s, const OSQPFloat *q, const OSQPFloat *l, const OSQPFloat *u) {
  return guarded(s, [&](Engine &e) { return e.update_data_vec(q, l, u); });
}
OSQPInt osqp_update_data_mat(OSQPSolver *s, const OSQPFloat *Px, const OSQPInt *Pi, OSQPInt Pn, const OSQPFloat *Ax, const OSQPInt *Ai, OSQPInt An) {
  return guarded(s, [&](Engine &e) { return e.update_data_mat(Px, Pi, Pn, Ax, Ai, An); });
}
OSQPInt osqp_update_settings(OSQPSolver *s, const OSQPSettings *ns) { return guarded(s, [&](Engine &e) { return e.update_settings(ns); }); }
OSQPInt osqp_update_rho(OSQPSolver *s, OSQPFloat rho) { return guarded(s, [&](Engine &e) { return e.update_rho(rho); }); }
void osqp_get_dimensions(OSQPSolver *s, OSQPInt *m, OSQPInt *n) { Engine *e = eng(s); if (e) { if (m) *m = e->m; if (n) *n = e->n; } }

OSQPInt osqp_adjoint_derivative_compute(OSQPSolver *, OSQPFloat *, OSQPFloat *) { return OSQP_FUNC_NOT_IMPLEMENTED; }
OSQPInt osqp_adjoint_derivative_get_mat(OSQPSolver *, OSQPCscMatrix *, OSQPCscMatrix *) { return OSQP_FUNC_NOT_IMPLEMENTED; }
OSQPInt osqp_adjoint_derivative_get_vec(OSQPSolver *, OSQPFloat *, OSQPFloat *, OSQPFloat *) { return OSQP_FUNC_NOT_IMPLEMENTED; }
OSQPInt osqp_codegen(OSQPSolver *, const char *, const char *, OSQPCodegenDefines *) { return OSQP_FUNC_NOT_IMPLEMENTED; }
void osqp_set_default_codegen_defines(OSQPCodegenDefines *d) {
  if (!d) return;
  d->embedded_mode = 1; d->float_type = 0; d->printing_enable = 0; d->profiling_enable = 0; d->interrupt_enable = 0; d->derivatives_enable = 0;
}

OSQPInt osqp_hip_get_stats(OSQPSolver *s, OSQPHipStats *out) { return guarded(s, [&](Engine &e) { return e.get_stats(out); }); }
OSQPInt osqp_hip_time_kernel(OSQPSolver *s, OSQPInt which, OSQPInt reps, double *ms) { return guarded(s, [&](Engine &e) { return e.time_kernel(which, reps, ms); }); }
OSQPInt osqp_hip_trace_read(OSQPSolver *s, unsigned long long *out, OSQPInt count) { return guarded(s, [&](Engine &e) { return e.trace_read(out, count); }); }
OSQPInt osqp_hip_test_spmv(OSQPSolver *s, OSQPInt which, const OSQPFloat *in, OSQPFloat *out) { return guarded(s, [&](Engine &e) { return e.test_spmv(which, in, out); }); }
OSQPInt osqp_hip_set_rho_eq_factor(OSQPSolver *s, OSQPFloat f) { return guarded(s, [&](Engine &e) { return e.set_rho_eq_factor(f); }); }
void osqp_hip_default_policy(OSQPHipPolicy *p) { Engine::default_policy(p); }
void osqp_hip_set_default_policy(const OSQPHipPolicy *p) { Engine::set_default_policy(p); }
OSQPInt osqp_hip_set_policy(OSQPSolver *s, const OSQPHipPolicy *p) { return guarded(s, [&](Engine &e) { return e.set_policy(p); }); }
OSQPInt osqp_hip_get_policy(OSQPSolver *s, OSQPHipPolicy *p) { return guarded(s, [&](Engine &e) { return e.get_policy(p); }); }
OSQPInt osqp_hip_batch_solve(OSQPSolver *s, OSQPInt nbatch, const OSQPFloat *q, const OSQPFloat *l, const OSQPFloat *u, OSQPFloat *x, OSQPFloat *y, OSQPFloat *rec, OSQPInt warm) {
  return guarded(s, [&](Engine &e) { return e.batch_solve(nbatch, q, l, u, x, y, rec, warm); });
}
OSQPInt osqp_hip_batch_solve_mat(OSQPSolver *s, OSQPInt nbatch, const OSQPFloat *Px, const OSQPFloat *Ax, const OSQPFloat *q, const OSQPFloat *l, const OSQPFloat *u, OSQPFloat *x, OSQPFloat *y, OSQPFloat *rec, OSQPInt warm) {
  return guarded(s, [&](Engine &e) { return e.batch_solve(nbatch, q, l, u, x, y, rec, warm, nullptr, Px, Ax); });
}
OSQPInt osqp_hip_batch_solve_mat_device(OSQPSolver *s, OSQPInt nbatch, const OSQPFloat *Px, const OSQPFloat *Ax, const OSQPFloat *q, const OSQPFloat *l, const OSQPFloat *u, OSQPFloat *x, OSQPFloat *y, OSQPFloat *rec, OSQPInt warm, void *stream) {
  return guarded(s, [&](Engine &e) { return e.batch_solve_device(nbatch, q, l, u, x, y, rec, warm, stream, Px, Ax); });
}
OSQPInt osqp_hip_batch_solve_device(OSQPSolver *s, OSQPInt nbatch, const OSQPFloat *q, const OSQPFloat *l, const OSQPFloat *u, OSQPFloat *x, OSQPFloat *y, OSQPFloat *rec, OSQPInt warm, void *stream) {
  return guarded(s, [&](Engine &e) { return e.batch_solve_device(nbatch, q, l, u, x, y, rec, warm, stream); });
}
OSQPInt osqp_hip_update_data_vec_device(OSQPSolver *s, const OSQPFloat *q, const OSQPFloat *l, const OSQPFloat *u, void *stream) {
  return guarded(s, [&](Engine &e) { return e.update_data_vec_device(q, l, u, stream); });
}
OSQPInt osqp_hip_warm_start_device(OSQPSolver *s, const OSQPFloat *x, const OSQPFloat *y, void *stream) {
  return guarded(s, [&](Engine &e) { return e.warm_start_device(x, y, stream); });
}
OSQPInt osqp_hip_get_scaling(OSQPSolver *s, OSQPFloat *D, OSQPFloat *E, OSQPFloat *c) { return guarded(s, [&](Engine &e) { return e.get_scaling(D, E, c); }); }
OSQPInt osqp_hip_get_reordering(OSQPSolver *s, OSQPInt *perm_cols, OSQPInt *perm_rows) { return guarded(s, [&](Engine &e) { return e.get_reordering(perm_cols, perm_rows); }); }

}  // extern "C"

// ---- LinSysSolver slot ----
namespace {
struct LinSysImpl {
  Engine eng;
  const OSQPFloat *prim = nullptr, *dual = nullptr;
  OSQPInt polishing = 0;
};
LinSysImpl *ls(OSQPHipLinSysSolver *s) { return s ? static_cast<LinSysImpl *>(s->impl) : nullptr; }
template <class F>
OSQPInt ls_guarded(OSQPHipLinSysSolver *s, F &&f) {
  LinSysImpl *im = ls(s);
  if (!im) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  try { return f(*im); }
  catch (const std::bad_alloc &) { return OSQP_MEM_ALLOC_ERROR; }
  catch (const osqp_hip::DeviceError &) { return OSQP_ALGEBRA_LOAD_ERROR; }
  catch (...) { return OSQP_LINSYS_SOLVER_INIT_ERROR; }
}
const char *ls_name(OSQPHipLinSysSolver *) { return "HIP reduced-KKT PCG (gfx950)"; }
OSQPInt ls_solve(OSQPHipLinSysSolver *s, OSQPFloat *b, OSQPInt) {
  return ls_guarded(s, [&](LinSysImpl &im) {
    double tol_rel = 1e-7, tol_abs = 0.0;                                  // no residual information: relative reduction
    if (im.polishing) tol_rel = 1e-12;
    else if (im.dual && *im.dual > 0) { tol_rel = 1e-14; tol_abs = im.eng.settings.cg_tol_fraction * *im.dual; }
    int it = 0;
    const int err = im.eng.ls_solve(b, tol_rel, tol_abs, &it);
    s->pcg_iters = it;
    return err;
  });
}
void ls_update_settings(OSQPHipLinSysSolver *s, const OSQPSettings *st) {
  LinSysImpl *im = ls(s);
  if (!im || !st) return;
  if (st->cg_max_iter > 0) im->eng.settings.cg_max_iter = st->cg_max_iter;
  if (st->cg_tol_fraction > 0 && st->cg_tol_fraction < 1) im->eng.settings.cg_tol_fraction = st->cg_tol_fraction;
  if (st->cg_tol_reduction > 0) im->eng.settings.cg_tol_reduction = st->cg_tol_reduction;
}
void ls_warm_start(OSQPHipLinSysSolver *s, const OSQPFloat *x) { (void)ls_guarded(s, [&](LinSysImpl &im) { return im.eng.ls_warm_start(x); }); }
OSQPInt ls_adjoint(OSQPHipLinSysSolver *) { return OSQP_FUNC_NOT_IMPLEMENTED; }
void ls_free(OSQPHipLinSysSolver *s) { if (!s) return; delete ls(s); delete s; }
OSQPInt ls_update_matrices(OSQPHipLinSysSolver *s, const OSQPCscMatrix *P, const OSQPInt *Pidx, OSQPInt Pn, const OSQPCscMatrix *A, const OSQPInt *Aidx, OSQPInt An) {
  // the slot hands over the FULL updated matrices (plus the changed positions, which a value-only refresh does not need)
  (void)Pidx; (void)Pn; (void)Aidx; (void)An;
  return ls_guarded(s, [&](LinSysImpl &im) {
    return im.eng.update_data_mat(P ? P->x : nullptr, nullptr, P ? P->p[P->n] : 0, A ? A->x : nullptr, nullptr, A ? A->p[A->n] : 0);
  });
}
OSQPInt ls_update_rho_vec(OSQPHipLinSysSolver *s, const OSQPFloat *rho_vec, OSQPFloat) {
  return ls_guarded(s, [&](LinSysImpl &im) { return im.eng.ls_set_rho_vec(rho_vec); });
}
}  // namespace

extern "C" OSQPInt osqp_hip_linsys_init(OSQPHipLinSysSolver **out, const OSQPCscMatrix *P, const OSQPCscMatrix *A, const OSQPFloat *rho_vec,
                                        const OSQPSettings *settings, const OSQPFloat *scaled_prim_res, const OSQPFloat *scaled_dual_res,
                                        OSQPInt polishing) {
  if (!out) return OSQP_DATA_VALIDATION_ERROR;
  *out = nullptr;
  try {
    std::unique_ptr<LinSysImpl> im(new LinSysImpl());
    im->prim = scaled_prim_res; im->dual = scaled_dual_res; im->polishing = polishing;
    const OSQPInt err = im->eng.ls_setup(P, A, rho_vec, settings);
    if (err) return err;
    OSQPHipLinSysSolver *s = new OSQPHipLinSysSolver();
    s->type = OSQP_INDIRECT_SOLVER; s->name = ls_name; s->solve = ls_solve; s->update_settings = ls_update_settings;
    s->warm_start = ls_warm_start; s->adjoint_derivative = ls_adjoint; s->free = ls_free; s->update_matrices = ls_update_matrices;
    s->update_rho_vec = ls_update_rho_vec; s->nthreads = 1; s->pcg_iters = 0; s->impl = im.release();
    *out = s;
    return OSQP_NO_ERROR;
  }
  catch (const std::bad_alloc &) { return OSQP_MEM_ALLOC_ERROR; }
  catch (const osqp_hip::DeviceError &) { return OSQP_ALGEBRA_LOAD_ERROR; }
  catch (...) { return OSQP_LINSYS_SOLVER_INIT_ERROR; }
}
