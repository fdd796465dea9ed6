// api.cpp -- extern "C" surface of libosqp_hip.so (include/osqp_hip.h).  Thin: every call forwards to Engine.
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
  catch (const osqp_hip::DeviceError &) { return OSQP_ALGEBRA_LOAD_ERROR; }
  catch (...) { return OSQP_LINSYS_SOLVER_INIT_ERROR; }
}
}

extern "C" {

const char *osqp_version(void) { return "1.0.0-hip.r1"; }
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
OSQPInt osqp_hip_batch_solve(OSQPSolver *s, OSQPInt nbatch, const OSQPFloat *q, const OSQPFloat *l, const OSQPFloat *u, OSQPFloat *x, OSQPFloat *y, OSQPFloat *rec, OSQPInt warm) {
  return guarded(s, [&](Engine &e) { return e.batch_solve(nbatch, q, l, u, x, y, rec, warm); });
}
OSQPInt osqp_hip_get_scaling(OSQPSolver *s, OSQPFloat *D, OSQPFloat *E, OSQPFloat *c) { return guarded(s, [&](Engine &e) { return e.get_scaling(D, E, c); }); }

}  // extern "C"
