"""Sensitivity of config 2 to the inner (PCG) tolerance fraction and iteration cap: ADMM iterations, PCG iterations, time."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT]
warnings.simplefilter('ignore')
import osqp_amd, problems  # noqa: E402
P, q, A, l, u = problems.banded_qp(100000)
for frac in (0.05, 0.1, 0.15, 0.25, 0.4, 0.6):
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, cg_tol_fraction=frac)
    m.solve(); m.update_settings(warm_starting=False)
    t = time.perf_counter(); r = m.solve(); dt = time.perf_counter() - t
    st = m._solver.hip_stats()
    print('cg_tol_fraction %.2f: %s, %d ADMM it, %.2f PCG it/it, %.1f ms, %.0f it/s' % (frac, r.info.status, r.info.iter, st['pcg_iters_total'] / r.info.iter, 1e3 * dt, r.info.iter / dt), flush=True)
