"""Robustness of the PCG path across settings: four mid-size problems x a grid of inner-solver / ADMM settings; every solve must
end SOLVED, and an iteration count above 3x the default-settings count is flagged (run-away / stalling regimes).
    python tools/robustness_grid.py"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT]
warnings.simplefilter('ignore')
import numpy as np, osqp_amd, problems

PROBS = {'banded_20k': lambda: problems.banded_qp(20000), 'unstructured_20k': lambda: problems.banded_qp(20000, window=20000),
         'lasso_500x1000': lambda: problems.lasso_qp(500, 1000), 'portfolio_2000x50': lambda: problems.portfolio_qp(2000, 50)}
GRID = [dict(), dict(cg_tol_fraction=0.05), dict(cg_tol_fraction=0.1), dict(cg_tol_fraction=0.3), dict(cg_tol_fraction=0.5),
        dict(cg_max_iter=10), dict(cg_max_iter=25), dict(alpha=1.0), dict(alpha=1.8), dict(rho=0.01), dict(rho=1.0), dict(rho=10.0),
        dict(adaptive_rho_interval=25), dict(adaptive_rho_interval=100), dict(adaptive_rho=False), dict(scaling=0), dict(check_termination=10),
        dict(eps_abs=1e-8, eps_rel=1e-8), dict(sigma=1e-4), dict(adaptive_rho_tolerance=2), dict(adaptive_rho_tolerance=20)]
bad = 0
for name, gen in PROBS.items():
    P, q, A, l, u = gen()
    base = None
    for kw in GRID:
        st = dict(eps_abs=1e-6, eps_rel=1e-6, max_iter=30000, verbose=False); st.update(kw)
        m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, **st)
        t = time.perf_counter(); r = m.solve(); dt = time.perf_counter() - t
        s = m._solver.hip_stats()
        if base is None: base = r.info.iter
        flag = '' if (r.info.status_val == 1 and r.info.iter <= 3 * base + 100) else '   <-- CHECK'
        if kw.get('adaptive_rho') is False or 'rho' in kw or kw.get('eps_abs') or kw.get('scaling') == 0: flag = flag if r.info.status_val != 1 else ''      # (legitimately slower)
        bad += bool(flag)
        print('%-18s %-32s %-8s %6d it %7.1f PCG/it %8.1f ms%s' % (name, ','.join('%s=%s' % kv for kv in kw.items()) or 'default', r.info.status, r.info.iter,
                                                                  s['pcg_iters_total'] / max(r.info.iter, 1), 1e3 * dt, flag), flush=True)
print('flagged', bad)
