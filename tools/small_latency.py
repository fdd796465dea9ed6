"""Latency of osqp_solve on small QPs: the one-launch direct path (default when it applies) vs the multi-kernel PCG engine."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT]
warnings.simplefilter('ignore')
import numpy as np, osqp_amd, problems
CASES = {'random_qp n=50 m=100': lambda: problems.random_qp(),
         'mpc n=120 m=240': lambda: (lambda P, q, A, L, U: (P, q, A, L[0], U[0]))(*problems.mpc_batch(1)),
         'banded n=150 m=300': lambda: problems.banded_qp(150, window=20)}
for name, g in CASES.items():
    P, q, A, l, u = g()
    row = []
    for mode in ('1', '0'):
        os.environ['OSQP_HIP_SMALL_DIRECT'] = mode
        m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, warm_starting=False)
        m.solve()
        ts = []
        for _ in range(5):
            t = time.perf_counter(); r = m.solve(); ts.append(time.perf_counter() - t)
        row.append('%s: %s, %d it, %.2f ms (launches %d)' % ('direct, one launch' if mode == '1' else 'multi-kernel PCG', r.info.status, r.info.iter, 1e3 * np.median(ts), m._solver.hip_stats()['kernel_launches']))
    print('%-22s %s' % (name, ' | '.join(row)), flush=True)
