"""BASELINE configs 2-4 (+ an unstructured variant of config 2 and a 10x larger one) on one MI355X: iterations, time,
ADMM it/s, PCG iterations, and the achieved algorithmic GB/s of the two PCG SpMV kernels (in-sequence pair time).

    python tools/config_sweep.py [name ...]"""
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT]
warnings.simplefilter('ignore')
import numpy as np  # noqa: E402
import osqp_amd  # noqa: E402
import problems  # noqa: E402

CASES = {
    'config2_banded_100k': lambda: problems.banded_qp(100000),
    'config2_banded_100k_seed1': lambda: problems.banded_qp(100000, seed=1),
    'config2_banded_100k_seed2': lambda: problems.banded_qp(100000, seed=2),
    'config2_unstructured_100k': lambda: problems.banded_qp(100000, window=100000),
    'banded_200k': lambda: problems.banded_qp(200000),
    'banded_300k': lambda: problems.banded_qp(300000),
    'banded_500k': lambda: problems.banded_qp(500000),
    'banded_1M': lambda: problems.banded_qp(1000000),
    'config3_lasso_5k_10k': lambda: problems.lasso_qp(5000, 10000),
    'config4_portfolio_10k_100': lambda: problems.portfolio_qp(10000, 100),
}
names = sys.argv[1:] or [k for k in CASES if k not in ('banded_200k', 'banded_300k', 'banded_500k', 'config2_banded_100k_seed1', 'config2_banded_100k_seed2')]
for name in names:
    P, q, A, l, u = CASES[name]()
    n, mm = len(q), len(l)
    m = osqp_amd.OSQP()
    kw = dict(verbose=False, eps_abs=1e-6, eps_rel=1e-6, max_iter=50000); kw.update(eval(os.environ.get('SWEEP_KW', '{}')))     # e.g. SWEEP_KW="dict(adaptive_rho_tolerance=2)"
    t = time.perf_counter(); m.setup(P, q, A, l, u, **kw); ts = time.perf_counter() - t
    t = time.perf_counter(); r = m.solve(); t1 = time.perf_counter() - t
    st = m._solver.hip_stats()
    nnzA, nnzB = int(st['nnzA']), int(st['nnzB'])
    pair_ms = m._solver.hip_time_kernel(10, 50)
    pair_bytes = (12 * nnzA + 4 * (mm + 1) + 16 * mm + 96 * n) + (12 * nnzB + 4 * (n + 1) + 8 * (n + mm) + 8 * n + 32 * n)
    print('%-28s n=%d m=%d nnzA=%d nnzB=%d | setup %.2f s | %s in %d it, %.1f ms = %.0f ADMM it/s, %.2f PCG it/ADMM it | PCG pair %.1f us, %.1f MB -> %.0f GB/s (%.0f %% of 8 TB/s)'
          % (name, n, mm, nnzA, nnzB, ts, r.info.status, r.info.iter, 1e3 * t1, r.info.iter / t1, st['pcg_iters_total'] / max(r.info.iter, 1),
             1e3 * pair_ms, pair_bytes / 1e6, pair_bytes / (pair_ms * 1e-3) / 1e9, 100 * pair_bytes / (pair_ms * 1e-3) / 8e12), flush=True)
