"""Aggregate ADMM it/s of T independent config-2 solves running concurrently on ONE GPU (one solver handle + HIP stream per
Python thread; solve() releases the GIL).  A single n=100k solve is launch-latency bound, so concurrent solves overlap."""
import os, sys, time, warnings
from multiprocessing.pool import ThreadPool
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [os.path.join(R, 'osqp-python_amd'), R]
warnings.simplefilter('ignore')
import osqp_amd, problems
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
P, q, A, l, u = problems.banded_qp(n)
def make():
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, check_termination=25, adaptive_rho_interval=50, warm_starting=False, verbose=False)
    m.solve(); return m
for T in (1, 2, 4, 8):
    ms = [make() for _ in range(T)]
    with ThreadPool(T) as pool:
        t0 = time.perf_counter(); its = pool.map(lambda m: m.solve().info.iter, ms); dt = time.perf_counter() - t0
    print('T=%d concurrent solves: %.1f ms wall, %d ADMM iterations -> %.0f ADMM it/s aggregate (%.0f per solve)' % (T, dt * 1e3, sum(its), sum(its) / dt, sum(its) / dt / T), flush=True)
