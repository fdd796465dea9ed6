"""Determinism of independent solver handles driven from concurrent threads: serial results of 8 problems, then `rounds` rounds of
the same solves on a 4-thread pool; prints every difference (seed, which of the handle's two solves, iterations).
    python tools/thread_stress.py [rounds] [threads]"""
import os, sys, warnings
from multiprocessing.pool import ThreadPool
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT]
warnings.simplefilter('ignore')
import numpy as np
import osqp_amd, problems

N = int(os.environ.get('STRESS_N', 3000))


def solve(seed):
    P, q, A, l, u = problems.banded_qp(N, window=60, seed=seed)
    m = osqp_amd.OSQP()
    m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=20000)
    out = []
    for _ in range(2):
        m.update_settings(warm_starting=False)
        r = m.solve()
        st = m._solver.hip_stats()
        out.append((r.info.status_val, r.info.iter, r.x.copy(), r.y.copy(), int(st['pcg_iters_total']), int(st['slot_topups']), r.info.rho_updates))
    return out


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seeds = list(range(40, 48))
serial = [solve(s) for s in seeds]
serial2 = [solve(s) for s in seeds]
bad = 0
for a, b, s in zip(serial, serial2, seeds):
    for k in range(2):
        if a[k][1] != b[k][1] or not np.array_equal(a[k][2], b[k][2]):
            print('SERIAL repeat differs: seed', s, 'solve', k, a[k][1], b[k][1]); bad += 1
for rnd in range(rounds):
    with ThreadPool(threads) as pool:
        thr = pool.map(solve, seeds)
    for a, b, s in zip(serial, thr, seeds):
        for k in range(2):
            if a[k][1] == b[k][1] and np.array_equal(a[k][2], b[k][2]) and (a[k][4], a[k][5]) != (b[k][4], b[k][5]):
                print('round %d seed %d solve %d: SAME result, but pcg %d/%d topups %d/%d' % (rnd, s, k, a[k][4], b[k][4], a[k][5], b[k][5]), flush=True)
            if a[k][1] != b[k][1] or not np.array_equal(a[k][2], b[k][2]):
                bad += 1
                print('round %d seed %d solve %d: serial it %d pcg %d topups %d rho_upd %d | threaded it %d pcg %d topups %d rho_upd %d | max|dx| %.2e'
                      % (rnd, s, k, a[k][1], a[k][4], a[k][5], a[k][6], b[k][1], b[k][4], b[k][5], b[k][6], np.abs(a[k][2] - b[k][2]).max()), flush=True)
print('rounds %d threads %d: %d differences' % (rounds, threads, bad))
