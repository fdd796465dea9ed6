"""Determinism of the whole handle API under concurrency: every thread drives its own solver through setup / solve / update q /
update bounds / warm start / update matrices / polish / batch solve; results must be bitwise those of the serial run.
    python tools/thread_stress_api.py [rounds] [threads]"""
import os, sys, warnings
from multiprocessing.pool import ThreadPool
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT]
warnings.simplefilter('ignore')
import numpy as np, scipy.sparse as sp
import osqp_amd, problems


def scenario(seed):
    rng = np.random.default_rng(seed)
    out = []
    # large path (multi-kernel PCG)
    P, q, A, l, u = problems.banded_qp(4000, window=80, seed=seed)
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=20000)
    r = m.solve(); out.append((r.info.iter, r.x.copy(), r.y.copy()))
    m.update(q=q * (1 + 0.01 * rng.standard_normal(len(q)))); r = m.solve(); out.append((r.info.iter, r.x.copy(), r.y.copy()))
    m.update(l=l - 0.05, u=u + 0.05); r = m.solve(); out.append((r.info.iter, r.x.copy(), r.y.copy()))
    m.warm_start(x=r.x * 0.9, y=r.y * 0.9); r = m.solve(); out.append((r.info.iter, r.x.copy(), r.y.copy()))
    Pt = sp.triu(P, format='csc')
    m.update(Px=Pt.data * (1 + 0.02 * rng.random(Pt.nnz)), Ax=A.data * (1 + 0.02 * rng.standard_normal(A.nnz))); r = m.solve(); out.append((r.info.iter, r.x.copy(), r.y.copy()))
    m.update_settings(polishing=True); r = m.solve(); out.append((r.info.iter, r.x.copy(), r.y.copy(), r.info.status_polish))
    # small path (one launch) + batch
    Pb, qb, Ab, L, U = problems.mpc_batch(32, seed=seed)
    s = osqp_amd.OSQP(); s.setup(Pb, qb, Ab, L[0], U[0], eps_abs=1e-6, eps_rel=1e-6, verbose=False, polishing=True)
    r = s.solve(); out.append((r.info.iter, r.x.copy(), r.y.copy(), r.info.status_polish))
    for _ in range(2):
        x, y, rec = s._solver.hip_batch_solve(l=L, u=U); out.append((int(rec[:, 1].sum()), x.copy(), y.copy()))
    return out


def same(a, b):
    return all(p[0] == q_[0] and np.array_equal(p[1], q_[1]) and np.array_equal(p[2], q_[2]) and p[3:] == q_[3:] for p, q_ in zip(a, b))


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seeds = list(range(70, 78))
serial = [scenario(s) for s in seeds]
bad = 0
for rnd in range(rounds):
    with ThreadPool(threads) as pool:
        thr = pool.map(scenario, seeds)
    for a, b, s in zip(serial, thr, seeds):
        if not same(a, b):
            bad += 1
            print('round %d seed %d differs at steps' % (rnd, s), [k for k, (p, q_) in enumerate(zip(a, b)) if not (p[0] == q_[0] and np.array_equal(p[1], q_[1]))], flush=True)
print('rounds %d threads %d scenarios %d: %d differences' % (rounds, threads, len(seeds), bad))
