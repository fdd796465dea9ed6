"""Step 4 of the multithread API scenario (update matrices + polish) in detail: serial vs threaded.
    python tools/mt_debug2.py [threads]"""
import os, sys, warnings
from multiprocessing.pool import ThreadPool
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), os.path.join(ROOT, 'tests'), ROOT]
warnings.simplefilter('ignore')
import numpy as np, scipy.sparse as sp
import osqp_amd, problems


def scen(seed):
    rng = np.random.default_rng(seed)
    P, q, A, l, u = problems.banded_qp(3000, window=60, seed=seed)
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=20000)
    r = m.solve()
    m.update(q=q * (1 + 0.01 * rng.standard_normal(len(q)))); r = m.solve()
    m.update(l=l - 0.05, u=u + 0.05); r = m.solve()
    m.warm_start(x=r.x * 0.9, y=r.y * 0.9); r = m.solve()
    Pt = sp.triu(P, format='csc')
    m.update(Px=Pt.data * (1 + 0.02 * rng.random(Pt.nnz)), Ax=A.data * (1 + 0.02 * rng.standard_normal(A.nnz)))
    pol = os.environ.get('MT_POLISH', '1') == '1'
    m.update_settings(polishing=pol); r = m.solve()
    return (r.info.iter, r.info.status_polish, r.info.prim_res, r.info.dual_res, r.info.obj_val, r.x.copy(), r.y.copy())


threads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
seeds = list(range(70, 76))
serial = [scen(s) for s in seeds]
for rnd in range(3):
    with ThreadPool(threads) as pool:
        thr = pool.map(scen, seeds)
    for a, b, s in zip(serial, thr, seeds):
        if not (np.array_equal(a[5], b[5]) and np.array_equal(a[6], b[6])):
            print('round %d seed %d: serial iter %d polish %d pri %.2e dua %.2e obj %.10e | threaded iter %d polish %d pri %.2e dua %.2e obj %.10e | dx %.2e dy %.2e'
                  % ((rnd, s) + a[:5] + b[:5] + (np.abs(a[5] - b[5]).max(), np.abs(a[6] - b[6]).max())), flush=True)
print('done')
