"""Device-memory leak check: repeated setup / solve / update / batch / linsys / cleanup cycles must not grow the allocation."""
import gc, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT]
warnings.simplefilter('ignore')
import numpy as np, torch, scipy.sparse as sp
import osqp_amd, problems
from osqp_amd.linsys import LinSysSolver

def used():
    torch.cuda.synchronize(); free, total = torch.cuda.mem_get_info(); return (total - free) / 2**20

P, q, A, l, u = problems.banded_qp(20000)
Pm, qm, Am, L, U = problems.mpc_batch(64)
Pl, ql, Al, ll, ul = problems.lasso_qp(301, 650)      # (device-factorised Woodbury correction: dense system, cached inverses, the inverse's auxiliary stream)
base = None
for cyc in range(6):
    for rep in range(10):
        m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, max_iter=200); m.solve(); m.update(q=q * 1.01); m.update(Px=sp.triu(P).tocsc().data, Ax=A.data); m.solve()
        del m
        s = osqp_amd.OSQP(); s.setup(Pm, qm, Am, L[0], U[0], verbose=False); s._solver.hip_batch_solve(l=L, u=U); del s
        w = osqp_amd.OSQP(); w.setup(Pl, ql, Al, ll, ul, verbose=False, max_iter=400); w.solve(); w.update(q=ql * 1.01); w.solve(); del w
        ls = LinSysSolver(sp.csc_matrix(P), sp.csc_matrix(A), np.full(A.shape[0], 0.1), polishing=True, cg_max_iter=200); ls.solve(np.ones(P.shape[0] + A.shape[0])); ls.free(); del ls
    gc.collect()
    now = used()
    if base is None: base = now
    print('cycle %d: %.1f MiB in use (delta vs first cycle %+.1f)' % (cyc, now, now - base), flush=True)
assert now - base < 16, 'device memory grows'
print('no growth')
