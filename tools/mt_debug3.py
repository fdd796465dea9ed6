import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), os.path.join(ROOT, 'tests'), ROOT]
warnings.simplefilter('ignore')
import numpy as np, scipy.sparse as sp
import osqp_amd, problems
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 72
rng = np.random.default_rng(seed)
P, q, A, l, u = problems.banded_qp(3000, window=60, seed=seed)
m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=20000)
r = m.solve()
m.update(q=q * (1 + 0.01 * rng.standard_normal(len(q)))); r = m.solve()
m.update(l=l - 0.05, u=u + 0.05); r = m.solve()
m.warm_start(x=r.x * 0.9, y=r.y * 0.9); r = m.solve()
Pt = sp.triu(P, format='csc')
m.update(Px=Pt.data * (1 + 0.02 * rng.random(Pt.nnz)), Ax=A.data * (1 + 0.02 * rng.standard_normal(A.nnz)))
m.update_settings(polishing=True, verbose=True)
os.environ['OSQP_HIP_SLOT_LOG'] = '1'
try:
    r = m.solve(raise_error=True)
    print('status', r.info.status, 'polish', r.info.status_polish, r.info.prim_res, r.info.dual_res)
except Exception as e:
    print('EXC', type(e), e)
print(m._solver.hip_stats() if hasattr(m._solver, 'hip_stats') else '')
