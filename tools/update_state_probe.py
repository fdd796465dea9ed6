import os, sys, warnings
sys.path[:0] = ['osqp-python_amd', '.', 'tests']
warnings.simplefilter('ignore')
import numpy as np, scipy.sparse as sp
import osqp_amd, problems
P, q, A, l, u = problems.banded_qp(20000, window=40)
rng = np.random.default_rng(7)
Pt = sp.triu(P, format='csc')
Px = Pt.data * (1 + 0.05 * rng.random(Pt.nnz)); Ax = A.data * (1 + 0.05 * rng.standard_normal(A.nnz))
P2 = sp.csc_matrix((Px, Pt.indices, Pt.indptr), shape=P.shape); A2 = sp.csc_matrix((Ax, A.indices, A.indptr), shape=A.shape)
kw = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=20000)
f = osqp_amd.OSQP(); f.setup(P2, q, A2, l, u, **kw); r = f.solve(); print('fresh', r.info.iter, r.info.rho_updates, r.info.obj_val)
r = f.solve(); print('fresh 2nd solve (warm)', r.info.iter)
f.update_settings(warm_starting=False, rho=0.1); r = f.solve(); print('fresh 3rd solve (cold, rho reset)', r.info.iter, r.info.rho_updates)
b = osqp_amd.OSQP(); b.setup(P, q, A, l, u, **kw); b.update(Px=Px, Ax=Ax); r = b.solve(); print('setup+update (no solve before)', r.info.iter, r.info.rho_updates)
c = osqp_amd.OSQP(); c.setup(P, q, A, l, u, **kw); c.solve(); c.update(Px=Px, Ax=Ax); c.update_settings(warm_starting=False, rho=0.1); r = c.solve(); print('setup+solve+update+cold', r.info.iter, r.info.rho_updates)
d = osqp_amd.OSQP(); d.setup(P, q, A, l, u, **kw); d.solve(); d.update_settings(warm_starting=False, rho=0.1); d.update(Px=Px, Ax=Ax); r = d.solve(); print('setup+solve+rho reset+update+cold', r.info.iter, r.info.rho_updates)
for name in ('OSQP_HIP_F1',):
    os.environ[name] = '0'
    c = osqp_amd.OSQP(); c.setup(P, q, A, l, u, **kw); c.solve(); c.update(Px=Px, Ax=Ax); c.update_settings(warm_starting=False, rho=0.1); r = c.solve(); print('F1=0: setup+solve+update+cold', r.info.iter)
    f = osqp_amd.OSQP(); f.setup(P2, q, A2, l, u, **kw); r = f.solve(); print('F1=0: fresh', r.info.iter)
