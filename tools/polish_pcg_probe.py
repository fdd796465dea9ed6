"""Polish on the PCG path vs the oracle's polish: which tolerances give both sides the same active set, and how far apart the polished points are."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT, os.path.join(ROOT, 'oracle')]
warnings.simplefilter('ignore')
import numpy as np
import osqp_amd, problems
from oracle import Oracle
rel = lambda a, b: np.abs(a - b).max() / (1 + np.abs(b).max())
for n, w in ((2000, 40),):
    P, q, A, l, u = problems.banded_qp(n, window=w)
    o8 = Oracle().setup(P, q, A, l, u, eps_abs=1e-10, eps_rel=1e-10, max_iter=200000, adaptive_rho_interval=50); x8, y8, i8 = o8.solve()
    for eps in (1e-3, 1e-4):
        st = dict(eps_abs=eps, eps_rel=eps, max_iter=20000, adaptive_rho_interval=50, check_termination=25)
        m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, polishing=True, **st)
        r = m.solve()
        o = Oracle().setup(P, q, A, l, u, **st); xo, yo, io = o.solve()
        xp, yp, ip, sp_ = o.polish(delta=1e-6, polish_refine_iter=3)
        print(os.environ.get('TAG', ''), 'n=%d eps=%g: engine %d it polish %d (%.1f ms) res %.1e/%.1e | oracle %d it polish %d res %.1e/%.1e | engine-vs-oraclepolish dx %.1e dy %.1e | engine-vs-exact dx %.1e dy %.1e | oraclepolish-vs-exact dx %.1e dy %.1e'
              % (n, eps, r.info.iter, r.info.status_polish, 1e3 * r.info.polish_time, r.info.prim_res, r.info.dual_res, io.iter, sp_, ip.pri_res, ip.dua_res,
                 rel(r.x, xp), rel(r.y, yp), rel(r.x, x8), rel(r.y, y8), rel(xp, x8), rel(yp, y8)), flush=True)
