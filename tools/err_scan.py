import sys, warnings
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [os.path.join(R, 'osqp-python_amd'), os.path.join(R, 'oracle'), os.path.join(R, 'tests'), R]
warnings.simplefilter('ignore')
import numpy as np, osqp_amd, problems
from oracle import Oracle
from test_gpu_parity import GENS
for name, g in GENS.items():
    P,q,A,l,u = g()
    xo,yo,io = Oracle().setup(P,q,A,l,u, eps_abs=1e-10, eps_rel=1e-10, max_iter=200000, adaptive_rho_interval=50).solve()
    for eps in (1e-6, 1e-8):
        m = osqp_amd.OSQP(); m.setup(P,q,A,l,u, eps_abs=eps, eps_rel=eps, verbose=False, max_iter=100000)
        r = m.solve(); s = m._solver.hip_stats()
        print('%-17s eps=%.0e it=%5d st=%d |dx|=%.2e/%.2e |dy|=%.2e/%.2e dobj=%.2e cg/it=%.1f oracle_it=%d' % (name, eps, r.info.iter, r.info.status_val, abs(r.x-xo).max(), 1+abs(xo).max(), abs(r.y-yo).max(), 1+abs(yo).max(), abs(r.info.obj_val-io.obj_val)/(1+abs(io.obj_val)), s['pcg_iters_total']/r.info.iter, io.iter), flush=True)
