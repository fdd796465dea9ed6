import sys, warnings
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [os.path.join(R, 'osqp-python_amd'), os.path.join(R, 'oracle'), os.path.join(R, 'tests'), R]
warnings.simplefilter('ignore')
import numpy as np, osqp_amd, problems
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
P,q,A,l,u = problems.banded_qp(n)
kw = dict(eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, check_termination=25, adaptive_rho_interval=50, verbose=True)
kw.update(eval(sys.argv[2]) if len(sys.argv) > 2 else {})
m = osqp_amd.OSQP(); m.setup(P,q,A,l,u, **kw)
r = m.solve(); print(m._solver.hip_stats())
