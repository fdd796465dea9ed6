"""Which band widths / shapes give which replica count D of the F1 form (and does every D agree with the two-kernel form)?"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT]
warnings.simplefilter('ignore')
import numpy as np
import osqp_amd, problems
CASES = ((40000, 80000, 20, 3), (40000, 80000, 60, 5), (40000, 80000, 120, 5), (40000, 80000, 200, 5), (40000, 80000, 300, 5), (60000, 60000, 200, 5), (30000, 90000, 100, 4), (100000, 200000, 200, 5))
if len(sys.argv) > 1:
    CASES = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for n, m, window, k in CASES:
    P, q, A, l, u = problems.banded_qp(n, m=m, window=window, nnz_per_row=k)
    out = {}
    for f1 in ('1', '0'):
        os.environ['OSQP_HIP_F1'] = f1
        s = osqp_amd.OSQP(); s.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=20000)
        r = s.solve(); st = s._solver.hip_stats()
        out[f1] = (r, st)
    r1, s1 = out['1']; r0, s0 = out['0']
    print('n %6d m %6d window %3d nnz/row %d: D = %d  F1 %d it (%.2f pcg)  two-kernel %d it   |dx| %.2e |dy| %.2e' % (
        n, m, window, k, s1['f1_replicas'], r1.info.iter, s1['pcg_iters_total'] / r1.info.iter, r0.info.iter,
        np.abs(r1.x - r0.x).max() / (1 + np.abs(r0.x).max()), np.abs(r1.y - r0.y).max() / (1 + np.abs(r0.y).max())), flush=True)
