"""Repeatability of the device-driven mode: several handles / launch modes on one problem; iteration and PCG counts must agree."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT]
warnings.simplefilter('ignore')
import numpy as np
import osqp_amd, problems
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 40
P, q, A, l, u = problems.banded_qp(n, window=w)
ref = None
for dd, graph in ((0, 1), (1, 1), (1, 0), (1, 1), (1, 0), (0, 0), (1, 1)):
    os.environ['OSQP_HIP_DEVICE_DRIVEN'] = str(dd); os.environ['OSQP_HIP_GRAPH'] = str(graph)
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, check_termination=25, adaptive_rho_interval=50, verbose=False, warm_starting=False)
    for rep in range(2):
        r = m.solve(); s = m._solver.hip_stats()
        key = (r.info.iter, int(s['pcg_iters_total']), r.info.rho_updates)
        if ref is None: ref = (key, r.x.copy())
        print('dd=%d graph=%d solve %d: iters %d pcg %d rho_updates %d launches %d topups %d  same-as-first-solve-of-first-handle: %s bitwise-x %s'
              % (dd, graph, rep, key[0], key[1], key[2], s['kernel_launches'], s['slot_topups'], key == ref[0], np.array_equal(r.x, ref[1])))
        sys.stdout.flush()
