"""Repeated cold solves of the mixed config (per-block mixing form) are bit-identical: python tools/mix_soak.py [seconds]"""
import sys, time, warnings
warnings.simplefilter("ignore")
sys.path[:0] = [".", "osqp-python_amd"]
import numpy as np, osqp_amd, problems
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
P, q, A, l, u = problems.banded_qp(100000, long_range=0.02)
m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=20000, adaptive_rho_interval=50, check_termination=25, warm_starting=False)
rho0 = m.settings.rho
r0 = m.solve(); s = m._solver.hip_stats()
import os
assert os.environ.get('OSQP_HIP_F1') == '2' or (s['pcg_fused'] == 2 and s['f1_far_columns'] > 0)
t0 = time.time(); n = 0; bad = 0
while time.time() - t0 < secs:
    m.update_settings(rho=rho0)                        # (a true cold start: x, z, y = 0 by warm_starting = False, rho back to its initial value -- as bench.py does)
    r = m.solve(); n += 1
    same = r.info.iter == r0.info.iter and np.array_equal(r.x, r0.x) and np.array_equal(r.y, r0.y)
    if not same and bad < 3: print("differs: iter %d vs %d, max |dx| %.3e, |dy| %.3e" % (r.info.iter, r0.info.iter, np.abs(r.x - r0.x).max(), np.abs(r.y - r0.y).max()))
    bad += not same
print("mixing form, n = 100k, %d far columns: %d repeated cold solves (%d iterations each), %d differing from the first" % (s['f1_far_columns'], n, r0.info.iter, bad))
