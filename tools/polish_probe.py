"""Accuracy and device time of the in-kernel polish (direct path) against the oracle's polish step, over the polish test cases and a few
(delta, polish_refine_iter) settings.  Runs ON AN MI355X:  python tools/polish_probe.py"""
import os
import sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, d) for d in ('tests', 'oracle', '', 'osqp-python_amd')]
warnings.simplefilter('ignore')
import numpy as np
import test_gpu_polish as T
for case in sorted(T.CASES):
    for (d, rf) in ((1e-6, 3), (1e-6, 0), (1e-3, 2)):
        m, r, (xp, yp, ip, sp_), _ = T.both(T.CASES[case], delta=d, refine=rf)
        print('%-22s d=%g rf=%d  sp %d/%d  pri %.2e/%.2e dua %.2e/%.2e  dx %.2e dy %.2e  t %.1f us' % (case, d, rf, r.info.status_polish, sp_, r.info.prim_res, ip.pri_res, r.info.dual_res, ip.dua_res,
              np.abs(r.x - xp).max(), np.abs(r.y - yp).max() if len(yp) else 0, 1e6 * r.info.polish_time))
