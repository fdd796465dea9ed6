"""Where does the one-launch PCG iteration (k_slot1 phase F, probed as k_f1_probe) spend its time?  Runs ON AN MI355X against the
diagnostic build (make -C osqp-python_amd trace): lane 0 of every workgroup stamps the 100 MHz clock at the phase boundaries.

    python tools/ktrace_f1.py [n]"""
import os
import subprocess
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'osqp-python_amd')
LIB = os.path.join(PKG, 'osqp_amd', 'libosqp_hip_trace.so')
if not os.path.exists(LIB):
    subprocess.check_call(['make', '-C', PKG, '-s', 'trace'])
os.environ['OSQP_HIP_LIBRARY'] = LIB
sys.path[:0] = [PKG, ROOT]
warnings.simplefilter('ignore')

import numpy as np  # noqa: E402
import osqp_amd  # noqa: E402
import problems  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
P, q, A, l, u = problems.banded_qp(n)
m = osqp_amd.OSQP()
m.setup(P, q, A, l, u, verbose=False)
m.update_settings(max_iter=60)
m.solve()
s = m._solver
NAMES = ['entry', 'scalars ready', 'descriptors arrived', 'loads issued', 'window staged (+ own update)', 'products staged', 'row sums done',
         'transposed products staged', 'column sums + stores done', 'exit']
for rep in range(3):
    ms = 0.5 * s.hip_time_kernel(15, 20)         # (the probe is two consecutive launches; the stamps are those of the last one)
    tr = s.hip_trace_read().reshape(1024, 16).astype(np.int64)
    t0 = tr[:, 0].min()
    print('--- repetition %d: launch time by hipEvent %.2f us' % (rep, ms * 1e3))
    for p, name in enumerate(NAMES):
        v = (tr[:, p] - t0) * 0.01
        print('  %-32s min %6.2f  med %6.2f  p90 %6.2f  max %6.2f' % (name, v.min(), np.median(v), np.percentile(v, 90), v.max()))
    d = np.diff(tr[:, :10], axis=1) * 0.01
    print('  phase durations (median per workgroup): ' + ' '.join('%.2f' % np.median(d[:, i]) for i in range(9)))
