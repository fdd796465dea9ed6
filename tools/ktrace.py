"""Where does a PCG kernel's time go?  Runs ON AN MI355X against the diagnostic build of the engine (make -C osqp-python_amd
trace), in which lane 0 of every workgroup stamps the 100 MHz wall clock at phase boundaries of k_k2f / k_k1f.

    python tools/ktrace.py [n]

Prints, per kernel, the distribution over the 1024 workgroups of each stamp relative to the kernel's first workgroup entry
(min / median / p90 / max, in microseconds), and the gap between the two kernels."""
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
NAMES = {0: 'entry', 1: 'flag+descriptor arrived', 2: 'matrix loads issued', 3: 'hook done (partials folded%s)',
         4: 'gathers done, products staged', 5: 'row sums + epilogue done', 6: 'exit (block reduce + partial store)',
         7: 'row sums done (late hook starts)'}


def report(tr, base, label, extra):
    t0 = tr[:, base].min()
    print('%s   (first workgroup entry = 0)' % label)
    for p in (0, 1, 2, 4, 7, 3, 5, 6):
        v = (tr[:, base + p].astype(np.int64) - int(t0)) * 0.01        # 100 MHz -> us
        v = v[tr[:, base + p] > 0]
        if len(v) == 0:
            continue
        name = NAMES[p] % extra if '%s' in NAMES[p] else NAMES[p]
        print('  %-46s min %6.2f  med %6.2f  p90 %6.2f  max %6.2f   (%d wgs)' % (name, v.min(), np.median(v), np.percentile(v, 90), v.max(), len(v)))
    return t0


for rep in range(3):
    ms = s.hip_time_kernel(10, 20)
    tr = s.hip_trace_read().reshape(1024, 16)
    print('--- repetition %d: pair time by hipEvent %.2f us' % (rep, ms * 1e3))
    a = report(tr, 0, 'k_k2f', '')
    b = report(tr, 8, 'k_k1f', ', alpha, vector update')
    for base, nm in ((0, 'k_k2f'), (8, 'k_k1f')):
        t0 = int(tr[:, base].min()); ex = (tr[:, base + 6].astype(np.int64) - t0) * 0.01; en = (tr[:, base].astype(np.int64) - t0) * 0.01
        wg = np.arange(1024)
        print('  %s exit by XCD (wg %% 8): ' % nm + ' '.join('%d:%.2f/%.2f' % (x, np.median(ex[wg % 8 == x]), ex[wg % 8 == x].max()) for x in range(8)) + '   (median/max us)')
        print('  %s exit by slot quarter (wg // 8 in 0-31, 32-63, 64-95, 96-127): ' % nm + ' '.join('%.2f/%.2f' % (np.median(ex[(wg // 8) // 32 == k]), ex[(wg // 8) // 32 == k].max()) for k in range(4)))
        print('  %s entry by slot quarter: ' % nm + ' '.join('%.2f' % np.median(en[(wg // 8) // 32 == k]) for k in range(4)) + '; slowest 20 wgs: ' + ' '.join(str(i) for i in np.argsort(-ex)[:20]))
    print('  k_k1f first entry - k_k2f last exit: %.2f us;  k_k2f entry -> k_k1f last exit: %.2f us'
          % ((int(b) - int(tr[:, 6].max())) * 0.01, (int(tr[:, 14].max()) - int(a)) * 0.01))
