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

    # ---- is the tail (slowest workgroup vs the median) IMBALANCE of the blocks' work or skew of their start / of the memory system?  (round-5 verdict, item 2c)
    ent, ex = (tr[:, 0] - t0) * 0.01, (tr[:, 9] - t0) * 0.01
    dur = ex - ent
    wg = np.arange(1024)
    def pc(v): return 'min %.2f p10 %.2f med %.2f p90 %.2f max %.2f' % (v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max())
    print('  workgroup duration (exit - entry): ' + pc(dur))
    print('  corr(entry, exit) %.2f   corr(entry, duration) %.2f   corr(blockIdx, entry) %.2f   corr(blockIdx, duration) %.2f' %
          (np.corrcoef(ent, ex)[0, 1], np.corrcoef(ent, dur)[0, 1], np.corrcoef(wg, ent)[0, 1], np.corrcoef(wg, dur)[0, 1]))
    print('  by XCD (blockIdx & 7): median entry ' + ' '.join('%.2f' % np.median(ent[wg % 8 == x]) for x in range(8)) + '   median duration ' + ' '.join('%.2f' % np.median(dur[wg % 8 == x]) for x in range(8)))
    slow = np.argsort(ex)[-32:]
    print('  the 32 last workgroups to exit: entry med %.2f (all: %.2f), duration med %.2f (all: %.2f); phase durations of theirs: ' % (np.median(ent[slow]), np.median(ent), np.median(dur[slow]), np.median(dur)) +
          ' '.join('%.2f' % np.median(d[slow, i]) for i in range(9)))
    if rep == 0:
        first = dur.copy()
    else:
        print('  corr(duration in repetition 0, duration now) %.2f  (1 = the same workgroups are slow every time: their WORK; 0 = whoever the memory system served last)' % np.corrcoef(first, dur)[0, 1])
