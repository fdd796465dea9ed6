"""Probe of the wave-per-problem batch kernel: solve B MPC QPs with batch_wave = 1 / -1 and print iteration counts, differences and wall time.
usage: python tools/batch_wave_probe.py [B] [max_iter]"""
import sys, time
sys.path.insert(0, 'osqp-python_amd'); sys.path.insert(0, '.')
import numpy as np
import osqp_amd, problems

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mi = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
P, q, A, L, U = problems.mpc_batch(B)
out = {}
for w in (-1, 1):
    s = osqp_amd.OSQP(); s.setup(P, q, A, L[0], U[0], eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=mi)
    s._solver.set_policy(batch_wave=w)
    print('wave', w, 'launching', flush=True)
    t0 = time.time(); x, y, r = s._solver.hip_batch_solve(l=L, u=U); t1 = time.time()
    t2 = time.time(); x, y, r = s._solver.hip_batch_solve(l=L, u=U); t3 = time.time()
    print('wave', w, 'status', np.unique(r[:, 0]), 'iters mean %.1f max %d' % (r[:, 1].mean(), r[:, 1].max()), 'first %.1f ms second %.2f ms' % (1e3 * (t1 - t0), 1e3 * (t3 - t2)), flush=True)
    out[w] = (x, y, r)
xs, ys, rs = out[-1]; xw, yw, rw = out[1]
print('iters equal', np.array_equal(rs[:, 1], rw[:, 1]), 'dx %.2e dy %.2e' % (np.abs(xs - xw).max(), np.abs(ys - yw).max()), 'iter diff', np.flatnonzero(rs[:, 1] != rw[:, 1])[:10])
import os
if os.environ.get('OSQP_HIP_LIBRARY', '').endswith('trace.so'):          # (make TRACE=1 build: rec[7..11] hold 100 MHz ticks per phase)
    it = rw[:, 1]
    print('ticks (100 MHz) per iteration: rhs+A\' %.2f us, solve %.2f us, A+update %.2f us, residuals per call %.2f us, total per iteration %.2f us' % (
        (rw[:, 7] / it).mean() / 100, (rw[:, 8] / it).mean() / 100, (rw[:, 9] / it).mean() / 100, (rw[:, 11] / np.maximum(1, it // 25)).mean() / 100, (rw[:, 10] / it).mean() / 100))
