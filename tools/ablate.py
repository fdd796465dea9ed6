"""Ablation timing of the PCG kernels.  Builds made with -DOSQP_HIP_KNOCK=<mask> (ab/libosqp_hip_k<mask>.so) drop phases of
k_k2f (hook, window gather, value loads, row sums, epilogue stores, block reduction ...): their RESULTS ARE WRONG, only the
probe timings mean anything.  Runs on one GPU box:

    python tools/ablate.py 0 1 2 4 ...

Prints per mask the best-of-3 probe times (us) of k_k2f alone (12), the PCG pair (10) and k_k1f alone (11)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, warnings; sys.path[:0] = ["osqp-python_amd", "."]; warnings.simplefilter("ignore")
import osqp_amd, problems
P, q, A, l, u = problems.banded_qp(100000)
m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False); m.update_settings(max_iter=60); m.solve()
s = m._solver
print(' '.join('%d:%.2f' % (w, min(s.hip_time_kernel(w, 300) for _ in range(3)) * 1e3) for w in (12, 10, 11)), 'us')
'''

for k in sys.argv[1:]:
    lib = os.path.join(ROOT, 'ab', 'libosqp_hip_k%s.so' % k)
    env = dict(os.environ, OSQP_HIP_LIBRARY=lib)
    out = subprocess.run([sys.executable, '-c', CHILD], cwd=ROOT, env=env, capture_output=True, text=True)
    print('knock %4s: %s' % (k, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]), flush=True)
