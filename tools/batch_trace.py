"""Where the batch kernel's time goes (diagnostic build, make -C osqp-python_amd trace): per problem, microseconds in the
factorisations, in the substitutions and in total.   python tools/batch_trace.py [nbatch]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P, q, A, L, U = problems.mpc_batch(B)
s = osqp_amd.OSQP(); s.setup(P, q, A, L[0], U[0], eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=4000)
for rep in range(2):
    x, y, rec = s._solver.hip_batch_solve(l=L, u=U)
it = rec[:, 1]; tf, ts, ta = rec[:, 5] * 0.01, rec[:, 6] * 0.01, rec[:, 7] * 0.01
print('problems %d, solved %d; iterations mean %.1f max %d' % (B, (rec[:, 0] == 1).sum(), it.mean(), it.max()))
print('per problem (us): total mean %.0f max %.0f | factorisations mean %.0f | substitutions mean %.0f = %.2f per iteration | rest %.2f per iteration'
      % (ta.mean(), ta.max(), tf.mean(), ts.mean(), (ts / it).mean(), ((ta - tf - ts) / it).mean()))
trhs, tupd, tfwd, tres = rec[:, 3] * 0.01, rec[:, 4] * 0.01, rec[:, 8] * 0.01, rec[:, 9] * 0.01
print('per iteration (us): rhs (t, B product) %.2f | solve %.2f (forward substitution %.2f) | A product + z, y, x update %.2f | residuals %.2f | unaccounted %.2f'
      % ((trhs / it).mean(), (ts / it).mean(), (tfwd / it).mean(), (tupd / it).mean(), (tres / it).mean(), ((ta - tf - ts - trhs - tupd - tres) / it).mean()))
print('sum of per-problem totals / (256 CUs x 2 resident) = %.1f ms' % (ta.sum() / 512 / 1e3))
