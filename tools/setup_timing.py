import os, sys, time, warnings
sys.path[:0] = ['osqp-python_amd', '.']
warnings.simplefilter('ignore')
import osqp_amd, problems
P, q, A, l, u = problems.banded_qp(100000)
for rep in range(3):
    os.environ['OSQP_HIP_SETUP_TIMING'] = '1' if rep == 2 else '0'
    if rep != 2: os.environ.pop('OSQP_HIP_SETUP_TIMING')
    t = time.time(); m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False); print('setup %d: %.1f ms' % (rep, 1e3 * (time.time() - t)), flush=True)
