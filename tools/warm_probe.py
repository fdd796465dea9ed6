"""Warm re-solves after a 1 % data change: iterations of the oracle (direct solves) vs the engine -- on the CPU through the host simulator
(same driver + policy.h, plain-loop device ops), or on the GPU with --gpu.    python tools/warm_probe.py [--gpu] [n] [window]"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
warnings.simplefilter('ignore')
import contextlib
import numpy as np
import osqp_amd, problems
from oracle import Oracle
args = [a for a in sys.argv[1:] if a != '--gpu']
gpu = '--gpu' in sys.argv
n = int(args[0]) if args else 2000
w = int(args[1]) if len(args) > 1 else 40
if gpu:
    ctx = contextlib.nullcontext()
else:
    from hostsim_util import hostsim
    ctx = hostsim()
gens = {'banded': lambda: problems.banded_qp(n, window=w)}
if n <= 5000:
    gens['mpc'] = lambda: (lambda P, q, A, L, U: (P, q, A, L[0], U[0]))(*problems.mpc_batch(1, seed=3))
with ctx:
    for name, gen in gens.items():
        P, q, A, l, u = gen()
        st = dict(eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, check_termination=25, adaptive_rho_interval=50)
        o = Oracle().setup(P, q, A, l, u, **st)
        _, _, io = o.solve()
        m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, warm_starting=True, **st)
        if name == 'mpc':
            m._solver.set_policy(small_direct=0)
        r = m.solve()
        print('%s n=%d: cold  oracle %d it (rho updates %d) | engine %d it (rho updates %d)' % (name, len(q), io.iter, io.rho_updates, r.info.iter, r.info.rho_updates))
        rng = np.random.default_rng(0)
        for rep in range(6):
            if rep % 2 == 0:
                q2 = q * (1 + 0.01 * rng.standard_normal(len(q))); o.update(q=q2); m.update(q=q2); what = 'q'
            else:
                d = 0.01 * rng.random(len(l)); l2, u2 = l - d, u + d; o.update(l=l2, u=u2); m.update(l=l2, u=u2); what = 'bounds'
            _, _, io = o.solve(); r = m.solve()
            print('   warm after 1%% %-6s: oracle %4d it (rho updates %d) | engine %4d it (rho updates %d, %s)' % (what, io.iter, io.rho_updates, r.info.iter, r.info.rho_updates, r.info.status))
