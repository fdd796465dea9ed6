"""Device-driven mode at config 2: the host's boundary-group log of one warm-graph solve (which groups came too early / too late?)."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT]
warnings.simplefilter('ignore')
import osqp_amd, problems
P, q, A, l, u = problems.banded_qp(100000)
m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, verbose=False, warm_starting=False)
m.solve(); m.update_settings(rho=0.1); m.solve(); m.update_settings(rho=0.1)
m._solver.set_policy(slot_log=1)
sys.stderr.write('=== logged solve\n'); sys.stderr.flush()
r = m.solve()
s = m._solver.hip_stats()
print('iterations %d pcg %d launches %d graph launches %d topups %d' % (r.info.iter, s['pcg_iters_total'], s['kernel_launches'], s['graph_launches'], s['slot_topups']))
print('productive slot launches %d' % (s['pcg_iters_total'] + 3 * r.info.iter))
