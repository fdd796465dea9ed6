"""Wall time of the first, second and third solve of one handle (graph capture happens in the first), graphs on/off."""
import os, sys, time, warnings
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [os.path.join(R, 'osqp-python_amd'), R]
warnings.simplefilter('ignore')
import osqp_amd, problems
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
P, q, A, l, u = problems.banded_qp(n)
w = osqp_amd.OSQP(); w.setup(P, q, A, l, u, verbose=False, max_iter=50); w.solve()     # runtime / module load warm-up
for g in ('1', '0'):
    os.environ['OSQP_HIP_GRAPH'] = g
    m = osqp_amd.OSQP(); t0 = time.perf_counter()
    m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, warm_starting=False, verbose=False); ts = time.perf_counter() - t0
    out = []
    for k in range(3):
        t0 = time.perf_counter(); r = m.solve(); out.append((time.perf_counter() - t0) * 1e3)
    print('graph=%s setup %.1f ms; solves: %s ms; iter %d; graph launches %d' % (g, ts * 1e3, ['%.1f' % t for t in out], r.info.iter, m._solver.hip_stats()['graph_launches']))
