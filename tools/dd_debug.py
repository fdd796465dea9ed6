"""Device-driven vs host-synchronous chunk boundaries on config 2: first cold solve, iteration counts, launches."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT]
warnings.simplefilter('ignore')
import numpy as np
import osqp_amd, problems
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
P, q, A, l, u = problems.banded_qp(n)
for dd in (0, 1):
    os.environ['OSQP_HIP_DEVICE_DRIVEN'] = str(dd)
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, check_termination=25, adaptive_rho_interval=50, verbose=False, warm_starting=False)
    for rep in range(3):
        t = time.perf_counter(); r = m.solve(); dt = time.perf_counter() - t
        s = m._solver.hip_stats()
        print('dd=%d solve %d: %.2f ms, %d iterations, %d rho updates, pcg total %d (%.2f/it) max %d unconv %d, launches %d, graph launches %d, obj %.9e'
              % (dd, rep, 1e3 * dt, r.info.iter, r.info.rho_updates, s['pcg_iters_total'], s['pcg_iters_total'] / r.info.iter, s['pcg_iters_max'], s['pcg_unconverged'], s['kernel_launches'], s['graph_launches'], r.info.obj_val))
        sys.stdout.flush()
