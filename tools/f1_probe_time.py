"""Time of one F launch / KA launch of the one-launch form (osqp_hip_time_kernel 15, 17) on configs[1] with a fraction of long-range entries.
   python tools/f1_probe_time.py [n] [window] [long_range]        (OSQP_HIP_F1=2: strict windows; 3: the mixing kernels regardless)"""
import sys, os, warnings
warnings.simplefilter("ignore")
sys.path[:0] = [".", "osqp-python_amd"]
import numpy as np, osqp_amd, problems
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
P, q, A, l, u = problems.banded_qp(n, window=w, long_range=fr)
m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=20000, adaptive_rho_interval=50, check_termination=25)
r = m.solve(); s = m._solver.hip_stats()
t = {k: m._solver.hip_time_kernel(k, 200) for k in (15, 17)} if s['pcg_fused'] == 2 else {}
print("long_range %.3f F1=%s: %s %d iterations, form %d, D %d, far %d, solve %.2f ms; F launch %s us, KA launch %s us"
      % (fr, os.environ.get('OSQP_HIP_F1', '1'), r.info.status, r.info.iter, s['pcg_fused'], s['f1_replicas'], s['f1_far_columns'], s['gpu_solve_ms'],
         '%.2f' % (1e3 * t[15]) if t else '-', '%.2f' % (1e3 * t[17]) if t else '-'))
