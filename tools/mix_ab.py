"""A/B of the mixed config's solve time inside ONE process-per-variant loop (gpu_solve_ms, median of 7 cold solves)."""
import sys, os, subprocess, json
code = r'''
import sys, os, warnings
warnings.simplefilter("ignore")
sys.path[:0] = [".", "osqp-python_amd"]
import numpy as np, osqp_amd, problems
P, q, A, l, u = problems.banded_qp(100000, window=200, long_range=float(sys.argv[1]))
m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=20000, adaptive_rho_interval=50, check_termination=25, warm_starting=False)
ts = []
for i in range(9):
    r = m.solve(); ts.append(m._solver.hip_stats()['gpu_solve_ms'])
s = m._solver.hip_stats()
print("%s iters %d form %d far %d  solve ms median %.2f min %.2f" % (sys.argv[2], r.info.iter, s['pcg_fused'], s['f1_far_columns'], sorted(ts[2:])[3], min(ts[2:])))
'''
fr = sys.argv[1] if len(sys.argv) > 1 else '0.02'
for rep in range(2):
    for name, env in (('mixing', {}), ('two-kernel', {'OSQP_HIP_F1': '2'})):
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, '-c', code, fr, name], env=e)
