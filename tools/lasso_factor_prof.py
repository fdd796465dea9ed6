"""One setup + one cold solve of the lasso (BASELINE configs[2]) for a kernel-level profile of the Woodbury factorisation (dense_hip.hip) and of the fused
iteration:  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -o lasso -- python $REPO/tools/lasso_factor_prof.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'osqp-python_amd')):
    sys.path.insert(0, p)
import osqp_amd, problems
P, q, A, l, u = problems.lasso_qp(5000, 10000)
m = osqp_amd.OSQP(algebra='hip')
t = time.time(); m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=50000, check_termination=25, adaptive_rho_interval=50, warm_starting=False); print('setup %.2f s' % (time.time() - t))
t = time.time(); r = m.solve(); print('solve %.1f ms, %d iterations, %s' % (1e3 * (time.time() - t), r.info.iter, r.info.status))
print(m._solver.hip_stats()['woodbury_factor_ms'], m._solver.hip_preconditioner())
