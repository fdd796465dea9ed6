"""Large-rank Woodbury correction on the lasso QPs: Jacobi / corrected preconditioner inside the PCG / direct mode, against the oracle
at a size it factorises in seconds, then timings at BASELINE configs[2].    python tools/lasso_wb_probe.py [nf ns] [big]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT, os.path.join(ROOT, 'oracle')]
warnings.simplefilter('ignore')
import numpy as np
import osqp_amd, problems

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 500
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
with_oracle = os.environ.get('ORACLE', '1') == '1'
P, q, A, l, u = problems.lasso_qp(nf, ns)
ref = None
if with_oracle:
    from oracle import Oracle
    t = time.time(); xo, yo, io = Oracle().setup(P, q, A, l, u, eps_abs=1e-8, eps_rel=1e-8, max_iter=100000).solve()
    print('oracle: status %d, %d iterations, obj %.10e (%.1f s)' % (io.status_val, io.iter, io.obj_val, time.time() - t), flush=True)
    ref = (xo, yo, io)
for name, env in (('jacobi', {'OSQP_HIP_WOODBURY_LARGE': '0'}), ('woodbury-pcg', {'OSQP_HIP_WOODBURY_DIRECT': '0'}), ('woodbury-direct', {})):
    for k in ('OSQP_HIP_WOODBURY_LARGE', 'OSQP_HIP_WOODBURY_DIRECT'): os.environ.pop(k, None)
    os.environ.update(env)
    m = osqp_amd.OSQP()
    t = time.time(); m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=20000); ts = time.time() - t
    r = m.solve()                               # first solve: graph captures
    m.update_settings(warm_starting=False, rho=0.1)
    t = time.time(); r = m.solve(); tt = time.time() - t
    s = m._solver.hip_stats()
    line = '%-16s setup %.2f s, solve %.3f s: status %d, %d iterations, %.2f PCG each, %d rho updates, obj %.10e' % (
        name, ts, tt, r.info.status_val, r.info.iter, s['pcg_iters_total'] / max(1, r.info.iter), r.info.rho_updates, r.info.obj_val)
    if ref is not None:
        line += '  |dx| %.2e |dy| %.2e' % (np.abs(r.x - ref[0]).max() / (1 + np.abs(ref[0]).max()), np.abs(r.y - ref[1]).max() / (1 + np.abs(ref[1]).max()))
    print(line, flush=True)
