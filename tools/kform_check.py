"""K form (backend.h DevKf: one launch per PCG iteration on the explicit reduced matrix) against the two-kernel form on the same unstructured
matrix: results, iteration counts, launches, cold-solve times, and the slot kernel's own F launch (time_kernel 16)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'osqp-python_amd')):
    sys.path.insert(0, p)
import osqp_amd, problems


def solve(P, q, A, l, u, kform, reps=3, eps=1e-6, **kw):
    os.environ['OSQP_HIP_KFORM'] = str(kform)
    m = osqp_amd.OSQP(algebra='hip')
    m.setup(P, q, A, l, u, eps_abs=eps, eps_rel=eps, verbose=False, max_iter=20000, cg_max_iter=50, adaptive_rho_interval=50, check_termination=25, warm_starting=False, **kw)
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); r = m.solve(); ts.append(1e3 * (time.perf_counter() - t))
    st = m._solver.hip_stats()
    return m, r, st, ts


def rel(a, b):
    return float(np.abs(a - b).max() / (1 + np.abs(b).max()))


out = {}
for n, window in [(2000, 2000), (20000, 20000), (100000, 100000)] + ([(100000, 200)] if '--banded' in sys.argv else []):
    P, q, A, l, u = problems.banded_qp(n, window=window)
    m1, r1, s1, t1 = solve(P, q, A, l, u, 1)
    m0, r0, s0, t0 = solve(P, q, A, l, u, 0)
    f1 = 0.5 * m1._solver.hip_time_kernel(16, 200) if int(s1['pcg_fused']) in (2, 3) else 0.0
    pair0 = m0._solver.hip_time_kernel(10, 200)
    row = dict(n=n, window=window, form1=int(s1['pcg_fused']), form0=int(s0['pcg_fused']), nnzK=s1.get('kform_nnz', 0), iter1=r1.info.iter, iter0=r0.info.iter,
               pcg1=s1['pcg_iters_total'] / max(r1.info.iter, 1), pcg0=s0['pcg_iters_total'] / max(r0.info.iter, 1), launches1=s1['kernel_launches'], launches0=s0['kernel_launches'],
               ms1=t1, ms0=t0, gpu_ms1=s1['gpu_solve_ms'], gpu_ms0=s0['gpu_solve_ms'], dx=rel(r1.x, r0.x), dy=rel(r1.y, r0.y), status1=r1.info.status, status0=r0.info.status,
               f_launch_us=1e3 * f1, two_kernel_pair_us=1e3 * pair0)
    print(json.dumps(row)); sys.stdout.flush()
    out['n%d_w%d' % (n, window)] = row
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'kform_check.json'), 'w'), indent=1)
