"""Round-4 soak: repeated cold solves of the NEW paths must be bit-identical -- the two-launch Woodbury direct mode (portfolio), a reordered
handle (shuffled band, one-launch PCG form on the permuted problem), the mid-size F1 blocking (n = 50k), F1 with many blocks per workgroup
(n = 1M).    python tools/soak_r04.py [seconds]"""
import os, sys, time, warnings, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT]
warnings.simplefilter('ignore')
import numpy as np, osqp_amd, problems

def digest(r):
    return hashlib.sha1(r.x.tobytes() + r.y.tobytes()).hexdigest()[:12], r.info.iter

def shuffled(n):
    P, q, A, l, u = problems.banded_qp(n)
    rng = np.random.default_rng(7); pc, pr = rng.permutation(n), rng.permutation(2 * n)
    P = P[pc][:, pc].tocsc(); A = A[pr][:, pc].tocsc(); P.sort_indices(); A.sort_indices()
    return P, q[pc], A, l[pr], u[pr]

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
cases = (('portfolio 2000x50 (two-launch direct mode)', lambda: problems.portfolio_qp(2000, 50)), ('shuffled band n=100k (reordered)', lambda: shuffled(100000)),
         ('banded n=50k (full row blocks)', lambda: problems.banded_qp(50000)), ('banded n=1M (ten blocks per workgroup)', lambda: problems.banded_qp(1000000)))
bad = 0
for name, gen in cases:
    P, q, A, l, u = gen()
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, eps_abs=1e-6, eps_rel=1e-6, max_iter=50000, warm_starting=False)
    st = m._solver.hip_stats()
    ref = digest(m.solve()); t0 = time.time(); k = 0; diff = 0
    while time.time() - t0 < budget / len(cases):
        m.update_settings(rho=0.1)
        d = digest(m.solve()); k += 1
        diff += d != ref
    bad += diff
    print('%-44s %4d repetitions, %d differing; digest %s, %d iterations; pcg_fused %d reordered %d woodbury_direct %d' % (name, k, diff, ref[0], ref[1], st['pcg_fused'], st['reordered'], st['woodbury_direct']))
print('soak: %d differing repetitions' % bad)
sys.exit(1 if bad else 0)
