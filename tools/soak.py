"""Soak / determinism check on an MI355X: many repeated cold solves must give bit-identical results (any data race in the
kernels' hand-offs, partial reductions or the fused PCG ping-pong would show up as a differing iterate or iteration count)."""
import os, sys, time, warnings, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT]
warnings.simplefilter('ignore')
import numpy as np, osqp_amd, problems

def digest(r):
    return hashlib.sha1(r.x.tobytes() + r.y.tobytes()).hexdigest()[:12], r.info.iter

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
for name, gen, reps in (('banded n=100k', lambda: problems.banded_qp(100000), 10**9), ('lasso 60x300 (long rows)', lambda: problems.lasso_qp(60, 300), 400),
                        ('mpc n=120 (one-launch direct path)', lambda: (lambda P, q, A, L, U: (P, q, A, L[0], U[0]))(*problems.mpc_batch(1)), 1500)):
    P, q, A, l, u = gen()
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, warm_starting=False)
    ref = digest(m.solve()); t0 = time.time(); k = 0
    while k < reps and time.time() - t0 < budget / 3:
        m.update_settings(rho=0.1)                 # the adapted rho stays with the handle (like the reference): start every repetition alike
        d = digest(m.solve()); k += 1
        assert d == ref, '%s: repetition %d differs: %s vs %s' % (name, k, d, ref)
    print('%-36s %5d repetitions bit-identical (%s, %d iterations)' % (name, k, ref[0], ref[1]), flush=True)
