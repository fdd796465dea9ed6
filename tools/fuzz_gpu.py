"""Randomised agreement check on an MI355X: small QPs of varied shape (n 1..40, m 0..60, random sparsity, equality rows,
one-sided and free rows, rank-deficient or zero P) through (a) the single-QP engine and (b) the batch kernel (all linear-solve
variants that apply), against the oracle (direct LDL').   python tools/fuzz_gpu.py [count] [seed]"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT, os.path.join(ROOT, 'oracle')]
warnings.simplefilter('ignore')
import numpy as np, scipy.sparse as sp
import osqp_amd
from oracle import Oracle

count = int(sys.argv[1]) if len(sys.argv) > 1 else 150
NMIN = int(os.environ.get('FUZZ_NMIN', 1)); NMAX = int(os.environ.get('FUZZ_NMAX', 40))        # variable-count range
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
EPS = 1e-6
bad = 0
soft = 0
stats = {}
for t in range(count):
    n = int(rng.integers(NMIN, NMAX + 1)); m = int(rng.integers(0, int(1.5 * NMAX) + 1))
    dens = rng.choice([0.1, 0.3, 0.8])
    kind = rng.choice(['spd', 'psd_lowrank', 'zero', 'diag'])
    if kind == 'spd':
        M = sp.random(n, n, density=dens, random_state=rng, data_rvs=rng.standard_normal); P = (M @ M.T + 0.05 * sp.eye(n)).tocsc()
    elif kind == 'psd_lowrank':
        M = sp.random(n, max(1, n // 3), density=0.6, random_state=rng, data_rvs=rng.standard_normal); P = (M @ M.T).tocsc()
    elif kind == 'zero':
        P = sp.csc_matrix((n, n))
    else:
        P = sp.diags(rng.uniform(0, 2, n)).tocsc()
    q = rng.standard_normal(n)
    A = sp.random(m, n, density=dens, random_state=rng, data_rvs=rng.standard_normal, format='csc') if m else sp.csc_matrix((0, n))
    x0 = rng.standard_normal(n); ax = A @ x0
    l = ax - rng.uniform(0, 1, m); u = ax + rng.uniform(0, 1, m)
    r = rng.random(m)
    l[r < 0.15] = ax[r < 0.15]; u[r < 0.15] = ax[r < 0.15]              # equalities
    l[(r > 0.15) & (r < 0.3)] = -np.inf; u[(r > 0.3) & (r < 0.4)] = np.inf
    if kind in ('zero', 'psd_lowrank') and m and rng.random() < 0.7:      # keep most of them bounded
        A = sp.vstack([A, sp.eye(n)]).tocsc(); l = np.concatenate([l, x0 - 2]); u = np.concatenate([u, x0 + 2]); m += n
    st = dict(eps_abs=EPS, eps_rel=EPS, max_iter=20000, verbose=False)
    xo, yo, io = Oracle().setup(P, q, A, l, u, eps_abs=EPS, eps_rel=EPS, max_iter=20000, adaptive_rho_interval=50, check_termination=25).solve()
    os.environ.pop('OSQP_HIP_BATCH_VARIANT', None)
    os.environ['OSQP_HIP_SMALL_DIRECT'] = '1'
    s = osqp_amd.OSQP(); s.setup(P, q, A, l, u, **st)
    res = s.solve()                                                       # default: one launch of the direct batch kernel when it applies
    fast = s._solver.hip_stats()['kernel_launches'] == 1
    os.environ['OSQP_HIP_SMALL_DIRECT'] = '0'
    s2 = osqp_amd.OSQP(); s2.setup(P, q, A, l, u, **st)
    res2 = s2.solve()                                                     # the multi-kernel PCG engine on the same problem
    tag = '%s n=%d m=%d' % (kind, n, m)
    def agree(name, status, x, obj, iters):
        global bad, soft
        ok = status == io.status_val or {status, io.status_val} <= {3, 4, 5, 6}     # infeasibility flavours may differ when both hold
        if ok and status == 1:
            ok = abs(obj - io.obj_val) <= 2e-4 * (1 + abs(io.obj_val)) and (np.abs(x - xo).max() <= 5e-3 * (1 + np.abs(xo).max()) or kind != 'spd')
        if not ok:
            # one side ran into max_iter (ADMM on LPs / rank-deficient QPs is slow and its iteration count is sensitive to rounding):
            # not a disagreement about the answer as long as the side that stopped early is where it should be heading
            limit = 7 in (status, io.status_val) or 2 in (status, io.status_val)
            if limit:
                soft += 1
            else:
                bad += 1
            print('%s %s [%s]: status %d (%d it) vs oracle %d (%d it), obj %.6g vs %.6g' % ('iteration-limit' if limit else 'MISMATCH', name, tag, status, iters,
                                                                                            io.status_val, io.iter, obj, io.obj_val), flush=True)
        stats[name] = stats.get(name, 0) + 1
    agree('engine:direct' if fast else 'engine:pcg(default)', res.info.status_val, res.x, res.info.obj_val, res.info.iter)
    if fast:
        agree('engine:pcg(forced)', res2.info.status_val, res2.x, res2.info.obj_val, res2.info.iter)
    for variant in ('direct256', 'direct', 'w64'):
        os.environ['OSQP_HIP_BATCH_VARIANT'] = variant
        try:
            xb, yb, rec = s._solver.hip_batch_solve(q=np.stack([q, q]), l=np.stack([l, l]) if m else None, u=np.stack([u, u]) if m else None)
        except ValueError:
            continue
        direct_ran = rec[0, 7] == 0 and rec[0, 1] > 0
        if variant.startswith('direct') and not direct_ran:
            continue                                                               # this pattern fell back to PCG
        agree('batch:' + variant, int(rec[0, 0]), xb[0], rec[0, 2], int(rec[0, 1]))
        assert np.array_equal(xb[0], xb[1], equal_nan=True)
print('problems %d, checks %s, iteration-limit differences %d, mismatches %d' % (count, stats, soft, bad))
sys.exit(1 if bad else 0)
