"""Per-block mixing against the two-kernel form on random band + long-range matrices (solutions, iteration counts, which form ran).
   python tools/mix_fuzz.py [cases] [seed]"""
import os, sys, warnings, time
warnings.simplefilter("ignore")
sys.path[:0] = [".", "osqp-python_amd"]
import numpy as np, scipy.sparse as sp, osqp_amd, problems

def solve(P, q, A, l, u, f1):
    os.environ['OSQP_HIP_F1'] = str(f1)
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-7, eps_rel=1e-7, verbose=False, max_iter=40000, adaptive_rho_interval=50, check_termination=25)
    r = m.solve(); return r, m._solver.hip_stats()

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0; mixed = 0
for t in range(cases):
    n = int(rng.choice([20000, 30000, 50000, 80000])); w = int(rng.choice([20, 40, 100, 200, 300])); k = int(rng.integers(3, 8)); fr = float(rng.choice([0.0005, 0.002, 0.01, 0.02, 0.05]))
    seed = int(rng.integers(1, 10**6))
    P, q, A, l, u = problems.banded_qp(n, window=w, nnz_per_row=k, seed=seed, long_range=fr)
    if t % 3 == 2:                                         # long-range couplings in P too: operands of P entries outside the gather window
        r2 = np.random.default_rng(seed + 7); i = r2.integers(0, n, n // 50); j = r2.integers(0, n, n // 50); v = r2.uniform(-0.01, 0.01, n // 50)
        E = sp.coo_matrix((np.concatenate([v, v]), (np.concatenate([i, j]), np.concatenate([j, i]))), shape=(n, n)).tocsc()
        P = (P + E + sp.diags(np.asarray(abs(E).sum(axis=1)).ravel())).tocsc(); P.sort_indices()
    r1, s1 = solve(P, q, A, l, u, 1); r0, s0 = solve(P, q, A, l, u, 2)
    dx = np.abs(r1.x - r0.x).max() / (1 + np.abs(r0.x).max()); dy = np.abs(r1.y - r0.y).max() / (1 + np.abs(r0.y).max())
    ok = r1.info.status_val == r0.info.status_val == 1 and dx < 2e-5 and dy < 5e-5
    mixed += s1['f1_far_columns'] > 0
    bad += not ok
    print('%s n=%d window=%d nnz/row=%d long_range=%.4f seed=%d P-long-range=%d: form %d D %d far %d | iters %d / %d | |dx| %.1e |dy| %.1e | %.1f / %.1f ms'
          % ('ok ' if ok else 'BAD', n, w, k, fr, seed, t % 3 == 2, s1['pcg_fused'], s1['f1_replicas'], s1['f1_far_columns'], r1.info.iter, r0.info.iter, dx, dy, s1['gpu_solve_ms'], s0['gpu_solve_ms']), flush=True)
print('%d cases, %d ran the mixing form, %d BAD' % (cases, mixed, bad))
