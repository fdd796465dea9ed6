#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/  (run ONLY in the build container).

For every hot-path test problem of the reference's own suite (src/osqp/tests/*_test.py,
SURVEY.md §4) this script
  1. rebuilds the seeded inputs exactly as the reference fixture does (file:line cited per case),
  2. attaches the C-core known answer shipped with the reference (src/osqp/tests/solutions/<name>.npz:
     x_val, y_val, obj -- data files the reference's tests hold),
  3. runs the importable pure-python reference solver  /root/reference/src/osqppurepy  on the same
     inputs and records x, y, obj, iter, status and the per-iteration (pri_res, dua_res) trace.
Everything is stored as small .npz files (inputs + expected outputs only; no reference source).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fixtures.py
"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sparse

REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, 'src'))
import osqppurepy  # noqa: E402  (the reference's pure-python solver)

OUT = os.path.dirname(os.path.abspath(__file__))
SOL = os.path.join(REF, 'src', 'osqp', 'tests', 'solutions')

# purepy status numbering (0.6, _osqp.py:14-22) -> v1.0.0 enum order (bindings.cpp.in:349-361)
PUREPY2V1 = {1: 1, 2: 2, -3: 3, 3: 4, -4: 5, 4: 6, -2: 7, -7: 9, -10: 11}


def full_sym(P):
    P = sparse.csc_matrix(P)
    if sparse.tril(P, -1).nnz == 0:
        P = sparse.triu(P, 1).T + P
    return sparse.csc_matrix(P)


def run_purepy(P, q, A, l, u, stg, ops=(), trace_cap=400):
    """ops: list of ('update', dict) / ('warm_start', dict) / ('update_settings', dict) / ('solve',) applied in order;
    returns the result of the LAST solve."""
    m = osqppurepy.OSQP()
    Pf = full_sym(P)
    m.setup(P=Pf, q=np.asarray(q, float), A=sparse.csc_matrix(A), l=np.asarray(l, float), u=np.asarray(u, float),
            verbose=False, **stg)
    work = m._model.work
    trace = {'pri': [], 'dua': []}
    orig = m._model.update_info

    def hooked(it, polish):
        orig(it, polish)
        if polish == 0 and len(trace['pri']) < trace_cap:
            trace['pri'].append(work.info.pri_res); trace['dua'].append(work.info.dua_res)

    m._model.update_info = hooked
    res = None
    for op in ops or (('solve',),):
        if op[0] == 'solve':
            trace['pri'].clear(); trace['dua'].clear()
            res = m.solve()
        elif op[0] == 'update':
            kw = dict(op[1])
            if 'P' in kw:
                kw['P'] = full_sym(kw['P'])
            m.update(**kw)
        elif op[0] == 'warm_start':
            m.warm_start(**op[1])
        elif op[0] == 'update_settings':
            m.update_settings(**op[1])
    out = {
        'ref_iter': res.info.iter, 'ref_status': PUREPY2V1[res.info.status_val],
        'ref_obj': res.info.obj_val, 'ref_pri_res': res.info.pri_res, 'ref_dua_res': res.info.dua_res,
        'ref_trace_pri': np.array(trace['pri']), 'ref_trace_dua': np.array(trace['dua']),
        'ref_rho_updates': res.info.rho_updates,
    }
    if res.x is not None and res.x.dtype != object:
        out['ref_x'] = res.x; out['ref_y'] = res.y
    return out, m


def save(name, P, q, A, l, u, stg, extra=None, golden=None):
    P = sparse.triu(sparse.csc_matrix(P), format='csc'); P.sort_indices()
    A = sparse.csc_matrix(A); A.sort_indices()
    d = dict(n=P.shape[0], m=A.shape[0], P_data=P.data.astype(float), P_indices=P.indices.astype(np.int32),
             P_indptr=P.indptr.astype(np.int32), q=np.asarray(q, float), A_data=A.data.astype(float),
             A_indices=A.indices.astype(np.int32), A_indptr=A.indptr.astype(np.int32),
             l=np.asarray(l, float), u=np.asarray(u, float), settings=json.dumps(stg))
    if golden:
        g = np.load(os.path.join(SOL, golden + '.npz'))
        for k in g.files:
            d['gold_' + k] = g[k]
    if extra:
        d.update(extra)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
    print('%-34s n=%-5d m=%-5d %s' % (name, d['n'], d['m'], {k: (v if np.isscalar(v) else '...') for k, v in d.items()
                                                          if k in ('ref_iter', 'ref_status')}))


# purepy-equivalent settings for a reference test's opts dict
def pp(opts, **over):
    s = dict(rho=0.1, sigma=1e-6, alpha=1.6, scaling=10, max_iter=4000, eps_abs=1e-3, eps_rel=1e-3,
             eps_prim_inf=1e-4, eps_dual_inf=1e-4, adaptive_rho=True, adaptive_rho_interval=50,
             adaptive_rho_tolerance=5, check_termination=1, warm_start=True, polish=False, scaled_termination=False)
    ren = {'polishing': 'polish', 'warm_starting': 'warm_start'}
    for k, v in opts.items():
        if k in ('verbose', 'solver_type', 'polish_refine_iter'):
            continue
        k = ren.get(k, k)
        if k == 'scaling' and v is True:
            v = 10
        s[k] = v
    s.update(over)
    return s


def main():
    # ---------------- basic_test.py:12-35 ----------------
    P = sparse.diags([11.0, 0.0], format='csc'); q = np.array([3.0, 4.0])
    A = sparse.csc_matrix([[-1, 0], [0, -1], [-1, -3], [2, 5], [3, 4]], dtype=float)
    u = np.array([0.0, 0.0, -15, 100, 80]); l = -1e06 * np.ones(len(u))
    opts = dict(eps_abs=1e-9, eps_rel=1e-9, max_iter=2500, rho=0.1, adaptive_rho=False, polishing=False,
                check_termination=1, warm_starting=True)
    s = pp(opts)
    r, _ = run_purepy(P, q, A, l, u, s); save('basic_QP', P, q, A, l, u, s, r, 'test_basic_QP')
    cases = {  # basic_test.py:50-99
        'basic_update_q': dict(q=np.array([10.0, 20.0])),
        'basic_update_l': dict(l=-50.0 * np.ones(5)),
        'basic_update_u': dict(u=1000.0 * np.ones(5)),
        'basic_update_bounds': dict(l=-100.0 * np.ones(5), u=1000.0 * np.ones(5)),
    }
    for name, upd in cases.items():
        r, _ = run_purepy(P, q, A, l, u, s, ops=[('update', upd), ('solve',)])
        ex = dict(r); ex.update({'upd_' + k: v for k, v in upd.items()})
        save(name, P, q, A, l, u, s, ex, 'test_' + name[len('basic_'):])

    # ---------------- update_matrices_test.py:11-42 ----------------
    np.random.seed(1)
    n, m, p = 5, 8, 0.7
    Pt = sparse.random(n, n, density=p); Pt_new = Pt.copy(); Pt_new.data += 0.1 * np.random.randn(Pt.nnz)
    P = (Pt.T.dot(Pt) + sparse.eye(n)).tocsc(); P_new = (Pt_new.T.dot(Pt_new) + sparse.eye(n)).tocsc()
    P_triu = sparse.triu(P, format='csc'); P_triu_new = sparse.triu(P_new, format='csc')
    q = np.random.randn(n); A = sparse.random(m, n, density=p, format='csc'); A_new = A.copy()
    A_new.data += np.random.randn(A_new.nnz); l = np.zeros(m); u = 30 + np.random.randn(m)
    s = pp(dict(eps_abs=1e-8, eps_rel=1e-8))
    r, _ = run_purepy(P, q, A, l, u, s); save('matrices_solve', P, q, A, l, u, s, r, 'test_solve')
    P_triu.sort_indices(); P_triu_new.sort_indices(); A_new.sort_indices()
    assert (P_triu.indices == P_triu_new.indices).all() and (A.indices == A_new.indices).all()
    for name, (dp, da) in {'matrices_update_P': (1, 0), 'matrices_update_A': (0, 1), 'matrices_update_P_A': (1, 1)}.items():
        upd = {}
        if dp:
            upd['P'] = P_new
        if da:
            upd['A'] = A_new
        r, _ = run_purepy(P, q, A, l, u, s, ops=[('update', upd), ('solve',)])
        ex = dict(r)
        if dp:
            ex['upd_Px'] = P_triu_new.data
        if da:
            ex['upd_Ax'] = A_new.data
        gold = {'matrices_update_P': 'test_update_P', 'matrices_update_A': 'test_update_A',
                'matrices_update_P_A': 'test_update_P_A_allind'}[name]
        save(name, P, q, A, l, u, s, ex, gold)

    # ---------------- feasibility_test.py:11-43 ----------------
    np.random.seed(4)
    n = m = 30
    P = sparse.csc_matrix((n, n)); q = np.zeros(n); A = sparse.random(m, n, density=1.0, format='csc')
    u = np.random.rand(m); l = u
    s = pp(dict(eps_abs=1e-6, eps_rel=1e-6, scaling=True, alpha=1.6, max_iter=5000, polishing=False, warm_starting=True))
    r, _ = run_purepy(P, q, A, l, u, s); save('feasibility', P, q, A, l, u, s, r, 'test_feasibility_problem')

    # ---------------- unconstrained_test.py:10-34 (purepy crashes for m=0: SURVEY App. D) ----------------
    np.random.seed(4)
    n = 30
    P = (sparse.diags(np.random.rand(n)) + 0.2 * sparse.eye(n)).tocsc(); q = np.random.randn(n)
    A = sparse.csc_matrix((0, n)); l = np.array([]); u = np.array([])
    s = pp(dict(eps_abs=1e-8, eps_rel=1e-8, polishing=False))
    save('unconstrained', P, q, A, l, u, s, None, 'test_unconstrained_problem')

    # ---------------- warm_start_test.py:25-41 ----------------
    np.random.seed(2)
    n, m = 100, 200
    A = sparse.random(m, n, density=0.9, format='csc'); l = -np.random.rand(m) * 2.0; u = np.random.rand(m) * 2.0
    P = sparse.random(n, n, density=0.9); P = sparse.triu(P.dot(P.T), format='csc'); q = np.random.randn(n)
    s = pp(dict(adaptive_rho=False, eps_abs=1e-8, eps_rel=1e-8, polishing=False, check_termination=1))
    r, _ = run_purepy(P, q, A, l, u, s, trace_cap=3000); save('warm_start', P, q, A, l, u, s, r)

    # ---------------- primal_infeasibility_test.py:25-58 ----------------
    np.random.seed(4)
    n, m = 50, 500
    Pt = np.random.rand(n, n); P = sparse.triu(Pt.T.dot(Pt), format='csc'); q = np.random.rand(n)
    A = sparse.random(m, n).tolil(); u = 3 + np.random.randn(m); l = -3 + np.random.randn(m)
    A[int(n / 2), :] = A[int(n / 2) + 1, :]
    l[int(n / 2)] = u[int(n / 2) + 1] + 10 * np.random.rand(); u[int(n / 2)] = l[int(n / 2)] + 0.5
    A = A.tocsc()
    s = pp(dict(eps_abs=1e-5, eps_rel=1e-5, eps_dual_inf=1e-20, max_iter=2500, polishing=False))
    r, _ = run_purepy(P, q, A, l, u, s); save('primal_infeasible', P, q, A, l, u, s, r, 'test_primal_infeasibility')

    # ---------------- primal_infeasibility_test.py:61-77 / dual_infeasibility_test.py:75-97 ----------------
    P = sparse.csc_matrix((2, 2)); q = np.array([-1.0, -1.0])
    A = sparse.csc_matrix([[1.0, -1.0], [-1.0, 1.0], [1.0, 0.0], [0.0, 1.0]]); l = np.array([1.0, 1.0, 0.0, 0.0]); u = np.inf * np.ones(4)
    r, _ = run_purepy(P, q, A, l, u, s); save('primal_dual_infeasible', P, q, A, l, u, s, r)

    # ---------------- dual_infeasibility_test.py:31-72 ----------------
    s = pp(dict(eps_abs=1e-5, eps_rel=1e-5, eps_prim_inf=1e-15, eps_dual_inf=1e-6, scaling=3, max_iter=2500,
                polishing=False, check_termination=1))
    P = sparse.csc_matrix((2, 2)); q = np.array([2.0, -1.0]); A = sparse.eye(2, format='csc')
    l = np.array([0.0, 0.0]); u = np.array([np.inf, np.inf])
    r, _ = run_purepy(P, q, A, l, u, s)
    save('dual_infeasible_lp', P, q, A, l, u, s, r, 'test_dual_infeasibility')
    P = sparse.diags([4.0, 0.0], format='csc'); q = np.array([0.0, 2.0]); A = sparse.csc_matrix([[1.0, 1.0], [-1.0, 1.0]])
    l = np.array([-np.inf, -np.inf]); u = np.array([2.0, 3.0])
    r, _ = run_purepy(P, q, A, l, u, s)
    save('dual_infeasible_qp', P, q, A, l, u, s, r, 'test_dual_infeasibility')

    # ---------------- non_convex_test.py:11-19 ----------------
    P = sparse.triu([[2.0, 5.0], [5.0, 1.0]], format='csc'); q = np.array([3.0, 4.0])
    A = sparse.csc_matrix([[-1.0, 0.0], [0.0, -1.0], [-1.0, 3.0], [2.0, 5.0], [3.0, 4]]); u = np.array([0.0, 0.0, -15, 100, 80])
    l = -np.inf * np.ones(5)
    save('non_convex', P, q, A, l, u, pp(dict()), None)

    # ---------------- polishing_test.py:79-99 (polish is a "next" row; the ADMM solution is still pinned) -----------
    np.random.seed(6)
    n, m = 30, 50
    Pt = sparse.random(n, n); P = Pt.T @ Pt; q = np.random.randn(n); A = sparse.csc_matrix(np.random.randn(m, n))
    l = -3 + np.random.randn(m); u = 3 + np.random.randn(m)
    s = pp(dict(eps_abs=1e-6, eps_rel=1e-6, scaling=True, rho=0.1, alpha=1.6, max_iter=2500))
    r, _ = run_purepy(P, q, A, l, u, s); save('polish_random_admm', P, q, A, l, u, s, r, 'test_polish_random')

    # ---------------- BASELINE.json configs[0]: random sparse QP n=50 m=100 density 0.15 -------
    rng = np.random.default_rng(0)
    n, m = 50, 100
    M = sparse.random(n, n, density=0.15, random_state=rng, data_rvs=rng.standard_normal)
    P = (M @ M.T + 1e-2 * sparse.eye(n)).tocsc(); q = rng.standard_normal(n)
    A = sparse.random(m, n, density=0.15, random_state=rng, data_rvs=rng.standard_normal, format='csc')
    l = -rng.random(m); u = rng.random(m)
    s = pp(dict(eps_abs=1e-6, eps_rel=1e-6))
    r, _ = run_purepy(P, q, A, l, u, s); save('config1_random_qp', P, q, A, l, u, s, r)

    # ---------------- polishing_test.py:32-99 with polish ON in the python reference -------------------------------
    popts = dict(eps_abs=1e-3, eps_rel=1e-3, scaling=True, rho=0.1, alpha=1.6, max_iter=2500, polishing=True, polish_refine_iter=4)
    def run_pol(P, q, A, l, u, name, gold):
        s = pp(popts); s['polish'] = True
        m = osqppurepy.OSQP()
        m.setup(P=full_sym(P), q=np.asarray(q, float), A=sparse.csc_matrix(A), l=np.asarray(l, float), u=np.asarray(u, float), verbose=False, polish_refine_iter=4, **s)
        res = m.solve()
        ex = dict(ref_x=res.x, ref_y=res.y, ref_obj=res.info.obj_val, ref_iter=res.info.iter, ref_status=PUREPY2V1[res.info.status_val],
                  ref_status_polish=res.info.status_polish, ref_pri_res=res.info.pri_res, ref_dua_res=res.info.dua_res)
        save(name, P, q, A, l, u, s, ex, gold)
    P = sparse.diags([11.0, 0.0], format='csc'); q = np.array([3.0, 4.0])                                     # :32-49
    A = sparse.csc_matrix([[-1, 0], [0, -1], [-1, -3], [2, 5], [3, 4]], dtype=float); u = np.array([0.0, 0, -15, 100, 80]); l = -1e05 * np.ones(5)
    run_pol(P, q, A, l, u, 'polish_simple', 'test_polish_simple')
    np.random.seed(6)                                                                                          # :79-99
    n, m = 30, 50
    Pt = sparse.random(n, n); P = Pt.T @ Pt; q = np.random.randn(n); A = sparse.csc_matrix(np.random.randn(m, n))
    l = -3 + np.random.randn(m); u = 3 + np.random.randn(m)
    run_pol(P, q, A, l, u, 'polish_random', 'test_polish_random')
    np.random.seed(4)                                                                                          # :52-76 (m = 0: purepy cannot run it)
    n = 30
    P = (sparse.diags(np.random.rand(n)) + 0.2 * sparse.eye(n)).tocsc(); q = np.random.randn(n)
    s = pp(popts); s['polish'] = True
    save('polish_unconstrained', P, q, sparse.csc_matrix((0, n)), np.array([]), np.array([]), s, None, 'test_polish_unconstrained')


if __name__ == '__main__':
    main()
