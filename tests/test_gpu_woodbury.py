"""GPU tier: the Woodbury-corrected preconditioner for a few dense rows (portfolio-like QPs) -- plain Jacobi, the
corrected preconditioner inside the PCG, and the direct mode (the rest of K diagonal: M^-1 is K^-1) reach the same solution as the
oracle's direct solve; the corrected forms need ~1 PCG iteration per ADMM iteration where Jacobi needs many."""
import contextlib
import os
import warnings

import numpy as np
import pytest

import osqp_amd
import problems
from oracle import Oracle, SOLVED

pytestmark = pytest.mark.gpu
warnings.simplefilter('ignore')


def _solve(P, q, A, l, u, woodbury, direct, **kw):
    old = {k: os.environ.get(k) for k in ('OSQP_HIP_WOODBURY', 'OSQP_HIP_WOODBURY_DIRECT')}
    os.environ['OSQP_HIP_WOODBURY'] = str(int(woodbury)); os.environ['OSQP_HIP_WOODBURY_DIRECT'] = str(int(direct))
    try:
        st = dict(eps_abs=1e-7, eps_rel=1e-7, max_iter=50000, adaptive_rho_interval=50, check_termination=25, verbose=False)
        st.update(kw)
        m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, **st)
        r = m.solve()
        return m, r, m._solver.hip_stats()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@contextlib.contextmanager
def _env(**kv):
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _rel(a, b):
    return np.abs(a - b).max() / (1 + np.abs(b).max())


def test_portfolio_three_preconditioner_modes_agree_with_the_oracle():
    P, q, A, l, u = problems.portfolio_qp(2000, 20)
    xo, yo, io = Oracle().setup(P, q, A, l, u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000, adaptive_rho_interval=50).solve()
    assert io.status_val == SOLVED
    res = {}
    for name, wb, direct in (('jacobi', 0, 0), ('woodbury-pcg', 1, 0), ('woodbury-direct', 1, 1)):
        m, r, s = _solve(P, q, A, l, u, wb, direct)
        assert r.info.status_val == 1, name
        res[name] = (r, s)
        print('%-16s %d iterations, %.2f PCG iterations each, %d launches; |dx| %.2e |dy| %.2e' %
              (name, r.info.iter, s['pcg_iters_total'] / r.info.iter, s['kernel_launches'], _rel(r.x, xo), _rel(r.y, yo)))
        assert _rel(r.x, xo) < 5e-5 and _rel(r.y, yo) < 2e-4            # (an eps = 1e-7 iterate against the 1e-9 solution)
        assert abs(r.info.obj_val - io.obj_val) <= 2e-5 * (1 + abs(io.obj_val))
    pj = res['jacobi'][1]['pcg_iters_total'] / res['jacobi'][0].info.iter
    pw = res['woodbury-pcg'][1]['pcg_iters_total'] / res['woodbury-pcg'][0].info.iter
    assert pw <= 1.5 and pj >= 3 * pw                                     # the correction is exact here: one iteration confirms it
    assert res['woodbury-direct'][1]['kernel_launches'] < 0.7 * res['woodbury-pcg'][1]['kernel_launches']
    assert abs(res['woodbury-direct'][0].info.iter - res['woodbury-pcg'][0].info.iter) <= 50


def test_dense_rows_next_to_a_banded_block_use_the_corrected_preconditioner_inside_the_pcg():
    """K0 not diagonal (banded rows with five entries): no direct mode, the correction works as a preconditioner."""
    import scipy.sparse as sp
    P, q, A, l, u = problems.banded_qp(3000, window=30)
    rng = np.random.default_rng(1)
    dense = sp.random(6, 3000, density=0.3, random_state=rng, data_rvs=rng.standard_normal, format='csc')
    A2 = sp.vstack([A, dense], format='csc')
    x0 = 0.1 * rng.standard_normal(3000)
    l2 = np.concatenate([l, dense @ x0 - 1.0]); u2 = np.concatenate([u, dense @ x0 + 1.0])
    xo, yo, io = Oracle().setup(P, q, A2, l2, u2, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000, adaptive_rho_interval=50).solve()
    assert io.status_val == SOLVED
    mj, rj, sj = _solve(P, q, A2, l2, u2, 0, 0)
    mw, rw, sw = _solve(P, q, A2, l2, u2, 1, 1)
    print('jacobi %d it %.2f pcg; woodbury %d it %.2f pcg' % (rj.info.iter, sj['pcg_iters_total'] / rj.info.iter, rw.info.iter, sw['pcg_iters_total'] / rw.info.iter))
    for r in (rj, rw):
        assert r.info.status_val == 1 and _rel(r.x, xo) < 5e-5 and _rel(r.y, yo) < 2e-4
    assert sw['pcg_iters_total'] / rw.info.iter < sj['pcg_iters_total'] / rj.info.iter


def test_lasso_many_dense_rows_large_rank_correction():
    """600 sample rows of 301 entries next to the two-entry rows  -t <= x <= t : more long rows than the host-factorised form takes, so the
    r x r system is formed, factorised and inverted on the device (backend.h kWbLargeMax).  The correction is exact here (the two-entry
    rows' contributions to the off-diagonal of K0 cancel) and the probe after each factorisation finds that out: direct mode."""
    P, q, A, l, u = problems.lasso_qp(300, 600)
    xo, yo, io = Oracle().setup(P, q, A, l, u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000).solve()
    assert io.status_val == SOLVED
    res = {}
    for name, env in (('jacobi', {'OSQP_HIP_WOODBURY_LARGE': '0'}), ('large-pcg', {'OSQP_HIP_WOODBURY_DIRECT': '0'}), ('large-direct', {})):
        with _env(**env):
            m = osqp_amd.OSQP()
            m.setup(P, q, A, l, u, eps_abs=1e-7, eps_rel=1e-7, verbose=False, max_iter=50000)
            r = m.solve(raise_error=True)
            s = m._solver.hip_stats()
        res[name] = (r, s)
        print('%-14s %d iterations, %.2f PCG iterations each, %d launches, rows %d direct %d; |dx| %.2e |dy| %.2e' %
              (name, r.info.iter, s['pcg_iters_total'] / r.info.iter, s['kernel_launches'], s['woodbury_rows'], s['woodbury_direct'], _rel(r.x, xo), _rel(r.y, yo)))
        assert _rel(r.x, xo) < 5e-5 and _rel(r.y, yo) < 2e-4
        assert abs(r.info.obj_val - io.obj_val) <= 2e-5 * (1 + abs(io.obj_val))
    assert res['jacobi'][1]['woodbury_rows'] == 0 and res['large-pcg'][1]['woodbury_rows'] == 600
    assert res['large-pcg'][1]['woodbury_direct'] == 0 and res['large-direct'][1]['woodbury_direct'] == 1
    pj = res['jacobi'][1]['pcg_iters_total'] / res['jacobi'][0].info.iter
    pw = res['large-pcg'][1]['pcg_iters_total'] / res['large-pcg'][0].info.iter
    assert pw <= 1.5 and pj >= 3 * pw
    assert res['large-direct'][1]['kernel_launches'] < 0.7 * res['large-pcg'][1]['kernel_launches']


def test_large_rank_correction_follows_matrix_updates_and_polish():
    import scipy.sparse as sp
    P, q, A, l, u = problems.lasso_qp(200, 400, seed=3)
    m = osqp_amd.OSQP()
    m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=50000, polishing=True)
    assert m._solver.hip_stats()['woodbury_rows'] == 400
    r0 = m.solve(raise_error=True)
    rng = np.random.default_rng(0)
    Ax = A.data * (1 + 0.05 * rng.standard_normal(A.nnz))
    m.update(Ax=Ax)
    r1 = m.solve(raise_error=True)
    A2 = sp.csc_matrix((Ax, A.indices, A.indptr), shape=A.shape)
    xo, yo, io = Oracle().setup(P, q, A2, l, u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000).solve()
    assert io.status_val == SOLVED
    assert abs(r1.info.obj_val - io.obj_val) <= 2e-5 * (1 + abs(io.obj_val)) and _rel(r1.x, xo) < 1e-4
    assert abs(r0.info.obj_val - r1.info.obj_val) > 1e-6 * (1 + abs(io.obj_val))          # (the update did change the problem)


def test_large_rank_correction_with_dense_P_is_only_a_preconditioner():
    """All 300 rows of A dense (200 entries) and P dense: K0 = P + sigma I is far from diagonal, the probe must refuse the direct mode and the
    PCG must still reach the oracle's solution with the corrected preconditioner."""
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    n, m = 200, 300
    M = rng.standard_normal((n, n)); P = sp.csc_matrix(M @ M.T / n + 0.1 * np.eye(n))
    A = sp.csc_matrix(rng.standard_normal((m, n)))
    q = rng.standard_normal(n); l = -rng.random(m) - 0.1; u = rng.random(m) + 0.1
    l[:20] = u[:20] = 0.05 * rng.standard_normal(20)
    xo, yo, io = Oracle().setup(P, q, A, l, u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000).solve()
    assert io.status_val == SOLVED
    mdl = osqp_amd.OSQP()
    mdl.setup(P, q, A, l, u, eps_abs=1e-7, eps_rel=1e-7, verbose=False, max_iter=50000)
    r = mdl.solve(raise_error=True)
    s = mdl._solver.hip_stats()
    print('dense P: %d iterations, %.2f PCG each, rows %d direct %d, |dx| %.2e |dy| %.2e' % (r.info.iter, s['pcg_iters_total'] / r.info.iter, s['woodbury_rows'], s['woodbury_direct'], _rel(r.x, xo), _rel(r.y, yo)))
    assert s['woodbury_rows'] == 300 and s['woodbury_direct'] == 0
    assert _rel(r.x, xo) < 5e-5 and _rel(r.y, yo) < 2e-4


@pytest.mark.parametrize('size', [(400, 20), (2000, 50), (10000, 100)])
def test_direct_mode_in_two_launches_equals_the_five_launch_form(size):
    """wbdirect_hip.hip (X / Y: two launches per ADMM iteration, the dense tile of the long rows in LDS, partial reductions folded in index
    order) against the r03 form of the same direct mode (KB, three kernels of M^-1, KA): same algorithm, other summation order -- equal
    ADMM iteration counts, x / y to 1e-9 of the solution's scale, the oracle's solution at the usual tolerance; updates of q / bounds, a rho
    update and a warm start go through both."""
    P, q, A, l, u = problems.portfolio_qp(*size)
    rng = np.random.default_rng(5)
    out = {}
    for fused in (0, 1):
        with _env(OSQP_HIP_WOODBURY_FUSED=str(fused)):
            m, r, s = _solve(P, q, A, l, u, 1, 1)
            assert r.info.status_val == 1 and s['woodbury_direct'] == (2 if fused else 1), (fused, s['woodbury_direct'])
            q2 = q * (1.0 + 0.05 * np.sin(np.arange(len(q))))
            m.update(q=q2); r2 = m.solve()
            m.update_settings(rho=0.37); m.warm_start(x=0.5 * r.x, y=0.5 * r.y); r3 = m.solve()
            out[fused] = (r, r2, r3, s, m._solver.hip_stats())
    for k in range(3):
        a, b = out[0][k], out[1][k]
        assert a.info.status_val == b.info.status_val == 1, k
        assert abs(a.info.iter - b.info.iter) <= 25, (k, a.info.iter, b.info.iter)
        assert _rel(b.x, a.x) < 1e-7 and _rel(b.y, a.y) < 1e-6, (k, _rel(b.x, a.x), _rel(b.y, a.y))
    assert out[1][3]['kernel_launches'] < 0.55 * out[0][3]['kernel_launches'], (out[1][3]['kernel_launches'], out[0][3]['kernel_launches'])
    if size[0] <= 2000:
        xo, yo, io = Oracle().setup(P, q, A, l, u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000, adaptive_rho_interval=50).solve()
        assert io.status_val == SOLVED and _rel(out[1][0].x, xo) < 5e-5 and _rel(out[1][0].y, yo) < 2e-4


def test_direct_mode_runs_device_driven_and_hands_over_when_a_refactorisation_fails():
    """Round 4: the two-launch direct mode is a slot form (k_wbx_slot_x / _y), the r x r system is inverted and checked on the device
    (k_wb_invert), so its chunk boundaries CAN be decided on the device like the banded QPs' (OSQPHipPolicy::device_driven = 2; the default
    stays host-synchronous for this form, which is faster): same solution as the host-synchronous loop.  Test hook OSQPHipPolicy::debug_fail_refactor: the next device-side inversion reports 'inaccurate' -- the solve
    must be handed to the host (PCG with the corrected preconditioner) and still arrive at the same solution."""
    P, q, A, l, u = problems.portfolio_qp(2000, 50)
    xo, yo, io = Oracle().setup(P, q, A, l, u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000, adaptive_rho_interval=50).solve()
    assert io.status_val == SOLVED
    res = {}
    for dd in (0, 2):
        with _env(OSQP_HIP_DEVICE_DRIVEN=str(dd)):
            m, r, s = _solve(P, q, A, l, u, 1, 1)
            assert r.info.status_val == 1 and s['woodbury_direct'] == 2 and r.info.rho_updates >= 1, (dd, s['woodbury_direct'], r.info.rho_updates)
            assert _rel(r.x, xo) < 5e-5 and _rel(r.y, yo) < 2e-4
            res[dd] = (r, s)
    assert res[0][0].info.iter == res[2][0].info.iter and res[0][0].info.rho_updates == res[2][0].info.rho_updates
    assert _rel(res[2][0].x, res[0][0].x) < 1e-9 and _rel(res[2][0].y, res[0][0].y) < 1e-8
    assert res[2][1]['graph_launches'] > 0
    # the hand-over
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-7, eps_rel=1e-7, max_iter=50000, adaptive_rho_interval=50, check_termination=25, verbose=False)
    m._solver.set_policy(debug_fail_refactor=1, device_driven=2)
    r = m.solve(); s = m._solver.hip_stats()
    assert r.info.status_val == 1 and r.info.rho_updates >= 1
    assert s['woodbury_direct'] == 0 and s['woodbury_rows'] > 0 and s['pcg_iters_total'] > 0      # finished by the PCG with the corrected preconditioner
    assert _rel(r.x, xo) < 5e-5 and _rel(r.y, yo) < 2e-4


def test_two_launch_direct_mode_is_deterministic():
    """every sum of wbdirect_hip.hip has a fixed order (partials folded in index order, no atomics): repeated cold solves on one handle and a
    second handle give bit-identical x, y and iteration counts"""
    P, q, A, l, u = problems.portfolio_qp(2000, 50)
    got = []
    for h in range(2):
        m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-7, eps_rel=1e-7, max_iter=50000, adaptive_rho_interval=50, check_termination=25, verbose=False, warm_starting=False)
        for rep in range(3):
            m.update_settings(rho=0.1)
            r = m.solve()
            assert r.info.status_val == 1 and m._solver.hip_stats()['woodbury_direct'] == 2
            got.append((r.info.iter, r.info.rho_updates, r.x.copy(), r.y.copy()))
    for g in got[1:]:
        assert g[:2] == got[0][:2] and np.array_equal(g[2], got[0][2]) and np.array_equal(g[3], got[0][3])


def test_duplicate_entry_in_a_dense_row_keeps_the_direct_mode_off():
    """a valid CSC that stores one (row, column) of a DENSE row twice (the engine keeps the two as separate CSR entries; every SpMV
    sums them) has one cell in the dense tiles of the Woodbury forms, which the fill kernels assign.  The direct mode (M taken for K) must then
    stay off -- the solve goes through the PCG with the (slightly different) corrected preconditioner and still matches the oracle on the
    matrix with the duplicates summed."""
    import scipy.sparse as sp
    P, q, A, l, u = problems.portfolio_qp(1500, 12)
    A = A.tocsc(); A.sort_indices()
    # duplicate one entry of the first dense row (the budget / factor rows hold hundreds of entries): (i, j) stored as 0.6 v and 0.4 v
    counts = np.diff(A.tocsr().indptr); i = int(np.argmax(counts)); assert counts[i] > 128
    j = int(A.tocsr().indices[A.tocsr().indptr[i] + 3])
    indptr, indices, data = [0], [], []
    for c in range(A.shape[1]):
        for k in range(A.indptr[c], A.indptr[c + 1]):
            if c == j and A.indices[k] == i:
                indices += [i, i]; data += [0.6 * A.data[k], 0.4 * A.data[k]]
            else:
                indices.append(A.indices[k]); data.append(A.data[k])
        indptr.append(len(indices))
    Adup = sp.csc_matrix((np.array(data), np.array(indices), np.array(indptr)), shape=A.shape)
    assert Adup.nnz == A.nnz + 1
    ext = osqp_amd.interface._backend('hip')
    st = ext.OSQPSettings(); ext.osqp_set_default_settings(st)
    st.verbose = 0; st.eps_abs = st.eps_rel = 1e-7; st.max_iter = 50000; st.adaptive_rho_interval = 50
    Pt = sp.triu(P).tocsc()
    solver = ext.OSQPSolver(ext.CSC(Pt), q, ext.CSC(Adup), l, u, A.shape[0], A.shape[1], st)      # straight through the C ABI: no duplicate summing on the way
    solver.solve()
    s = solver.hip_stats()
    assert solver.info.status_val == 1
    assert s['woodbury_rows'] > 0 and s['woodbury_direct'] == 0          # preconditioner yes, direct mode no
    xo, yo, io = Oracle().setup(P, q, A, l, u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000, adaptive_rho_interval=50).solve()
    assert io.status_val == SOLVED
    assert _rel(np.array(solver.solution.x), xo) < 5e-5
    # the same data without the duplicate takes the direct mode (the check above is not vacuous)
    solver2 = ext.OSQPSolver(ext.CSC(Pt), q, ext.CSC(A), l, u, A.shape[0], A.shape[1], st)
    solver2.solve()
    assert solver2.hip_stats()['woodbury_direct'] >= 1


def test_cached_inverses_serve_repeated_solves_and_die_with_the_matrices():
    """backend.h DevWb::cache_buf: a handle that restarts from the setting's rho and walks the same rho values again finds its inverses by rho_bar
    (validated by the numerical probe): the second solve factorises nothing and is bit-identical to the first.  After a matrix update the same
    rho_bar values meet OTHER matrices: the probe rejects every entry, the solve factorises again and matches the oracle on the new data."""
    import scipy.sparse as sp
    P, q, A, l, u = problems.lasso_qp(300, 600)
    st = dict(eps_abs=1e-7, eps_rel=1e-7, verbose=False, max_iter=50000, warm_starting=False)
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, **st)
    r1 = m.solve(raise_error=True); s1 = m._solver.hip_stats()
    m.update_settings(rho=0.1)
    r2 = m.solve(raise_error=True); s2 = m._solver.hip_stats()
    assert s1['woodbury_direct'] == 1 and s1['woodbury_cache_hits'] == 0
    assert s1['woodbury_factorisations'] >= 1 and s2['woodbury_factorisations'] == 0 and s2['woodbury_cache_hits'] == s1['woodbury_factorisations']      # (the rho reset itself is a hit too, outside the solve's statistics)
    assert r1.info.iter == r2.info.iter and np.array_equal(r1.x, r2.x) and np.array_equal(r1.y, r2.y)
    rng = np.random.default_rng(5)
    Ax = A.data * (1 + 0.05 * rng.standard_normal(A.nnz))
    m.update(Ax=Ax); m.update_settings(rho=0.1)
    r3 = m.solve(raise_error=True); s3 = m._solver.hip_stats()
    assert s3['woodbury_cache_hits'] == 0 and s3['woodbury_factorisations'] >= 1
    A2 = sp.csc_matrix((Ax, A.indices, A.indptr), shape=A.shape)
    xo, yo, io = Oracle().setup(P, q, A2, l, u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000).solve()
    assert io.status_val == SOLVED and _rel(r3.x, xo) < 5e-5
    with _env(OSQP_HIP_WOODBURY_CACHE='0'):                                   # the switch: every rho update factorises
        m2 = osqp_amd.OSQP(); m2.setup(P, q, A, l, u, **st)
        m2.solve(); m2.update_settings(rho=0.1); r4 = m2.solve(raise_error=True); s4 = m2._solver.hip_stats()
    assert s4['woodbury_cache_hits'] == 0 and s4['woodbury_factorisations'] >= 1 and np.array_equal(r4.x, r1.x)


def test_column_space_form_equals_the_row_space_form():
    """backend.h DevWb::dual (OSQPHipPolicy::woodbury_dual): with fewer dense columns than 3/4 of the dense rows the device-factorised correction works on
    the cd x cd system of the dense columns (lasso 300 x 600: cd = 300 against r = 600; the -y_i entries are singletons).  The same M as the row-space
    form: same direct mode, same solution as that form and as the oracle."""
    P, q, A, l, u = problems.lasso_qp(300, 600)
    xo, yo, io = Oracle().setup(P, q, A, l, u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000).solve()
    assert io.status_val == SOLVED
    out = {}
    for name, env in (('dual', {'OSQP_HIP_WOODBURY_DUAL': '1'}), ('row', {'OSQP_HIP_WOODBURY_DUAL': '0'})):
        with _env(**env):
            m = osqp_amd.OSQP()
            m.setup(P, q, A, l, u, eps_abs=1e-8, eps_rel=1e-8, verbose=False, max_iter=50000)
            r = m.solve(raise_error=True)
            out[name] = (r, m._solver.hip_stats())
    (rd, sd), (rr, sr) = out['dual'], out['row']
    assert sd['woodbury_dual_cols'] == 300 and sr['woodbury_dual_cols'] == 0 and sd['woodbury_rows'] == sr['woodbury_rows'] == 600
    assert sd['woodbury_direct'] == 1 and sr['woodbury_direct'] == 1
    print('column space: %d iterations, row space: %d; |dx| %.2e |dy| %.2e; vs oracle |dx| %.2e' % (rd.info.iter, rr.info.iter, _rel(rd.x, rr.x), _rel(rd.y, rr.y), _rel(rd.x, xo)))
    assert rd.info.iter == rr.info.iter
    assert _rel(rd.x, rr.x) < 1e-7 and _rel(rd.y, rr.y) < 1e-6
    assert _rel(rd.x, xo) < 5e-6 and _rel(rd.y, yo) < 2e-5


def test_column_space_form_with_several_singletons_per_row():
    """Rows  A_d x - y1 + 0.5 y2 = b : TWO singleton columns per dense row (the per-row Sherman-Morrison elimination in its general form, sigma_a = sum of two
    terms), P diagonal, one-entry short rows: the correction is exact (direct mode) and the solution is the oracle's."""
    import scipy.sparse as sp
    rng = np.random.default_rng(11)
    nf, ns = 150, 400
    Ad = sp.csc_matrix(rng.standard_normal((ns, nf)))
    b = rng.standard_normal(ns)
    n = nf + 2 * ns
    P = sp.diags(np.concatenate([0.01 * np.ones(nf), 2.0 * np.ones(ns), 1.0 * np.ones(ns)]), format='csc')
    q = np.concatenate([0.1 * rng.standard_normal(nf), np.zeros(2 * ns)])
    A = sp.vstack([sp.hstack([Ad, -sp.eye(ns), 0.5 * sp.eye(ns)]), sp.hstack([sp.eye(nf), sp.csc_matrix((nf, 2 * ns))])], format='csc')
    l = np.concatenate([b, -np.ones(nf)]); u = np.concatenate([b, np.ones(nf)])
    xo, yo, io = Oracle().setup(P, q, A, l, u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000).solve()
    assert io.status_val == SOLVED
    m = osqp_amd.OSQP()
    m.setup(P, q, A, l, u, eps_abs=1e-8, eps_rel=1e-8, verbose=False, max_iter=50000)
    r = m.solve(raise_error=True)
    s = m._solver.hip_stats()
    print('two singletons per row: %d iterations, %.2f PCG each, rows %d, dense columns %d, direct %d; |dx| %.2e |dy| %.2e' %
          (r.info.iter, s['pcg_iters_total'] / r.info.iter, s['woodbury_rows'], s['woodbury_dual_cols'], s['woodbury_direct'], _rel(r.x, xo), _rel(r.y, yo)))
    assert s['woodbury_rows'] == ns and s['woodbury_dual_cols'] == nf and s['woodbury_direct'] == 1
    assert _rel(r.x, xo) < 5e-6 and _rel(r.y, yo) < 2e-5


def test_fused_column_space_iteration_equals_the_unfused_one():
    """backend.h DevWb::fused: the ADMM iteration of the column-space direct mode as seven launches that stream the dense block twice (no KB / KA launch)
    against the unfused sequence KB, five M^-1 launches, KA (OSQP_HIP_WOODBURY_FUSED=0): the same iteration in another order of operations -- equal
    iteration counts, x / y to 1e-9, both equal to the oracle; warm-started re-solve and polish go through the same kernels."""
    P, q, A, l, u = problems.lasso_qp(300, 600)
    xo, yo, io = Oracle().setup(P, q, A, l, u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000).solve()
    out = {}
    for name, env in (('fused', {'OSQP_HIP_WOODBURY_FUSED': '1'}), ('unfused', {'OSQP_HIP_WOODBURY_FUSED': '0'})):
        with _env(**env):
            m = osqp_amd.OSQP()
            m.setup(P, q, A, l, u, eps_abs=1e-8, eps_rel=1e-8, verbose=False, max_iter=50000, polishing=True)
            r = m.solve(raise_error=True)
            out[name] = (m, r, m._solver.hip_stats())
    (mf, rf, sf), (mu, ru, su) = out['fused'], out['unfused']
    assert sf['woodbury_fused_iteration'] == 1 and su['woodbury_fused_iteration'] == 0 and sf['woodbury_dual_cols'] == su['woodbury_dual_cols'] == 300
    print('fused %d iterations (%d launches), unfused %d (%d); |dx| %.2e |dy| %.2e; vs oracle %.2e' % (rf.info.iter, sf['kernel_launches'], ru.info.iter, su['kernel_launches'], _rel(rf.x, ru.x), _rel(rf.y, ru.y), _rel(rf.x, xo)))
    assert rf.info.iter == ru.info.iter and rf.info.status_polish == ru.info.status_polish
    assert _rel(rf.x, ru.x) < 1e-9 and _rel(rf.y, ru.y) < 1e-8
    assert _rel(rf.x, xo) < 5e-6 and _rel(rf.y, yo) < 2e-5
    # a parametric re-solve on the fused handle: new q / bounds, warm start
    rng = np.random.default_rng(1)
    q2 = q * (1 + 0.1 * rng.standard_normal(len(q)))
    with _env(OSQP_HIP_WOODBURY_FUSED='1'):
        mf.update(q=q2); mf.update_settings(warm_starting=True)
        r2 = mf.solve(raise_error=True)
    x2, y2, i2 = Oracle().setup(P, q2, A, l, u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000).solve()
    assert _rel(r2.x, x2) < 5e-6 and abs(r2.info.obj_val - i2.obj_val) <= 1e-6 * (1 + abs(i2.obj_val))


def test_fused_iteration_with_the_dense_block_held_dense():
    """backend.h DevWb::dense: from 512 dense columns on (even count, block at least half full) the fused iteration's two passes over the dense block are
    dense kernels on a row-major copy (k_wbf_gd + k_wbf_gr, k_wbf_td) and the short rows / columns take one thread each (k_wbf_rb, k_wbf_s2;
    OSQPHipStats::woodbury_fused_iteration = 3) instead of CSR passes
    (OSQP_HIP_WOODBURY_FUSED=2 keeps those): the same iteration -- equal iteration counts, x / y to 1e-9, both equal to the oracle; the copy follows
    osqp_update_data_mat."""
    P, q, A, l, u = problems.lasso_qp(600, 1200)
    xo, yo, io = Oracle().setup(P, q, A, l, u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000).solve()
    out = {}
    for name, env in (('dense', {'OSQP_HIP_WOODBURY_FUSED': '1'}), ('csr', {'OSQP_HIP_WOODBURY_FUSED': '2'}), ('unfused', {'OSQP_HIP_WOODBURY_FUSED': '0'})):
        with _env(**env):
            m = osqp_amd.OSQP()
            m.setup(P, q, A, l, u, eps_abs=1e-8, eps_rel=1e-8, verbose=False, max_iter=50000)
            r = m.solve(raise_error=True)
            out[name] = (m, r, m._solver.hip_stats())
    (md, rd, sd), (mc, rc, sc), (mu, ru, su) = out['dense'], out['csr'], out['unfused']
    assert sd['woodbury_fused_iteration'] == 3 and sc['woodbury_fused_iteration'] == 1 and su['woodbury_fused_iteration'] == 0 and sd['woodbury_dual_cols'] == 600
    print('dense %d iterations (%d launches), csr %d (%d), unfused %d; |dx| %.2e %.2e' % (rd.info.iter, sd['kernel_launches'], rc.info.iter, sc['kernel_launches'], ru.info.iter, _rel(rd.x, rc.x), _rel(rd.x, ru.x)))
    assert rd.info.iter == rc.info.iter == ru.info.iter
    assert _rel(rd.x, rc.x) < 1e-9 and _rel(rd.y, rc.y) < 1e-8 and _rel(rd.x, ru.x) < 1e-9
    assert _rel(rd.x, xo) < 5e-6 and _rel(rd.y, yo) < 2e-5
    # new matrix values: the dense copy is refilled
    rng = np.random.default_rng(3)
    Ax = A.data * (1 + 0.02 * rng.standard_normal(A.nnz) * (np.abs(np.abs(A.data) - 1.0) > 1e-12))
    import scipy.sparse as sp
    A2 = sp.csc_matrix((Ax, A.indices, A.indptr), shape=A.shape)
    with _env(OSQP_HIP_WOODBURY_FUSED='1'):
        md.update(Ax=Ax)
        r2 = md.solve(raise_error=True)
    x2, y2, i2 = Oracle().setup(P, q, A2, l, u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000).solve()
    assert r2.info.status_val == 1 and _rel(r2.x, x2) < 5e-6 and abs(r2.info.obj_val - i2.obj_val) <= 1e-6 * (1 + abs(i2.obj_val))
