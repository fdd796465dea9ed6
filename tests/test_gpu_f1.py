"""GPU tier: the one-launch-per-PCG-iteration form (k_slot1, DESIGN.md §4.5: A-only operator apply with replica vectors for
A't, double-buffered r / s, device-scheduled KB / F_k / KA phases) against the two-kernel form of the same engine and against the
oracle's direct solve.  Both forms execute the same Chronopoulos-Gear recurrences; they differ in summation order only, so their
solutions agree far inside the termination tolerance and their iteration counts to within a termination check."""
import os
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import osqp_amd
import problems
from oracle import Oracle, SOLVED

pytestmark = pytest.mark.gpu
warnings.simplefilter('ignore')


def _solve(P, q, A, l, u, f1, graph=True, **kw):
    old = {k: os.environ.get(k) for k in ('OSQP_HIP_F1', 'OSQP_HIP_GRAPH')}
    os.environ['OSQP_HIP_F1'] = '1' if f1 else '0'
    os.environ['OSQP_HIP_GRAPH'] = '1' if graph else '0'
    try:
        st = dict(eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, adaptive_rho_interval=50, check_termination=25, verbose=False)
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


def _rel(a, b):
    return np.abs(a - b).max() / (1 + np.abs(b).max())


# (50 000: the plan needs FULL row blocks, fewer than the grid -- round 4; 1 000 000: A re-blocked to <= 1024 entries, ten blocks per workgroup -- round 4)
@pytest.mark.parametrize('n,window', [(20000, 40), (50000, 200), (100000, 200), (1000000, 200)])
def test_f1_form_matches_two_kernel_form_and_oracle(n, window):
    P, q, A, l, u = problems.banded_qp(n, window=window)
    m1, r1, s1 = _solve(P, q, A, l, u, True)
    m0, r0, s0 = _solve(P, q, A, l, u, False)
    assert int(s1['pcg_fused']) == 2 and 1 <= int(s1['f1_replicas']) <= 4, s1
    assert int(s0['pcg_fused']) == 1 and int(s0['f1_replicas']) == 0
    assert r1.info.status_val == r0.info.status_val == osqp_amd.SolverStatus.OSQP_SOLVED
    print('n=%d F1: %d iterations, %.1f PCG each, %d launches; two-kernel: %d iterations, %.1f PCG each, %d launches; |dx| %.2e |dy| %.2e'
          % (n, r1.info.iter, s1['pcg_iters_total'] / r1.info.iter, s1['kernel_launches'], r0.info.iter, s0['pcg_iters_total'] / r0.info.iter,
             s0['kernel_launches'], _rel(r1.x, r0.x), _rel(r1.y, r0.y)))
    assert _rel(r1.x, r0.x) < 1e-4 and _rel(r1.y, r0.y) < 1e-4          # (two iterates that both stopped at residuals <= 1e-6)
    assert abs(r1.info.obj_val - r0.info.obj_val) <= 1e-6 * (1 + abs(r0.info.obj_val))
    assert abs(r1.info.iter - r0.info.iter) <= 50
    assert s1['kernel_launches'] < 0.75 * s0['kernel_launches']
    if n <= 20000:                                     # the oracle's direct solve takes seconds at this size
        xo, yo, io = Oracle().setup(P, q, A, l, u, eps_abs=1e-8, eps_rel=1e-8, max_iter=50000, adaptive_rho_interval=50).solve()
        assert io.status_val == SOLVED
        assert _rel(r1.x, xo) < 1e-4 and _rel(r1.y, yo) < 1e-4


def test_f1_form_tight_tolerance_against_oracle():
    P, q, A, l, u = problems.banded_qp(20000, window=40)
    kw = dict(eps_abs=1e-8, eps_rel=1e-8, max_iter=50000)
    _, r1, s1 = _solve(P, q, A, l, u, True, **kw)
    assert int(s1['pcg_fused']) == 2
    xo, yo, io = Oracle().setup(P, q, A, l, u, adaptive_rho_interval=50, check_termination=25, **kw).solve()
    assert io.status_val == SOLVED and r1.info.status_val == osqp_amd.SolverStatus.OSQP_SOLVED
    print('tight: engine %d iterations, oracle %d; |dx| %.2e |dy| %.2e' % (r1.info.iter, io.iter, _rel(r1.x, xo), _rel(r1.y, yo)))
    assert _rel(r1.x, xo) < 2e-6 and _rel(r1.y, yo) < 2e-6


def test_f1_form_is_deterministic_and_graph_replay_equals_eager_launches():
    P, q, A, l, u = problems.banded_qp(20000, window=40)
    m, ra, _ = _solve(P, q, A, l, u, True)
    rb = m.solve()                                      # cold start again (warm_starting defaults to True in the front-end: force it)
    m2, rc, _ = _solve(P, q, A, l, u, True, graph=False)
    m3, rd, _ = _solve(P, q, A, l, u, True)
    assert ra.info.iter == rc.info.iter == rd.info.iter
    assert np.array_equal(ra.x, rc.x) and np.array_equal(ra.y, rc.y)          # graph replay vs eager launches: bit-identical
    assert np.array_equal(ra.x, rd.x) and np.array_equal(ra.y, rd.y)          # a second handle: bit-identical
    assert rb.info.status_val == osqp_amd.SolverStatus.OSQP_SOLVED


def test_f1_form_warm_start_update_and_cap():
    """Vector updates, warm starts and a binding PCG cap (cg_max_iter = 3: every solve stops at the cap, the last budgeted update
    applies no operator) through the F1 phases."""
    P, q, A, l, u = problems.banded_qp(20000, window=40)
    m, r, s = _solve(P, q, A, l, u, True, warm_starting=True)
    assert int(s['pcg_fused']) == 2
    rng = np.random.default_rng(3)
    q2 = q + 0.01 * rng.standard_normal(q.size)
    m.update(q=q2)
    r2 = m.solve()
    m0, _, _ = _solve(P, q2, A, l, u, False)
    r0 = m0.solve()
    assert r2.info.status_val == r0.info.status_val == osqp_amd.SolverStatus.OSQP_SOLVED
    assert _rel(r2.x, r0.x) < 1e-4 and _rel(r2.y, r0.y) < 1e-4
    mc, rc, sc = _solve(P, q, A, l, u, True, cg_max_iter=3)
    assert rc.info.status_val == osqp_amd.SolverStatus.OSQP_SOLVED
    assert sc['pcg_unconverged'] > 0 or sc['cg_cap_escalations'] > 0          # the cap did bind (and may have been escalated)
    assert _rel(rc.x, r.x) < 1e-4 and _rel(rc.y, r.y) < 1e-4


def test_f1_form_is_not_taken_where_it_does_not_apply():
    rng = np.random.default_rng(0)
    P, q, A, l, u = problems.banded_qp(30000, window=30000)        # unstructured columns: no windows
    _, r, s = _solve(P, q, A, l, u, True)
    assert int(s['pcg_fused']) == 1 and int(s['f1_replicas']) == 0
    assert r.info.status_val == osqp_amd.SolverStatus.OSQP_SOLVED


@pytest.mark.parametrize('n,m,window,k,D', [(60000, 60000, 1, 1, 1), (40000, 80000, 20, 3, 2), (40000, 80000, 60, 5, 3), (50000, 100000, 100, 5, 4)])
def test_every_replica_count_has_its_own_kernel_and_agrees_with_the_two_kernel_form(n, m, window, k, D, monkeypatch):
    """The F1 kernels are templates on the replica count D (DevF1::D = how many row blocks apart two blocks must be for their column
    windows not to overlap): one problem per value, each against the two-kernel form of the same engine (tools/f1_replica_sweep.py)."""
    P, q, A, l, u = problems.banded_qp(n, m=m, window=window, nnz_per_row=k)
    res = {}
    for f1 in ('1', '0'):
        monkeypatch.setenv('OSQP_HIP_F1', f1)
        s = osqp_amd.OSQP(); s.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=20000)
        r = s.solve(raise_error=True)
        res[f1] = (r, s._solver.hip_stats())
    (r1, s1), (r0, s0) = res['1'], res['0']
    assert s1['f1_replicas'] == D and s1['pcg_fused'] == 2 and s0['f1_replicas'] == 0
    assert abs(r1.info.iter - r0.info.iter) <= 0.1 * r0.info.iter + 25
    assert np.abs(r1.x - r0.x).max() <= 2e-5 * (1 + np.abs(r0.x).max()) and np.abs(r1.y - r0.y).max() <= 1e-4 * (1 + np.abs(r0.y).max())
    assert abs(r1.info.obj_val - r0.info.obj_val) <= 1e-6 * (1 + abs(r0.info.obj_val))


def test_f1_form_follows_matrix_updates(monkeypatch):
    """update(Px=, Ax=) on a handle that runs the F1 form: the packed copies the form keeps (P + sigma I in its own CSR) are refreshed with the
    re-assembled matrices (re-scaled with the EXISTING scaling, as the reference does: _osqp.py:1443, :1463 -- so the iteration path is that
    of an updated handle, not of a fresh one): same iterations as the two-kernel form after the same update, the solution of a fresh handle
    on the new data."""
    import scipy.sparse as sp
    P, q, A, l, u = problems.banded_qp(20000, window=40)
    rng = np.random.default_rng(7)
    Pt = sp.triu(P, format='csc')
    Px = Pt.data * (1 + 0.05 * rng.random(Pt.nnz))
    Ax = A.data * (1 + 0.05 * rng.standard_normal(A.nnz))
    kw = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=20000)
    res = {}
    for f1 in ('1', '0'):
        monkeypatch.setenv('OSQP_HIP_F1', f1)
        m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, **kw)
        assert m._solver.hip_stats()['pcg_fused'] == (2 if f1 == '1' else 1)
        r0 = m.solve(raise_error=True)
        m.update(Px=Px, Ax=Ax)
        m.update_settings(warm_starting=False, rho=0.1)
        res[f1] = (r0, m.solve(raise_error=True))
    monkeypatch.setenv('OSQP_HIP_F1', '1')
    P2 = sp.csc_matrix((Px, Pt.indices, Pt.indptr), shape=P.shape)
    A2 = sp.csc_matrix((Ax, A.indices, A.indptr), shape=A.shape)
    f = osqp_amd.OSQP(); f.setup(P2, q, A2, l, u, **kw)
    rf = f.solve(raise_error=True)
    (r0, r1), (_, r1b) = res['1'], res['0']
    assert abs(r1.info.iter - r1b.info.iter) <= 0.1 * r1b.info.iter + 25
    for other in (r1b, rf):
        assert _rel(r1.x, other.x) < 2e-5 and _rel(r1.y, other.y) < 1e-4
        assert abs(r1.info.obj_val - other.info.obj_val) <= 1e-6 * (1 + abs(other.info.obj_val))
    assert abs(r1.info.obj_val - r0.info.obj_val) > 1e-6 * (1 + abs(r0.info.obj_val))      # (the update did change the problem)


def _solve_f1(P, q, A, l, u, f1, **kw):
    old = os.environ.get('OSQP_HIP_F1')
    os.environ['OSQP_HIP_F1'] = str(f1)
    try:
        st = dict(eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, adaptive_rho_interval=50, check_termination=25, verbose=False)
        st.update(kw)
        m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, **st)
        r = m.solve()
        return m, r, m._solver.hip_stats()
    finally:
        if old is None:
            os.environ.pop('OSQP_HIP_F1', None)
        else:
            os.environ['OSQP_HIP_F1'] = old


@pytest.mark.parametrize('n,window,frac', [(20000, 40, 0.02), (20000, 40, 0.002), (100000, 200, 0.02)])
def test_band_plus_long_range_couplings_run_the_one_launch_form(n, window, frac):
    """`bench.py --config mixed`: a fraction of A's entries moved to columns drawn from the whole range.  Per-block mixing (backend.h DevF1::mix): a block
    keeps its densest window and takes the other columns as far columns -- the one-launch form applies, with the same results as the two-kernel
    form (OSQP_HIP_F1=2: strict windows, which such a matrix does not have) and as the oracle's direct solve."""
    P, q, A, l, u = problems.banded_qp(n, window=window, long_range=frac)
    m1, r1, s1 = _solve_f1(P, q, A, l, u, 1)
    m0, r0, s0 = _solve_f1(P, q, A, l, u, 2)
    assert int(s1['pcg_fused']) == 2 and 1 <= int(s1['f1_replicas']) <= 4 and s1['f1_far_columns'] > 0, s1
    assert int(s0['pcg_fused']) == 1 and int(s0['f1_replicas']) == 0 and s0['f1_far_columns'] == 0
    assert r1.info.status_val == r0.info.status_val == osqp_amd.SolverStatus.OSQP_SOLVED
    print('n=%d long-range %.3f: %d far columns; mixing: %d iterations, %.1f PCG each, %d launches; two-kernel: %d iterations, %.1f PCG each, %d launches; |dx| %.2e |dy| %.2e'
          % (n, frac, s1['f1_far_columns'], r1.info.iter, s1['pcg_iters_total'] / r1.info.iter, s1['kernel_launches'], r0.info.iter,
             s0['pcg_iters_total'] / r0.info.iter, s0['kernel_launches'], _rel(r1.x, r0.x), _rel(r1.y, r0.y)))
    assert _rel(r1.x, r0.x) < 1e-4 and _rel(r1.y, r0.y) < 1e-4
    assert abs(r1.info.obj_val - r0.info.obj_val) <= 1e-6 * (1 + abs(r0.info.obj_val))
    assert abs(r1.info.iter - r0.info.iter) <= 50
    assert s1['kernel_launches'] < 0.75 * s0['kernel_launches']
    _, rb, _ = _solve_f1(P, q, A, l, u, 1)               # deterministic: fixed summation order, no atomics
    assert rb.info.iter == r1.info.iter and np.array_equal(rb.x, r1.x) and np.array_equal(rb.y, r1.y)
    if n <= 20000:
        kw = dict(eps_abs=1e-8, eps_rel=1e-8, max_iter=50000)
        _, rt, st = _solve_f1(P, q, A, l, u, 1, **kw)
        assert int(st['pcg_fused']) == 2 and st['f1_far_columns'] > 0
        xo, yo, io = Oracle().setup(P, q, A, l, u, adaptive_rho_interval=50, check_termination=25, **kw).solve()
        assert io.status_val == SOLVED and rt.info.status_val == osqp_amd.SolverStatus.OSQP_SOLVED
        print('tight: engine %d iterations, oracle %d; |dx| %.2e |dy| %.2e' % (rt.info.iter, io.iter, _rel(rt.x, xo), _rel(rt.y, yo)))
        assert _rel(rt.x, xo) < 2e-6 and _rel(rt.y, yo) < 4e-6
        assert abs(rt.info.obj_val - io.obj_val) <= 1e-7 * (1 + abs(io.obj_val))


def test_mixing_form_through_updates_warm_start_and_polish():
    """The per-block mixing form behind the rest of the API: data updates (q, bounds, the values of A and P by index), a warm-started re-solve and the
    polish step give what a fresh handle in the two-kernel form gives for the updated problem."""
    n = 20000
    P, q, A, l, u = problems.banded_qp(n, window=40, long_range=0.01)
    m1, r1, s1 = _solve_f1(P, q, A, l, u, 1, polishing=True)
    assert int(s1['pcg_fused']) == 2 and s1['f1_far_columns'] > 0
    rng = np.random.default_rng(3)
    q2 = q + 0.05 * rng.standard_normal(n); l2 = l - 0.1; u2 = u + 0.2
    A2 = A.copy(); A2.data = A2.data * (1.0 + 0.05 * rng.standard_normal(A2.nnz))
    P2 = P.copy(); P2.data = P2.data * 1.1
    old = os.environ.get('OSQP_HIP_F1'); os.environ['OSQP_HIP_F1'] = '1'
    try:
        m1.update(q=q2, l=l2, u=u2)
        m1.update(Ax=A2.data, Px=sp.triu(P2, format='csc').data)
        rw = m1.solve()                                   # warm-started from the first solution
    finally:
        if old is None:
            os.environ.pop('OSQP_HIP_F1', None)
        else:
            os.environ['OSQP_HIP_F1'] = old
    sw = m1._solver.hip_stats()
    assert int(sw['pcg_fused']) == 2 and sw['f1_far_columns'] == s1['f1_far_columns']
    m0, r0, s0 = _solve_f1(P2, q2, A2, l2, u2, 2, polishing=True)
    assert int(s0['pcg_fused']) == 1
    assert rw.info.status_val == r0.info.status_val == osqp_amd.SolverStatus.OSQP_SOLVED
    print('updated problem: mixing (warm) %d iterations, two-kernel (cold) %d; polish %d / %d; |dx| %.2e |dy| %.2e'
          % (rw.info.iter, r0.info.iter, rw.info.status_polish, r0.info.status_polish, _rel(rw.x, r0.x), _rel(rw.y, r0.y)))
    assert _rel(rw.x, r0.x) < 1e-4 and _rel(rw.y, r0.y) < 1e-4
    assert abs(rw.info.obj_val - r0.info.obj_val) <= 1e-6 * (1 + abs(r0.info.obj_val))
    assert rw.info.status_polish == r0.info.status_polish


def test_mixing_plan_on_strided_columns_with_outliers():
    """Columns at a constant stride of 8 inside the band (no run of neighbouring columns at all: the plan's window choice has to fall back to the
    window holding the most entries) plus 1 % long-range entries: whatever form the plan arrives at, the solution is the two-kernel form's."""
    n, m, k = 40000, 80000, 5
    rng = np.random.default_rng(11)
    centre = (np.arange(m, dtype=np.int64) * n) // m
    base = np.clip((centre // 8) * 8 - 80, 0, n - 8 * 24)
    pick = np.sort(np.argsort(rng.random((m, 24)), axis=1)[:, :k], axis=1)
    cols = base[:, None] + 8 * pick
    far = rng.random((m, k)) < 0.01
    cols = np.where(far, rng.integers(0, n, size=(m, k)), cols)
    cols.sort(axis=1)
    dup = (np.diff(cols, axis=1) == 0).any(axis=1)
    cols[dup] = (base[dup, None] + 8 * pick[dup])
    A = sp.csr_matrix((rng.standard_normal(m * k), cols.ravel().astype(np.int32), np.arange(0, m * k + 1, k, dtype=np.int32)), shape=(m, n)).tocsc()
    A.sort_indices()
    P = sp.diags(rng.uniform(0.5, 1.5, n)).tocsc()
    q = rng.standard_normal(n)
    x0 = 0.1 * rng.standard_normal(n); ax0 = A @ x0
    l = ax0 - rng.uniform(0, 1, m); u = ax0 + rng.uniform(0, 1, m)
    m1, r1, s1 = _solve_f1(P, q, A, l, u, 1)
    m0, r0, s0 = _solve_f1(P, q, A, l, u, 2)
    print('strided: form %d, D %d, far %d; %d / %d iterations; |dx| %.2e |dy| %.2e; %.1f / %.1f ms'
          % (s1['pcg_fused'], s1['f1_replicas'], s1['f1_far_columns'], r1.info.iter, r0.info.iter, _rel(r1.x, r0.x), _rel(r1.y, r0.y), s1['gpu_solve_ms'], s0['gpu_solve_ms']))
    assert r1.info.status_val == r0.info.status_val == osqp_amd.SolverStatus.OSQP_SOLVED
    assert _rel(r1.x, r0.x) < 1e-4 and _rel(r1.y, r0.y) < 1e-4
    assert abs(r1.info.obj_val - r0.info.obj_val) <= 1e-6 * (1 + abs(r0.info.obj_val))
