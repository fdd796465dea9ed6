"""Bandwidth-reducing reordering (OSQPHipPolicy::reorder, Engine::compute_reorder / apply_reorder): the engine may work on a permuted copy
of the problem -- P(pc, pc), q(pc), A(pr, pc), l(pr), u(pr) -- so that the one-launch PCG form applies; nothing of that may show at the
API: solutions, certificates, warm starts, vector and matrix updates BY INDEX (reference semantics: /root/reference/src/bindings.cpp.in
:216-281, src/osqp/interface.py:330-401), the scaling read-back all keep the caller's numbering.

CPU tier: OSQP_HIP_REORDER=2 forces the permutation on the host simulator (whose kernels gain nothing from it): every scenario must give
what the un-permuted handle gives.  GPU tier: a banded QP with randomly shuffled variables and constraints is recognised (reordered = 1,
one launch per PCG iteration) and solved to the un-shuffled problem's solution."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import osqp_amd
import problems
from backend_param import engine
from oracle import Oracle, SOLVED
from util import Fixture

warnings.simplefilter('ignore')
S = osqp_amd.SolverStatus
BACKENDS = [pytest.param('hostsim'), pytest.param('hip-pcg', marks=pytest.mark.gpu)]


def _both(monkeypatch, build):
    """build() -> result tuple, once per reorder mode (0: as numbered by the caller, 2: always permuted)"""
    out = []
    for mode in ('0', '2'):
        monkeypatch.setenv('OSQP_HIP_REORDER', mode)
        out.append(build(mode == '2'))
    return out


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('case', ['basic_QP', 'config1_random_qp', 'warm_start', 'polish_random_admm', 'feasibility'])
def test_forced_reordering_is_invisible_at_the_api(case, backend, monkeypatch):
    f = Fixture(case)
    with engine(backend):
        def build(forced):
            m = osqp_amd.OSQP(algebra='hip')
            m.setup(f.P, f.q, f.A, f.l, f.u, **f.hip_settings(eps_abs=1e-9, eps_rel=1e-9, max_iter=20000, cg_max_iter=500, cg_tol_fraction=1e-3, polishing=False))
            assert bool(m._solver.hip_stats()['reordered']) == forced
            r = m.solve()
            D, E, c = m._solver.hip_scaling()
            return r, D, E, c
        (r0, D0, E0, c0), (r1, D1, E1, c1) = _both(monkeypatch, build)
        assert r0.info.status_val == r1.info.status_val == S.OSQP_SOLVED
        assert np.abs(r0.x - r1.x).max() <= 1e-6 * (1 + np.abs(r0.x).max()) and np.abs(r0.y - r1.y).max() <= 1e-6 * (1 + np.abs(r0.y).max())
        assert abs(r0.info.obj_val - r1.info.obj_val) <= 1e-8 * (1 + abs(r0.info.obj_val))
        # Ruiz equilibration commutes with the permutation (norms are maxima; the cost scale is a mean: rounding only)
        np.testing.assert_allclose(D1, D0, rtol=1e-12); np.testing.assert_allclose(E1, E0, rtol=1e-12); assert abs(c1 - c0) <= 1e-12 * c0


@pytest.mark.parametrize('backend', BACKENDS)
def test_updates_by_index_and_warm_start_on_a_reordered_handle(backend, monkeypatch):
    """update_matrices_test.py's scenario (P and A values replaced by index) + vector updates + warm start, permuted vs not"""
    f = Fixture('matrices_update_P_A')
    Pu = sp.triu(f.P, format='csc'); Pu.sort_indices()
    A = f.A.copy(); A.sort_indices()
    rng = np.random.default_rng(1)
    Px_idx = np.arange(0, Pu.nnz, 2, dtype=np.int32); Px_new = Pu.data[Px_idx] * 1.2
    Ax_idx = rng.permutation(A.nnz)[:A.nnz // 2].astype(np.int32); Ax_new = A.data[Ax_idx] * 0.8
    q2 = f.q + 0.1 * rng.standard_normal(f.n); l2 = f.l - 0.1; u2 = f.u + 0.1
    x_w, y_w = rng.standard_normal(f.n), rng.standard_normal(f.m)
    with engine(backend):
        def build(forced):
            m = osqp_amd.OSQP(algebra='hip')
            m.setup(Pu, f.q, A, f.l, f.u, verbose=False, eps_abs=1e-9, eps_rel=1e-9, max_iter=20000, check_termination=1, cg_max_iter=500, cg_tol_fraction=1e-3)
            assert bool(m._solver.hip_stats()['reordered']) == forced
            out = [m.solve()]
            m.update(Px=Px_new, Px_idx=Px_idx, Ax=Ax_new, Ax_idx=Ax_idx); out.append(m.solve())
            m.update(q=q2, l=l2, u=u2); out.append(m.solve())
            m.update(l=f.l - 0.2); out.append(m.solve())                 # (one bound alone: validated against the resident other one)
            m.warm_start(x=x_w, y=y_w); m.update_settings(max_iter=3)   # three iterations from a given point: the start itself must have arrived un-permuted
            out.append(m.solve())
            return out
        a, b = _both(monkeypatch, build)
        for k, (r0, r1) in enumerate(zip(a, b)):
            assert r0.info.status_val == r1.info.status_val, k
            assert np.abs(r0.x - r1.x).max() <= 1e-6 * (1 + np.abs(r0.x).max()), k
            assert np.abs(r0.y - r1.y).max() <= 1e-6 * (1 + np.abs(r0.y).max()), k
        assert a[4].info.iter == b[4].info.iter == 3 and a[4].info.status_val == S.OSQP_MAX_ITER_REACHED
        m = osqp_amd.OSQP(algebra='hip'); m.setup(Pu, f.q, A, f.l, f.u, verbose=False)      # (OSQP_HIP_REORDER is still 2)
        bad = f.l.copy(); bad[3] = f.u[3] + 1.0                          # l > u in ONE row is still refused on a permuted handle (bindings.cpp.in:237: the code is returned)
        assert m._solver.update_data_vec(None, bad, None) == 1 and m._solver.update_data_vec(None, f.l - 1.0, None) == 0


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('case', ['primal_infeasible', 'dual_infeasible_qp'])
def test_certificates_of_a_reordered_handle_keep_the_callers_numbering(case, backend, monkeypatch):
    f = Fixture(case)
    with engine(backend):
        def build(forced):
            m = osqp_amd.OSQP(algebra='hip'); m.setup(f.P, f.q, f.A, f.l, f.u, **f.hip_settings())
            assert bool(m._solver.hip_stats()['reordered']) == forced
            return m.solve()
        r0, r1 = _both(monkeypatch, build)
        assert r0.info.status_val == r1.info.status_val and r0.info.status_val in (S.OSQP_PRIMAL_INFEASIBLE, S.OSQP_DUAL_INFEASIBLE)
        c0, c1 = (r0.prim_inf_cert, r1.prim_inf_cert) if case == 'primal_infeasible' else (r0.dual_inf_cert, r1.dual_inf_cert)
        c0, c1 = c0 / np.abs(c0).max(), c1 / np.abs(c1).max()
        assert np.abs(c0 - c1).max() <= 1e-3


def test_reordering_recovers_a_shuffled_band_symbolically(monkeypatch):
    """CPU tier, symbolic: the permutation the host computes for a shuffled banded QP (n = 20000) gives row blocks whose column windows are
    as narrow as the un-shuffled problem's -- checked through the scaling read-back of a FORCED reorder on the host simulator (the
    simulator does not use the plan; this pins compute_reorder itself)."""
    import ctypes
    from hostsim_util import hostsim
    n = 20000
    P, q, A, l, u = problems.banded_qp(n, window=200)
    rng = np.random.default_rng(0)
    pc, pr = rng.permutation(n), rng.permutation(2 * n)
    Ps = P[pc][:, pc].tocsc(); As = A[pr][:, pc].tocsc(); Ps.sort_indices(); As.sort_indices()
    monkeypatch.setenv('OSQP_HIP_REORDER', '2')
    with hostsim() as h:
        m = osqp_amd.OSQP(algebra='hip'); m.setup(Ps, q[pc], As, l[pr], u[pr], verbose=False, max_iter=5)
        perm_c = np.empty(n, np.int32); perm_r = np.empty(2 * n, np.int32)
        assert h.osqp_hip_get_reordering(m._solver._p, perm_c.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), perm_r.ctypes.data_as(ctypes.POINTER(ctypes.c_int))) == 0
    assert sorted(perm_c) == list(range(n)) and sorted(perm_r) == list(range(2 * n))
    A2 = As.tocsr()[perm_r][:, perm_c].tocsr()                      # the problem as the engine sees it
    widths = []
    for r0 in range(0, 2 * n, 195):                                # ~ 1000-entry row blocks
        c = A2.indices[A2.indptr[r0]:A2.indptr[min(2 * n, r0 + 195)]]
        widths.append(int(c.max() - c.min() + 1))
    assert max(widths) <= 400 and np.mean(widths) <= 320, (max(widths), np.mean(widths))      # as generated: 296 / 287; shuffled: ~ n


@pytest.mark.gpu
def test_shuffled_banded_qp_takes_the_one_launch_form_and_matches_the_unshuffled_solve():
    n = 100000
    P, q, A, l, u = problems.banded_qp(n)
    rng = np.random.default_rng(0)
    pc, pr = rng.permutation(n), rng.permutation(2 * n)
    Ps = P[pc][:, pc].tocsc(); As = A[pr][:, pc].tocsc(); Ps.sort_indices(); As.sort_indices()
    st = dict(eps_abs=1e-8, eps_rel=1e-8, max_iter=50000, check_termination=25, adaptive_rho_interval=50, verbose=False)
    m0 = osqp_amd.OSQP(); m0.setup(P, q, A, l, u, **st); r0 = m0.solve()
    m1 = osqp_amd.OSQP(); m1.setup(Ps, q[pc], As, l[pr], u[pr], **st); r1 = m1.solve()
    s0, s1 = m0._solver.hip_stats(), m1._solver.hip_stats()
    assert s0['reordered'] == 0 and s0['pcg_fused'] == 2
    assert s1['reordered'] == 1 and s1['pcg_fused'] == 2, s1          # band found: one launch per PCG iteration
    assert r0.info.status_val == r1.info.status_val == S.OSQP_SOLVED
    x1 = np.empty(n); x1[pc] = r1.x
    y1 = np.empty(2 * n); y1[pr] = r1.y
    assert np.abs(x1 - r0.x).max() <= 2e-6 * (1 + np.abs(r0.x).max()) and np.abs(y1 - r0.y).max() <= 2e-6 * (1 + np.abs(r0.y).max())
    assert r1.info.iter <= 1.25 * r0.info.iter + 50


@pytest.mark.gpu
def test_reordered_handle_device_pointer_updates_equal_host_pointer_updates():
    """osqp_hip_update_data_vec_device / osqp_hip_warm_start_device on a reordered handle (device-side gathers) against the host-pointer
    entry points (host-side permutation): bit-identical solves; and against the oracle's update() on the shuffled problem."""
    import torch
    n = 20000
    P, q, A, l, u = problems.banded_qp(n, window=40)
    rng = np.random.default_rng(3)
    pc, pr = rng.permutation(n), rng.permutation(2 * n)
    Ps = P[pc][:, pc].tocsc(); As = A[pr][:, pc].tocsc(); Ps.sort_indices(); As.sort_indices()
    qs, ls, us = q[pc], l[pr], u[pr]
    st = dict(eps_abs=1e-7, eps_rel=1e-7, max_iter=50000, check_termination=25, adaptive_rho_interval=50, verbose=False)
    q2 = qs + 0.05 * rng.standard_normal(n); l2, u2 = ls - 0.02, us + 0.03
    xw, yw = 0.1 * rng.standard_normal(n), 0.1 * rng.standard_normal(2 * n)
    res = []
    for dev in (False, True):
        m = osqp_amd.OSQP(); m.setup(Ps, qs, As, ls, us, **st)
        assert m._solver.hip_stats()['reordered'] == 1
        m.solve()
        if dev:
            t = [torch.tensor(v, device='cuda') for v in (q2, l2, u2, xw, yw)]
            torch.cuda.synchronize()
            assert m._solver.hip_update_data_vec_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr()) == 0
            assert m._solver.hip_warm_start_device(t[3].data_ptr(), t[4].data_ptr()) == 0
        else:
            m.update(q=q2, l=l2, u=u2); m.warm_start(x=xw, y=yw)
        m.update_settings(rho=0.1)
        res.append(m.solve())
    assert res[0].info.iter == res[1].info.iter and np.array_equal(res[0].x, res[1].x) and np.array_equal(res[0].y, res[1].y)
    o = Oracle().setup(Ps, q2, As, l2, u2, eps_abs=1e-9, eps_rel=1e-9, max_iter=100000, adaptive_rho_interval=50)
    xo, yo, io = o.solve()
    assert io.status_val == SOLVED
    assert np.abs(res[0].x - xo).max() <= 2e-5 * (1 + np.abs(xo).max()) and np.abs(res[0].y - yo).max() <= 2e-5 * (1 + np.abs(yo).max())


@pytest.mark.gpu
def test_polish_and_settings_updates_on_a_forced_reordered_handle(monkeypatch):
    """polish (the refinement recurrence on the PCG path), update_settings and update_rho on a permuted handle: same polished solution as the
    un-permuted handle (both polish to rounding-level residuals of the same reduced KKT system)."""
    monkeypatch.setenv('OSQP_HIP_SMALL_DIRECT', '0')
    P, q, A, l, u = problems.banded_qp(2000, window=40)
    out = []
    for mode in ('0', '2'):
        monkeypatch.setenv('OSQP_HIP_REORDER', mode)
        m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-4, eps_rel=1e-4, max_iter=20000, verbose=False, polishing=True, adaptive_rho_interval=50)
        assert bool(m._solver.hip_stats()['reordered']) == (mode == '2')
        r = m.solve()
        m.update_settings(rho=0.05, alpha=1.5); r2 = m.solve()
        out.append((r, r2))
    for a, b in zip(out[0], out[1]):
        assert a.info.status_val == b.info.status_val == S.OSQP_SOLVED and a.info.status_polish == b.info.status_polish == 1
        assert np.abs(a.x - b.x).max() <= 1e-8 * (1 + np.abs(a.x).max()) and np.abs(a.y - b.y).max() <= 1e-7 * (1 + np.abs(a.y).max())


@pytest.mark.gpu
def test_matrix_updates_by_index_on_an_automatically_reordered_handle():
    """osqp_update_data_mat with index lists in the CALLER's CSC order (bindings.cpp.in:240-281) on a handle the engine reordered by itself:
    the updated handle must solve what a fresh handle built from the updated matrices solves."""
    n = 50000
    P, q, A, l, u = problems.banded_qp(n)
    rng = np.random.default_rng(11)
    pc, pr = rng.permutation(n), rng.permutation(2 * n)
    Ps = sp.triu(P[pc][:, pc], format='csc'); Ps.sort_indices()
    As = A[pr][:, pc].tocsc(); As.sort_indices()
    qs, ls, us = q[pc], l[pr], u[pr]
    st = dict(eps_abs=1e-8, eps_rel=1e-8, max_iter=50000, check_termination=25, adaptive_rho_interval=50, verbose=False)
    m1 = osqp_amd.OSQP(); m1.setup(Ps, qs, As, ls, us, **st)
    assert m1._solver.hip_stats()['reordered'] == 1 and m1._solver.hip_stats()['pcg_fused'] == 2
    m1.solve()
    Ax_idx = rng.permutation(As.nnz)[:As.nnz // 3].astype(np.int32); Ax_new = As.data[Ax_idx] * (1.0 + 0.05 * rng.standard_normal(Ax_idx.size))
    Px_idx = np.flatnonzero(Ps.indices == np.repeat(np.arange(n), np.diff(Ps.indptr))).astype(np.int32)      # the diagonal entries: scaled up, P stays PSD
    Px_new = Ps.data[Px_idx] * 1.1
    m1.update(Px=Px_new, Px_idx=Px_idx, Ax=Ax_new, Ax_idx=Ax_idx)
    m1.update_settings(rho=0.1); m1.warm_start(x=np.zeros(n), y=np.zeros(2 * n))
    r1 = m1.solve()
    P2 = Ps.copy(); P2.data[Px_idx] = Px_new
    A2 = As.copy(); A2.data[Ax_idx] = Ax_new
    m2 = osqp_amd.OSQP(); m2.setup(P2, qs, A2, ls, us, **st); r2 = m2.solve()
    assert r1.info.status_val == r2.info.status_val == S.OSQP_SOLVED
    assert np.abs(r1.x - r2.x).max() <= 2e-6 * (1 + np.abs(r2.x).max()) and np.abs(r1.y - r2.y).max() <= 2e-6 * (1 + np.abs(r2.y).max())
