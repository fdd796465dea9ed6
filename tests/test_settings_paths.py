"""Less-travelled settings of the engine against the oracle (both tiers): scaling on/off, scaled termination, scalar rho,
no preconditioner, alpha change after setup (captured graphs must be rebuilt), time limit, settings validation."""
import warnings

import numpy as np
import numpy.testing as npt
import scipy.sparse as sp
import pytest

import osqp_amd
import problems
from backend_param import BACKENDS, engine
from oracle import Oracle, SOLVED

warnings.simplefilter('ignore')
S = osqp_amd.SolverStatus
P, q, A, l, u = problems.banded_qp(600, window=30, seed=9)
XO, YO, IO = Oracle().setup(P, q, A, l, u, eps_abs=1e-10, eps_rel=1e-10, max_iter=200000, adaptive_rho_interval=50).solve()
assert IO.status_val == SOLVED


def solve(**kw):
    st = dict(eps_abs=1e-7, eps_rel=1e-7, verbose=False, max_iter=50000)
    st.update(kw)
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, **st)
    return m, m.solve()


def close(r, tol=2e-5):
    assert r.info.status_val == S.OSQP_SOLVED
    npt.assert_allclose(r.x, XO, rtol=0, atol=tol * (1 + np.abs(XO).max()))
    npt.assert_allclose(r.y, YO, rtol=0, atol=tol * (1 + np.abs(YO).max()))
    assert abs(r.info.obj_val - IO.obj_val) <= 1e-6 * (1 + abs(IO.obj_val))


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('kw', [dict(scaling=0), dict(scaling=3), dict(scaled_termination=True), dict(rho_is_vec=False),
                                dict(cg_preconditioner=None), dict(adaptive_rho=False, rho=0.3), dict(check_termination=7, adaptive_rho_interval=21),
                                dict(alpha=1.0), dict(sigma=1e-4), dict(cg_max_iter=7), dict(cg_tol_fraction=0.5)],
                         ids=lambda k: ','.join('%s=%s' % kv for kv in k.items()))
def test_setting_variants_reach_the_same_solution(backend, kw):
    with engine(backend):
        m, r = solve(**kw)
        close(r)
        if 'scaled_termination' in kw or kw.get('scaling') == 0:       # residual definitions follow the oracle's for these modes
            o = Oracle().setup(P, q, A, l, u, eps_abs=1e-7, eps_rel=1e-7, max_iter=50000, adaptive_rho_interval=50, check_termination=25,
                               scaling=kw.get('scaling', 10), scaled_termination=int(kw.get('scaled_termination', False)))
            xo, yo, io = o.solve()
            assert io.status_val == SOLVED and abs(np.log10(r.info.prim_res + 1e-30) - np.log10(io.pri_res + 1e-30)) < 3


@pytest.mark.parametrize('backend', BACKENDS)
def test_alpha_update_after_solve_rebuilds_launch_graphs(backend):
    with engine(backend):
        m, r1 = solve(alpha=1.6, warm_starting=False)
        m.update_settings(alpha=1.0)
        r2 = m.solve()
        m2, r3 = solve(alpha=1.0, warm_starting=False, rho=m.settings.rho if False else 0.1)
        close(r2)
        if r1.info.rho_updates == 0:                                  # same state as a fresh alpha=1.0 solver
            assert r2.info.iter == r3.info.iter and np.array_equal(r2.x, r3.x)


@pytest.mark.parametrize('backend', BACKENDS)
def test_time_limit_and_validation(backend):
    with engine(backend):
        m, r = solve(time_limit=1e-9, eps_abs=1e-12, eps_rel=1e-12)
        assert r.info.status_val == S.OSQP_TIME_LIMIT_REACHED and r.info.iter > 0
        for bad in (dict(rho=-1.0), dict(sigma=0.0), dict(alpha=2.5), dict(max_iter=0), dict(eps_abs=-1.0), dict(cg_max_iter=0),
                    dict(adaptive_rho_tolerance=0.5), dict(check_termination=-1), dict(scaling=-2), dict(verbose=3)):
            with pytest.raises(osqp_amd.OSQPException) as ei:
                osqp_amd.OSQP().setup(P, q, A, l, u, **bad)
            assert ei.value == osqp_amd.SolverError.OSQP_SETTINGS_VALIDATION_ERROR, bad
        with pytest.raises(osqp_amd.OSQPException):                   # update_settings validates too (bindings.cpp.in:204-209)
            m.update_settings(eps_rel=-1.0)
        # malformed data: P with a lower-triangular entry straight through the ext layer, wrong dimensions
        ext = m.ext
        import scipy.sparse as sp
        s = ext.OSQPSettings(); ext.osqp_set_default_settings(s)
        bad_P = ext.CSC(sp.csc_matrix(np.array([[1.0, 0.0], [1.0, 1.0]])))
        with pytest.raises(ValueError, match='1'):
            ext.OSQPSolver(bad_P, np.zeros(2), ext.CSC(sp.eye(2, format='csc')), -np.ones(2), np.ones(2), 2, 2, s)
        with pytest.raises(ValueError, match='1'):
            ext.OSQPSolver(ext.CSC(sp.eye(3, format='csc')), np.zeros(2), ext.CSC(sp.eye(2, format='csc')), -np.ones(2), np.ones(2), 2, 2, s)


@pytest.mark.parametrize('backend', BACKENDS)
def test_batch_direct_symbolic_ordering(backend):
    """Host-side symbolic work of the direct batch / small-QP solve (Engine::prepare_batch_direct): reverse Cuthill-McKee on the
    pattern of P + sigma I + A' A.  MPC (horizon 10, nx 8, nu 4): natural bandwidth 87, ordered <= 30; a tridiagonal chain stays 1-2;
    disconnected components are handled; the analysis runs on every backend (the device solve itself only on the GPU)."""
    import problems
    with engine(backend):
        P, q, A, L, U = problems.mpc_batch(2)
        s = osqp_amd.OSQP(); s.setup(P, q, A, L[0], U[0], verbose=False)
        try:
            s._solver.hip_batch_solve(l=L, u=U)
        except ValueError:
            assert backend == 'hostsim'                                     # no batch kernel in the simulator
        bw = s._solver.hip_stats()['batch_direct_bw']
        assert 12 <= bw <= 30, bw
        n = 40                                                             # two disconnected tridiagonal chains
        T = sp.diags([np.ones(n - 1), 2 * np.ones(n), np.ones(n - 1)], [-1, 0, 1]).tolil()
        T[n // 2 - 1, n // 2] = 0; T[n // 2, n // 2 - 1] = 0
        perm = np.random.default_rng(0).permutation(n)
        Pm = sp.csc_matrix(T.tocsr()[perm][:, perm])
        s = osqp_amd.OSQP(); s.setup(Pm, np.ones(n), sp.eye(n, format='csc'), -np.ones(n), np.ones(n), verbose=False)
        try:
            s._solver.hip_batch_solve(q=np.ones((2, n)))
        except ValueError:
            assert backend == 'hostsim'
        assert 1 <= s._solver.hip_stats()['batch_direct_bw'] <= 2


@pytest.mark.parametrize('backend', BACKENDS)
def test_duality_gap_fields_and_check_dualgap(backend):
    """The v1 info fields purepy does not have (bindings.cpp.in:475, 478, 491-492) and the check_dualgap termination test (:442):
    at a solved point the dual objective closes on the primal one, rel_kkt_error is at the tolerance level, the primal-dual
    integral is positive; with check_dualgap the returned gap satisfies eps_abs + eps_rel max(|obj|, |dual obj|)."""
    with engine(backend):
        m, r = solve()
        i = r.info
        assert i.status_val == S.OSQP_SOLVED
        assert abs(i.dual_obj_val - IO.obj_val) <= 1e-4 * (1 + abs(IO.obj_val))
        assert abs(i.duality_gap - (i.obj_val - i.dual_obj_val)) <= 1e-12 * (1 + abs(i.obj_val))
        assert 0 <= i.rel_kkt_error <= 1e-4 and i.primdual_int > 0
        m2, r2 = solve(check_dualgap=True)
        j = r2.info
        assert j.status_val == S.OSQP_SOLVED and j.iter >= i.iter
        assert abs(j.duality_gap) < 1e-7 + 1e-7 * max(abs(j.obj_val), abs(j.dual_obj_val))
        close(r2)
    Ps, qs, As, ls, us = problems.random_qp()                 # a problem of the one-launch direct path (GPU): same fields, computed on the host
    with engine(backend):
        s = osqp_amd.OSQP(); s.setup(Ps, qs, As, ls, us, eps_abs=1e-7, eps_rel=1e-7, verbose=False)
        k = s.solve().info
        assert k.status_val == S.OSQP_SOLVED and abs(k.duality_gap) <= 1e-4 * (1 + abs(k.obj_val)) and 0 <= k.rel_kkt_error <= 1e-3


@pytest.mark.parametrize('backend', BACKENDS)
def test_extrapolated_pcg_start(backend, monkeypatch):
    """the PCG of an ADMM iteration starts from x~ extrapolated along the last step (weight OSQP_HIP_EXTRAP, default 0.9).
    Same fixed point whatever the weight (it only changes where the inner solver starts); fewer inner iterations than the plain
    warm start (weight 0) on a problem whose PCG takes several iterations; the history is cleared by warm_start / updates."""
    if backend == 'hip-pcg':
        pytest.skip('same kernels as hip at this size')
    with engine(backend):
        Pb, qb, Ab, lb, ub = problems.banded_qp(3000, window=60, seed=4)
        out = {}
        for theta in ('0', '0.9'):
            monkeypatch.setenv('OSQP_HIP_EXTRAP', theta)
            monkeypatch.setenv('OSQP_HIP_SMALL_DIRECT', '0')
            m = osqp_amd.OSQP(); m.setup(Pb, qb, Ab, lb, ub, eps_abs=1e-7, eps_rel=1e-7, verbose=False, max_iter=50000)
            r = m.solve()
            assert r.info.status_val == S.OSQP_SOLVED
            out[theta] = (r, m._solver.hip_stats()['pcg_iters_total'])
            # a re-solve from the solution (history cleared by warm_start) stops at the first check
            m.warm_start(x=r.x, y=r.y)
            r2 = m.solve()
            assert r2.info.status_val == S.OSQP_SOLVED and r2.info.iter <= 50
        (r0, pcg0), (r9, pcg9) = out['0'], out['0.9']
        npt.assert_allclose(r9.x, r0.x, rtol=0, atol=2e-5 * (1 + np.abs(r0.x).max()))
        npt.assert_allclose(r9.y, r0.y, rtol=0, atol=2e-5 * (1 + np.abs(r0.y).max()))
        assert pcg9 < 0.95 * pcg0, (pcg0, pcg9)
