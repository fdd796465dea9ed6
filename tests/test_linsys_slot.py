"""The LinSysSolver slot (include/osqp_hip.h: OSQPHipLinSysSolver; north_star's second boundary) against a
direct solve of the reference's quasi-definite KKT system (osqppurepy/_osqp.py:286-311, update_xz_tilde :644-658):

    [[P + sigma I, A'], [A, -diag(1/rho)]] [x~; nu] = [rhs_x; rhs_z],   z~ = rhs_z + nu / rho

on the host simulator (CPU tier: the driver logic) and on the MI355X (gpu tier)."""
import numpy as np
import numpy.testing as npt
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import problems
from backend_param import BACKENDS, engine


def kkt_reference(P, A, rho, sigma, b):
    n, m = P.shape[0], A.shape[0]
    Pf = sp.csc_matrix(P); Pf = sp.triu(Pf) + sp.triu(Pf, 1).T
    K = sp.bmat([[Pf + sigma * sp.eye(n), A.T], [A, -sp.diags(1.0 / rho)]], format='csc')
    s = spla.spsolve(K, b)
    return np.concatenate([s[:n], b[n:] + s[n:] / rho])


def make(name):
    P, q, A, l, u = {'banded': lambda: problems.banded_qp(600, window=30, seed=4),
                     'random': lambda: problems.random_qp(40, 70, seed=2),
                     'lasso': lambda: problems.lasso_qp(30, 150)}[name]()
    return sp.csc_matrix(P), sp.csc_matrix(A)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('name', ['banded', 'random', 'lasso'])
def test_solve_matches_direct_kkt_solve(backend, name):
    with engine(backend):
        from osqp_amd.linsys import LinSysSolver
        P, A = make(name)
        n, m = P.shape[0], A.shape[0]
        rng = np.random.default_rng(1)
        rho = np.where(rng.random(m) < 0.2, 10.0, 0.1) * (0.5 + rng.random(m))
        sigma = 1e-6
        ls = LinSysSolver(P, A, rho, sigma=sigma, polishing=True, cg_max_iter=1000)       # polishing: tight relative tolerance
        assert 'PCG' in ls.name and ls.adjoint_derivative() != 0           # adjoint derivatives: out of scope, says so
        for rep in range(3):                                                             # repeated solves warm-start from the last x~
            b = rng.standard_normal(n + m)
            ref = kkt_reference(P, A, rho, sigma, b)
            out = ls.solve(b, admm_iter=rep + 1)
            scale = 1 + np.abs(ref).max()
            npt.assert_allclose(out[:n], ref[:n], rtol=0, atol=2e-7 * scale)
            npt.assert_allclose(out[n:], ref[n:], rtol=0, atol=2e-7 * scale)
            assert ls.pcg_iters > 0
        ls.free()


@pytest.mark.parametrize('backend', BACKENDS)
def test_rho_and_matrix_updates_and_warm_start(backend):
    with engine(backend):
        from osqp_amd.linsys import LinSysSolver
        P, A = make('banded')
        n, m = P.shape[0], A.shape[0]
        rng = np.random.default_rng(7)
        rho = np.full(m, 0.1)
        ls = LinSysSolver(P, A, rho, sigma=1e-6, polishing=True, cg_max_iter=2000)
        b = rng.standard_normal(n + m)
        ref = kkt_reference(P, A, rho, 1e-6, b)
        out = ls.solve(b)
        it_cold = ls.pcg_iters
        npt.assert_allclose(out, ref, rtol=0, atol=2e-7 * (1 + np.abs(ref).max()))
        # same system again, started from its solution: (almost) nothing left to do
        ls.warm_start(ref[:n])
        out = ls.solve(b)
        assert ls.pcg_iters <= max(2, it_cold // 4), (ls.pcg_iters, it_cold)
        # new rho_vec (an adaptive-rho step of the caller) and new matrix values with the same pattern
        rho2 = rho * np.where(rng.random(m) < 0.5, 7.0, 1.0)
        ls.update_rho_vec(rho2)
        P2 = P.copy(); P2.data = P2.data * 1.0; P2 = P2 + sp.diags(0.3 * rng.random(n))
        A2 = A.copy(); A2.data = A2.data * (1 + 0.1 * rng.standard_normal(A2.nnz))
        ls.update_matrices(P=P2, A=A2)
        ref2 = kkt_reference(P2, A2, rho2, 1e-6, b)
        out2 = ls.solve(b)
        npt.assert_allclose(out2, ref2, rtol=0, atol=2e-7 * (1 + np.abs(ref2).max()))
        ls.free()


@pytest.mark.parametrize('backend', BACKENDS)
def test_tolerance_follows_callers_dual_residual(backend):
    """With the residual pointers given, a solve stops at cg_tol_fraction * (scaled dual residual): looser residual, fewer
    PCG iterations, and the achieved residual of the reduced system obeys the bound."""
    with engine(backend):
        from osqp_amd.linsys import LinSysSolver
        P, A = make('banded')
        n, m = P.shape[0], A.shape[0]
        rng = np.random.default_rng(9)
        rho = np.full(m, 0.1); sigma = 1e-6
        res = np.array([1.0, 1e-2])
        ls = LinSysSolver(P, A, rho, sigma=sigma, scaled_residuals=res, cg_max_iter=2000, cg_tol_fraction=0.15)
        Pf = sp.triu(P) + sp.triu(P, 1).T
        K = (Pf + sigma * sp.eye(n) + A.T @ sp.diags(rho) @ A).tocsr()
        its = []
        for dual in (1e-2, 1e-6):
            res[1] = dual
            ls.warm_start(np.zeros(n))
            b = rng.standard_normal(n + m)
            out = ls.solve(b)
            r = b[:n] + A.T @ (rho * b[n:]) - K @ out[:n]
            assert np.abs(r).max() <= 0.15 * dual * 1.0001
            npt.assert_allclose(out[n:], A @ out[:n], rtol=0, atol=1e-12 * (1 + np.abs(out).max()))
            its.append(ls.pcg_iters)
        assert its[0] < its[1]
        ls.free()
