"""GPU tier: osqp_solve on small QPs = one launch of the batch kernel's direct variant (Engine::solve_small_direct).
Checks that the path is taken, that it reproduces the oracle's direct-LDL' ADMM iteration for iteration (same algorithm,
same rho rule), and the handle semantics around it: warm continuation, updates, rho persistence, certificates, opt-out."""
import warnings

import numpy as np
import numpy.testing as npt
import pytest
import scipy.sparse as sp

import osqp_amd
import problems
from oracle import Oracle, SOLVED, PRIMAL_INFEASIBLE

pytestmark = pytest.mark.gpu
warnings.simplefilter('ignore')
EPS = 1e-6
ST = dict(eps_abs=EPS, eps_rel=EPS, max_iter=20000, adaptive_rho_interval=50, check_termination=25)


def mpc1(seed=3):
    P, q, A, L, U = problems.mpc_batch(1, seed=seed)
    return P, q, A, L[0], U[0]


def launches(m):
    return int(m._solver.hip_stats()['kernel_launches'])


@pytest.mark.parametrize('gen', [mpc1, lambda: problems.banded_qp(150, window=20), lambda: problems.banded_qp(60, m=90, window=12, seed=5)])
def test_one_launch_and_same_iterations_as_oracle(gen):
    P, q, A, l, u = gen()
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, **ST)
    r = m.solve()
    assert launches(m) == 1 and r.info.status_val == 1
    xo, yo, io = Oracle().setup(P, q, A, l, u, **ST).solve()
    assert io.status_val == SOLVED and r.info.iter == io.iter and r.info.rho_updates == io.rho_updates
    npt.assert_allclose(r.x, xo, rtol=0, atol=1e-7 * (1 + np.abs(xo).max()))
    npt.assert_allclose(r.y, yo, rtol=0, atol=1e-7 * (1 + np.abs(yo).max()))
    assert abs(r.info.obj_val - io.obj_val) <= 1e-9 * (1 + abs(io.obj_val))


@pytest.mark.parametrize('n', [57, 64, 65, 72, 100, 117, 128, 129, 136])
def test_sizes_around_the_register_resident_substitutions(n):
    """n <= 128 takes the kernel instantiation whose substitutions keep a lane's two elements (lane, lane + 64) in registers
    (batch_hip.hip ksolve, N128), larger n the windowed general form: sizes on both sides of 64 and of 128, multiples of 8 and
    not -- iteration for iteration against the oracle, and the polish on the same factorisation code."""
    P, q, A, l, u = problems.banded_qp(n, window=14, seed=n)
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, polishing=True, **ST)
    r = m.solve()
    assert launches(m) == 1 and r.info.status_val == 1
    o = Oracle().setup(P, q, A, l, u, **ST)
    xo, yo, io = o.solve()
    assert io.status_val == SOLVED and r.info.iter == io.iter and r.info.rho_updates == io.rho_updates
    xp, yp, ip, sp_ = o.polish()
    assert r.info.status_polish == sp_
    xr, yr = (xp, yp) if sp_ == 1 else (xo, yo)
    tol = 1e-9 if sp_ == 1 else 1e-7
    npt.assert_allclose(r.x, xr, rtol=0, atol=tol * (1 + np.abs(xr).max()))
    npt.assert_allclose(r.y, yr, rtol=0, atol=tol * (1 + np.abs(yr).max()))


def test_opt_out_and_excluded_settings(monkeypatch):
    P, q, A, l, u = mpc1()
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, polishing=True, **ST)
    r = m.solve()
    assert r.info.status_val == 1 and r.info.status_polish == 1       # direct solve and polish in the same launch
    k = problems.kkt_certificate(P, q, A, l, u, r.x, r.y)
    assert k['pri'] <= 1e-7 and k['dua'] <= 1e-7                       # polished: far below eps = 1e-6
    monkeypatch.setenv('OSQP_HIP_SMALL_DIRECT', '0')
    m2 = osqp_amd.OSQP(); m2.setup(P, q, A, l, u, verbose=False, **ST)
    r2 = m2.solve()
    assert launches(m2) > 1 and r2.info.status_val == 1
    monkeypatch.delenv('OSQP_HIP_SMALL_DIRECT')
    m3 = osqp_amd.OSQP(); m3.setup(P, q, A, l, u, verbose=False, **ST)
    r3 = m3.solve()
    assert launches(m3) == 1
    npt.assert_allclose(r3.x, r2.x, rtol=0, atol=2e-5 * (1 + np.abs(r2.x).max()))


def test_handle_state_follows_warm_continuation_updates_rho():
    P, q, A, l, u = mpc1(seed=9)
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, **ST)
    r1 = m.solve()
    assert launches(m) == 1 and r1.info.rho_updates >= 0
    rho_after = m._solver.get_settings().rho                            # the C-side settings (bindings.cpp.in:163)
    r2 = m.solve()                                                       # continues from the solution: done at the first check
    assert r2.info.iter <= 25 and r2.info.status_val == 1
    npt.assert_allclose(r2.x, r1.x, rtol=0, atol=1e-5 * (1 + np.abs(r1.x).max()))
    # update the linear cost, re-solve warm, compare with a fresh cold solve of the new problem
    q2 = q + 0.05 * np.random.default_rng(1).standard_normal(len(q))
    m.update(q=q2)
    r3 = m.solve()
    f = osqp_amd.OSQP(); f.setup(P, q2, A, l, u, verbose=False, **ST)
    r4 = f.solve()
    assert r3.info.status_val == 1 and r3.info.iter <= r4.info.iter
    npt.assert_allclose(r3.x, r4.x, rtol=0, atol=2e-5 * (1 + np.abs(r4.x).max()))
    # explicit warm start with the optimum of the new problem
    m.warm_start(x=r4.x, y=r4.y)
    r5 = m.solve()
    assert r5.info.iter <= 25
    # the adapted rho stays with the handle, as the reference's work.settings.rho does (_osqp.py:923-930)
    assert (rho_after != 0.1) == (r1.info.rho_updates > 0)


def test_infeasible_problem_certificate_through_the_direct_path():
    P, q, A, l, u = mpc1(seed=4)
    l = l.copy(); u = u.copy(); l[:8] += 500.0; u[:8] += 500.0
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, **ST)
    r = m.solve()
    xo, yo, io = Oracle().setup(P, q, A, l, u, **ST).solve()
    assert launches(m) == 1 and io.status_val == PRIMAL_INFEASIBLE and r.info.status_val == PRIMAL_INFEASIBLE
    cert = r.prim_inf_cert / np.abs(r.prim_inf_cert).max()
    assert np.all(np.isnan(r.x)) and np.abs(A.T @ cert).max() < 1e-3 and u @ np.maximum(cert, 0) + l @ np.minimum(cert, 0) < 0


def test_unconstrained_and_tiny_problems():
    n = 5
    P = sp.diags([1.0, 2.0, 0.5, 3.0, 1.5]).tocsc(); q = np.array([1.0, -2.0, 0.3, 0.0, 4.0])
    m = osqp_amd.OSQP(); m.setup(P, q, sp.csc_matrix((0, n)), np.zeros(0), np.zeros(0), verbose=False, eps_abs=1e-9, eps_rel=1e-9)
    r = m.solve()
    assert launches(m) == 1 and r.info.status_val == 1
    npt.assert_allclose(r.x, -q / P.diagonal(), rtol=0, atol=1e-8)
    # n = 1, one constraint
    m = osqp_amd.OSQP(); m.setup(sp.csc_matrix([[2.0]]), np.array([-1.0]), sp.csc_matrix([[1.0]]), np.array([1.0]), np.array([3.0]), verbose=False, eps_abs=1e-9, eps_rel=1e-9)
    r = m.solve()
    assert launches(m) == 1 and r.info.status_val == 1 and abs(r.x[0] - 1.0) < 1e-7 and abs(r.y[0] + 1.0) < 1e-6


def test_baseline_config1_takes_the_direct_path_and_matches_the_python_reference_fixture():
    """BASELINE configs[0] (n = 50, m = 100, dense P = M M' + 0.01 I: 3300 stored entries in B -> the 16-entries-per-lane variant):
    one launch, same iterations as the oracle, solution within the fixture's tolerance."""
    P, q, A, l, u = problems.random_qp()
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, **ST)
    r = m.solve()
    xo, yo, io = Oracle().setup(P, q, A, l, u, **ST).solve()
    assert launches(m) == 1 and r.info.status_val == 1 and io.status_val == SOLVED and r.info.iter == io.iter
    npt.assert_allclose(r.x, xo, rtol=0, atol=1e-7 * (1 + np.abs(xo).max()))
    npt.assert_allclose(r.y, yo, rtol=0, atol=1e-7 * (1 + np.abs(yo).max()))


def test_interrupted_solve_continues_like_the_oracle_and_reports_the_rho_estimate():
    """A solve stopped by max_iter and continued by a second solve() keeps ALL its iterates -- x, y and the z iterate itself, not
    z = A x (_osqp.py:1197-1204) -- so the continuation takes exactly the oracle's iterations; info.rho_estimate is the reference's
    estimate at the final ADMM point (:1275), not the rho in use."""
    P, q, A, l, u = mpc1(seed=6)
    st = dict(ST, max_iter=60)
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, **st)
    o = Oracle().setup(P, q, A, l, u, **st)
    r1 = m.solve(); x1, y1, i1 = o.solve()
    assert launches(m) == 1 and r1.info.status_val == i1.status_val == 7 and r1.info.iter == i1.iter == 60      # MAX_ITER_REACHED
    npt.assert_allclose(r1.info.rho_estimate, i1.rho_estimate, rtol=1e-6)
    m.update_settings(max_iter=20000); o.update_settings(max_iter=20000)
    r2 = m.solve(); x2, y2, i2 = o.solve()
    assert launches(m) == 1 and r2.info.status_val == 1 and i2.status_val == SOLVED
    assert r2.info.iter == i2.iter and r2.info.rho_updates == i2.rho_updates
    npt.assert_allclose(r2.x, x2, rtol=0, atol=1e-8 * (1 + np.abs(x2).max()))
    npt.assert_allclose(r2.y, y2, rtol=0, atol=1e-8 * (1 + np.abs(y2).max()))
    npt.assert_allclose(r2.info.rho_estimate, i2.rho_estimate, rtol=1e-5)
    assert r2.info.rho_estimate != m._solver.get_settings().rho or r2.info.rho_updates == 0
