"""GPU tier: the batch path's wave-per-problem kernel (batch_hip.hip k_batch_wave: V and the matrices once per CU in LDS, a problem's iterates in the registers
of ONE wave, eight problems in flight per CU; OSQPHipPolicy::batch_wave: 1 forces it at any batch size, -1 switches it off, 0 = large batches only) against
the workgroup-per-problem spectral kernel and the banded LDL' kernel: the same algorithm with the same rho rule -- equal
iteration counts, x / y to 1e-9 -- also for warm starts, after a matrix update, for a problem with other constraint classes (left to the banded kernel),
for infeasible problems (certificates) and against the CPU oracle per problem."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import osqp_amd
import problems
from oracle import Oracle, SOLVED

pytestmark = pytest.mark.gpu
warnings.simplefilter('ignore')
ST = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=4000)


def _solver(P, q, A, l, u, wave=1, banded=False, **kw):
    s = osqp_amd.OSQP()
    st = dict(ST); st.update(kw)
    s.setup(P, q, A, l, u, **st)
    s._solver.set_policy(batch_wave=wave, **({'batch_variant': 2} if banded else {}))
    return s


@pytest.mark.parametrize('B', [40, 700, 4096])
def test_wave_equals_workgroup_forms(B):
    P, q, A, L, U = problems.mpc_batch(B)
    xw, yw, rw = _solver(P, q, A, L[0], U[0], wave=1)._solver.hip_batch_solve(l=L, u=U)
    xs, ys, rs = _solver(P, q, A, L[0], U[0], wave=-1)._solver.hip_batch_solve(l=L, u=U)
    assert (rw[:, 0] == 1).all() and (rs[:, 0] == 1).all()
    assert np.array_equal(rw[:, 1], rs[:, 1]) and np.array_equal(rw[:, 6], rs[:, 6])          # iterations, rho updates
    assert np.abs(xw - xs).max() <= 1e-9 * (1 + np.abs(xs).max()) and np.abs(yw - ys).max() <= 1e-9 * (1 + np.abs(ys).max())
    assert np.abs(rw[:, 2] - rs[:, 2]).max() <= 1e-9 * (1 + np.abs(rs[:, 2]).max())           # objective
    if B <= 700:
        xb, yb, rb = _solver(P, q, A, L[0], U[0], wave=-1, banded=True)._solver.hip_batch_solve(l=L, u=U)
        assert np.array_equal(rw[:, 1], rb[:, 1]) and np.abs(xw - xb).max() <= 1e-9 * (1 + np.abs(xb).max())


def test_wave_default_rule_and_repeats():
    B = 1024
    P, q, A, L, U = problems.mpc_batch(B)
    s = osqp_amd.OSQP(); s.setup(P, q, A, L[0], U[0], **ST)
    assert s._solver.get_policy()['batch_wave'] == 0
    xd, yd, rd = s._solver.hip_batch_solve(l=L, u=U)
    x0, y0, r0 = _solver(P, q, A, L[0], U[0], wave=-1)._solver.hip_batch_solve(l=L, u=U)
    assert np.array_equal(xd, x0) and np.array_equal(rd[:, :7], r0[:, :7])                      # 1024 problems: the default is the workgroup kernel
    sw = _solver(P, q, A, L[0], U[0], wave=1)
    x1, y1, r1 = sw._solver.hip_batch_solve(l=L, u=U)           # (no launch order yet: every problem on the wave kernel)
    x2, y2, r2 = sw._solver.hip_batch_solve(l=L, u=U)           # (launch order from the first call's iteration counts: other waves get other problems, and the
    x3, y3, r3 = sw._solver.hip_batch_solve(l=L, u=U)           #  longest-expected ones go to the workgroup kernel on the second stream -- batch_hip.hip batch_solve)
    assert np.array_equal(x2, x3) and np.array_equal(y2, y3) and np.array_equal(r2[:, :7], r3[:, :7])      # same order, same routing: bit-identical
    assert np.array_equal(r1[:, 1], r2[:, 1]) and np.abs(x1 - x2).max() <= 1e-9 * (1 + np.abs(x2).max())  # another routing: the kernels agree to rounding
    assert np.array_equal(r1[:, 1], rd[:, 1]) and np.abs(x1 - xd).max() <= 1e-9 * (1 + np.abs(xd).max())


def test_wave_against_the_oracle_per_problem():
    B = 64
    P, q, A, L, U = problems.mpc_batch(B)
    x, y, rec = _solver(P, q, A, L[0], U[0], wave=1)._solver.hip_batch_solve(l=L, u=U)
    assert (rec[:, 0] == 1).all()
    for i in (0, 5, 31, 63):
        xo, yo, io = Oracle().setup(P, q, A, L[i], U[i], eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, adaptive_rho_interval=50, check_termination=25).solve()
        assert io.status_val == SOLVED and int(rec[i, 1]) == io.iter, (i, rec[i, 1], io.iter)
        assert np.abs(x[i] - xo).max() <= 1e-7 * (1 + np.abs(xo).max()) and np.abs(y[i] - yo).max() <= 1e-7 * (1 + np.abs(yo).max())
        assert abs(rec[i, 2] - io.obj_val) <= 1e-7 * (1 + abs(io.obj_val))


def test_wave_warm_start_and_per_problem_q():
    B = 96
    P, q, A, L, U = problems.mpc_batch(B)
    rng = np.random.default_rng(2)
    Q = q[None, :] + 0.05 * rng.standard_normal((B, q.size))
    sw, ss = _solver(P, q, A, L[0], U[0], wave=1), _solver(P, q, A, L[0], U[0], wave=-1)
    xw, yw, rw = sw._solver.hip_batch_solve(q=Q, l=L, u=U)
    xs, ys, rs = ss._solver.hip_batch_solve(q=Q, l=L, u=U)
    assert (rw[:, 0] == 1).all() and np.array_equal(rw[:, 1], rs[:, 1])
    assert np.abs(xw - xs).max() <= 1e-9 * (1 + np.abs(xs).max())
    # warm start from the solution: done at the first check
    xw2, yw2, rw2 = sw._solver.hip_batch_solve(q=Q, l=L, u=U, x0=xw, y0=yw)
    xs2, ys2, rs2 = ss._solver.hip_batch_solve(q=Q, l=L, u=U, x0=xs, y0=ys)
    assert (rw2[:, 0] == 1).all() and np.array_equal(rw2[:, 1], rs2[:, 1]) and rw2[:, 1].max() <= 50
    assert np.abs(xw2 - xs2).max() <= 1e-9 * (1 + np.abs(xs2).max())


def test_wave_follows_matrix_updates():
    B = 48
    P, q, A, L, U = problems.mpc_batch(B)
    s = _solver(P, q, A, L[0], U[0], wave=1)
    x0, y0, r0 = s._solver.hip_batch_solve(l=L, u=U)
    rng = np.random.default_rng(4)
    Ax = A.data.copy(); sel = np.abs(np.abs(Ax) - 1.0) > 1e-12
    Ax[sel] *= 1 + 0.05 * rng.standard_normal(int(sel.sum()))
    s.update(Ax=Ax)
    x1, y1, r1 = s._solver.hip_batch_solve(l=L, u=U)
    A2 = sp.csc_matrix((Ax, A.indices, A.indptr), shape=A.shape)
    assert (r1[:, 0] == 1).all() and np.abs(x1 - x0).max() > 1e-6
    for i in (0, 13, 47):
        xo, yo, io = Oracle().setup(P, q, A2, L[i], U[i], eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, adaptive_rho_interval=50, check_termination=25).solve()
        assert io.status_val == SOLVED and int(r1[i, 1]) == io.iter
        assert np.abs(x1[i] - xo).max() <= 1e-7 * (1 + np.abs(xo).max())


def test_wave_leaves_other_constraint_classes_to_the_banded_kernel():
    B = 40
    P, q, A, L, U = problems.mpc_batch(B)
    L = L.copy(); U = U.copy()
    box = int(np.argmax(U[0] - L[0] > 1.0))
    L[7, box] = U[7, box] = 0.0
    L[11, box] = -1e30; U[11, box] = 1e30
    x, y, rec = _solver(P, q, A, L[0], U[0], wave=1)._solver.hip_batch_solve(l=L, u=U)
    xb, yb, rb = _solver(P, q, A, L[0], U[0], wave=-1, banded=True)._solver.hip_batch_solve(l=L, u=U)
    assert (rec[:, 0] == 1).all() and np.array_equal(rec[:, 1], rb[:, 1])
    assert np.array_equal(x[7], xb[7]) and np.array_equal(x[11], xb[11])                        # those two WERE solved by the banded kernel: bit-identical


def test_wave_infeasible_problem_in_the_batch():
    B = 36
    P, q, A, L, U = problems.mpc_batch(B)
    L = L.copy(); U = U.copy()
    # problem 3: two box rows of the same variable contradict the dynamics' reachable set -> primal infeasible: both kernels must report the same thing
    box = np.flatnonzero(U[0] - L[0] > 1.0)
    L[3, box[:4]] = 50.0; U[3, box[:4]] = 60.0
    xw, yw, rw = _solver(P, q, A, L[0], U[0], wave=1)._solver.hip_batch_solve(l=L, u=U)
    xs, ys, rs = _solver(P, q, A, L[0], U[0], wave=-1)._solver.hip_batch_solve(l=L, u=U)
    assert np.array_equal(rw[:, 0], rs[:, 0]) and np.array_equal(rw[:, 1], rs[:, 1])
    ok = rw[:, 0] == 1
    assert ok.sum() >= B - 1 and np.abs(xw[ok] - xs[ok]).max() <= 1e-9 * (1 + np.abs(xs[ok]).max())
    if not ok[3]:
        assert np.allclose(yw[3], ys[3], rtol=1e-6, atol=1e-9, equal_nan=True)


def test_wave_is_the_default_for_large_batches():
    B = 4096
    P, q, A, L, U = problems.mpc_batch(B)
    s = osqp_amd.OSQP(); s.setup(P, q, A, L[0], U[0], **ST)
    xd, yd, rd = s._solver.hip_batch_solve(l=L, u=U)
    xw, yw, rw = _solver(P, q, A, L[0], U[0], wave=1)._solver.hip_batch_solve(l=L, u=U)
    assert (rd[:, 0] == 1).all() and np.array_equal(xd, xw) and np.array_equal(yd, yw) and np.array_equal(rd[:, :7], rw[:, :7])


@pytest.mark.parametrize('shape', ['random30', 'banded60', 'mpc_small'])
def test_wave_on_other_patterns(shape):
    """other sizes and sparsity patterns than the MPC batch: n far below the padded size of the LDS copy of V, fewer rows than lane slots, rows of A of
    mixed lengths (the ELL groups), per-problem q -- against the workgroup kernel and the oracle"""
    B = 48
    rng = np.random.default_rng(11)
    if shape == 'mpc_small':
        P, q, A, L, U = problems.mpc_batch(B, nx=3, nu=2, N=4)
        Q = np.repeat(q[None, :], B, axis=0)
    else:
        P, q, A, l, u = problems.random_qp(30, 50, density=0.15, seed=5) if shape == 'random30' else problems.banded_qp(60, window=8, seed=5)
        n, m = P.shape[0], A.shape[0]
        l = np.maximum(l, -1e3); u = np.minimum(u, 1e3)
        ineq = (u - l > 1e-3)[None, :]                                     # (equality rows stay equalities: the classes V was built for)
        L = l[None, :] - 0.3 * rng.random((B, m)) * ineq; U = u[None, :] + 0.3 * rng.random((B, m)) * ineq
        Q = q[None, :] + 0.1 * rng.standard_normal((B, n))
    sw, ss = _solver(P, q, A, L[0], U[0], wave=1), _solver(P, q, A, L[0], U[0], wave=-1)
    xw, yw, rw = sw._solver.hip_batch_solve(q=Q, l=L, u=U)
    xs, ys, rs = ss._solver.hip_batch_solve(q=Q, l=L, u=U)
    assert sw._solver.hip_stats()['batch_wave_split'] >= 0, 'the wave form does not apply to this pattern: the test would compare a kernel with itself'
    assert ss._solver.hip_stats()['batch_wave_split'] == -1
    assert np.array_equal(rw[:, 0], rs[:, 0]) and (rw[:, 0] == 1).all()
    assert np.array_equal(rw[:, 1], rs[:, 1]) and np.array_equal(rw[:, 6], rs[:, 6])
    assert np.abs(xw - xs).max() <= 1e-9 * (1 + np.abs(xs).max()) and np.abs(yw - ys).max() <= 1e-9 * (1 + np.abs(ys).max())
    for i in (0, 17, 47):
        xo, yo, io = Oracle().setup(P, Q[i], A, L[i], U[i], eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, adaptive_rho_interval=50, check_termination=25).solve()
        # (the kernels' common trajectory may pass a termination check the oracle's misses by rounding, or the other way round: one check interval apart)
        assert io.status_val == SOLVED and abs(int(rw[i, 1]) - io.iter) <= 25, (i, rw[i, 1], io.iter)
        assert np.abs(xw[i] - xo).max() <= 1e-4 * (1 + np.abs(xo).max()) and abs(rw[i, 2] - io.obj_val) <= 1e-5 * (1 + abs(io.obj_val))
