"""GPU tier: the spectral form of the batch path's direct solve (engine.hpp BatchSpectral, batch_hip.hip SPEC) against the banded LDL' form it
replaces in the launch (OSQPHipPolicy::batch_variant = 2 forces the banded kernel): the same algorithm with the same rho rule -- equal iteration
counts, x / y to 1e-9 -- for a batch of every size class (one workgroup per CU / two), after a matrix update (V is rebuilt), and for a batch that
holds a problem whose OWN bounds give other constraint classes than the solver's (left to the banded kernel by the marker / second launch)."""
import warnings

import numpy as np
import pytest

import osqp_amd
import problems
from oracle import Oracle, SOLVED

pytestmark = pytest.mark.gpu
warnings.simplefilter('ignore')
ST = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=4000)


def _solver(P, q, A, l, u, banded=False, **kw):
    s = osqp_amd.OSQP()
    st = dict(ST); st.update(kw)
    s.setup(P, q, A, l, u, **st)
    if banded:
        s._solver.set_policy(batch_variant=2)          # "direct256": the banded kernel, no spectral launch
    return s


@pytest.mark.parametrize('B', [1, 64, 1100])          # (B = 1: below kBatchSpectralMin both handles run the banded kernel -- the single-QP path of osqp_solve)
def test_spectral_equals_banded(B):
    P, q, A, L, U = problems.mpc_batch(B)
    xs, ys, rs = _solver(P, q, A, L[0], U[0])._solver.hip_batch_solve(l=L, u=U)
    xb, yb, rb = _solver(P, q, A, L[0], U[0], banded=True)._solver.hip_batch_solve(l=L, u=U)
    assert (rs[:, 0] == 1).all() and (rb[:, 0] == 1).all()
    assert np.array_equal(rs[:, 1], rb[:, 1]) and np.array_equal(rs[:, 6], rb[:, 6])          # iterations, rho updates
    assert np.abs(xs - xb).max() <= 1e-9 * (1 + np.abs(xb).max()) and np.abs(ys - yb).max() <= 1e-9 * (1 + np.abs(yb).max())
    assert np.abs(rs[:, 5] - rb[:, 5]).max() <= 1e-5 * np.abs(rb[:, 5]).max()                 # the rho each problem ended with (a ratio of residuals: amplifies the 1e-12 differences of the solves)


def test_spectral_follows_matrix_updates():
    B = 32
    P, q, A, L, U = problems.mpc_batch(B)
    s = _solver(P, q, A, L[0], U[0])
    x0, y0, r0 = s._solver.hip_batch_solve(l=L, u=U)
    rng = np.random.default_rng(4)
    Ax = A.data.copy(); sel = np.abs(np.abs(Ax) - 1.0) > 1e-12                                 # (keep the +-1 entries of the dynamics / box rows)
    Ax[sel] *= 1 + 0.05 * rng.standard_normal(int(sel.sum()))
    s.update(Ax=Ax)
    x1, y1, r1 = s._solver.hip_batch_solve(l=L, u=U)
    import scipy.sparse as sp
    A2 = sp.csc_matrix((Ax, A.indices, A.indptr), shape=A.shape)
    assert (r1[:, 0] == 1).all() and np.abs(x1 - x0).max() > 1e-6
    for i in (0, 13, 31):
        xo, yo, io = Oracle().setup(P, q, A2, L[i], U[i], eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, adaptive_rho_interval=50, check_termination=25).solve()
        assert io.status_val == SOLVED and int(r1[i, 1]) == io.iter
        assert np.abs(x1[i] - xo).max() <= 1e-7 * (1 + np.abs(xo).max())


def test_problem_with_other_constraint_classes_takes_the_banded_kernel():
    B = 40                                                            # (>= 32: batches below that keep the banded kernel, engine_api.cpp kBatchSpectralMin)
    P, q, A, L, U = problems.mpc_batch(B)
    L = L.copy(); U = U.copy()
    m = L.shape[1]
    # problem 7: one box row becomes an equality (u = l), problem 11: one box row becomes loose -- other classes than the solver's own bounds give
    box = int(np.argmax(U[0] - L[0] > 1.0))
    L[7, box] = U[7, box] = 0.0
    L[11, box] = -1e30; U[11, box] = 1e30
    s = _solver(P, q, A, L[0], U[0])
    x, y, rec = s._solver.hip_batch_solve(l=L, u=U)
    assert (rec[:, 0] == 1).all()
    for i in (0, 7, 11, 39):
        xo, yo, io = Oracle().setup(P, q, A, L[i], U[i], eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, adaptive_rho_interval=50, check_termination=25).solve()
        assert io.status_val == SOLVED and int(rec[i, 1]) == io.iter, (i, rec[i, 1], io.iter)
        assert np.abs(x[i] - xo).max() <= 1e-7 * (1 + np.abs(xo).max())
    xb, yb, rb = _solver(P, q, A, L[0], U[0], banded=True)._solver.hip_batch_solve(l=L, u=U)
    assert np.array_equal(rec[:, 1], rb[:, 1])
    assert np.array_equal(x[7], xb[7]) and np.array_equal(x[11], xb[11])                        # those two WERE solved by the banded kernel: bit-identical
