"""GPU tier: the K form (backend.h DevKf, pcg_hip.hip k_slotk; OSQPHipPolicy::kform, off by default) -- one launch per PCG iteration on the explicit
reduced matrix K = P + sigma I + A' diag(rho) A (the Schur complement of the KKT matrix of /root/reference/src/osqppurepy/_osqp.py:291-301) for a
matrix with UNSTRUCTURED columns (problems.banded_qp(window = n): no row block has a window, the one-launch form on A alone does not apply).
Checked: the form is taken, it agrees with the oracle's direct solve and with the two-kernel form at eps = 1e-8, repeats are bit-identical,
K follows rho updates (inside a device-driven solve) and matrix updates, and a solve needs k + 3 launches per ADMM iteration instead of 2 k + 4."""
import os
import warnings

import numpy as np
import pytest

import osqp_amd
import problems
from oracle import Oracle, SOLVED
from util import record_deviation

pytestmark = pytest.mark.gpu
warnings.simplefilter('ignore')


def _rel(a, b):
    return float(np.abs(a - b).max() / (1 + np.abs(b).max()))


def _solver(P, q, A, l, u, kform, **kw):
    old = os.environ.get('OSQP_HIP_KFORM')
    os.environ['OSQP_HIP_KFORM'] = str(kform)
    try:
        st = dict(eps_abs=1e-8, eps_rel=1e-8, verbose=False, max_iter=50000, adaptive_rho_interval=50, check_termination=25, cg_max_iter=50)
        st.update(kw)
        m = osqp_amd.OSQP(algebra='hip')
        m.setup(P, q, A, l, u, **st)
        return m
    finally:
        if old is None:
            os.environ.pop('OSQP_HIP_KFORM', None)
        else:
            os.environ['OSQP_HIP_KFORM'] = old


def test_kform_matches_the_oracle_direct_solve():
    n = 2000
    P, q, A, l, u = problems.banded_qp(n, window=n)
    m = _solver(P, q, A, l, u, 1)
    r = m.solve(); st = m._solver.hip_stats()
    assert int(st['pcg_fused']) == 3 and st['kform_nnz'] > A.nnz, st
    xo, yo, io = Oracle().setup(P, q, A, l, u, eps_abs=1e-8, eps_rel=1e-8, max_iter=50000, adaptive_rho_interval=50, check_termination=25).solve()
    assert r.info.status_val == osqp_amd.SolverStatus.OSQP_SOLVED and io.status_val == SOLVED
    ex, ey = _rel(r.x, xo), _rel(r.y, yo)
    record_deviation('test_kform_matches_the_oracle_direct_solve', 'unstructured n=%d eps=1e-08' % n, dx_rel=ex, dy_rel=ey, iters=r.info.iter, oracle_iters=io.iter)
    assert ex <= 2e-6 and ey <= 2e-6, (ex, ey)
    assert abs(r.info.obj_val - io.obj_val) <= 1e-7 * (1 + abs(io.obj_val))


@pytest.mark.parametrize('n', [20000])
def test_kform_equals_the_two_kernel_form_with_fewer_launches(n):
    P, q, A, l, u = problems.banded_qp(n, window=n)
    m1 = _solver(P, q, A, l, u, 1); r1 = m1.solve(); s1 = m1._solver.hip_stats()
    m0 = _solver(P, q, A, l, u, 0); r0 = m0.solve(); s0 = m0._solver.hip_stats()
    assert int(s1['pcg_fused']) == 3 and int(s0['pcg_fused']) == 1 and s0['kform_nnz'] == 0
    assert r1.info.status_val == r0.info.status_val == osqp_amd.SolverStatus.OSQP_SOLVED
    assert r1.info.rho_updates > 0                                   # K was refreshed by the boundary groups of a device-driven solve
    ex, ey = _rel(r1.x, r0.x), _rel(r1.y, r0.y)
    record_deviation('test_kform_equals_the_two_kernel_form', 'unstructured n=%d eps=1e-08' % n, dx_rel=ex, dy_rel=ey, iters=r1.info.iter, two_kernel_iters=r0.info.iter,
                     launches=s1['kernel_launches'], two_kernel_launches=s0['kernel_launches'])
    assert ex <= 2e-6 and ey <= 2e-6, (ex, ey)
    assert abs(r1.info.iter - r0.info.iter) <= 50
    assert s1['kernel_launches'] < 0.75 * s0['kernel_launches']
    # bit-identical repeats (cold start, rho back to the setting): the form is a deterministic function of the data
    m1.update_settings(rho=0.1); m1._solver.warm_start(np.zeros(n), np.zeros(len(l)))
    ra = m1.solve()
    m1.update_settings(rho=0.1); m1._solver.warm_start(np.zeros(n), np.zeros(len(l)))
    rb = m1.solve()
    assert ra.info.iter == rb.info.iter and np.array_equal(ra.x, rb.x) and np.array_equal(ra.y, rb.y)


def test_kform_follows_matrix_updates():
    n = 4000
    P, q, A, l, u = problems.banded_qp(n, window=n)
    rng = np.random.default_rng(5)
    Ax = A.data * (1 + 0.05 * rng.standard_normal(A.nnz))
    import scipy.sparse as sp
    A2 = sp.csc_matrix((Ax, A.indices, A.indptr), shape=A.shape)
    m = _solver(P, q, A, l, u, 1, eps_abs=1e-7, eps_rel=1e-7)
    m.solve()
    m.update(Ax=Ax)                                                   # K's values must follow (be::precond -> k_kf_values)
    m.update_settings(rho=0.1)
    r = m.solve()
    assert int(m._solver.hip_stats()['pcg_fused']) == 3
    m2 = _solver(P, q, A2, l, u, 0, eps_abs=1e-7, eps_rel=1e-7)
    r2 = m2.solve()
    assert r.info.status_val == r2.info.status_val == osqp_amd.SolverStatus.OSQP_SOLVED
    k = problems.kkt_certificate(P, q, A2, l, u, r.x, r.y)
    assert _rel(r.x, r2.x) <= 2e-5 and _rel(r.y, r2.y) <= 2e-5, (k, _rel(r.x, r2.x), _rel(r.y, r2.y))


def test_dense_rows_keep_the_form_off():
    """fill gate (backend.h kKfMaxFill): a matrix whose rows are long (sum of squared row lengths > 8 nnz(A)) never builds K"""
    P, q, A, l, u = problems.lasso_qp(60, 120)
    m = _solver(P, q, A, l, u, 1, eps_abs=1e-6, eps_rel=1e-6)
    m._solver.set_policy(small_direct=0)
    r = m.solve()
    assert r.info.status_val == osqp_amd.SolverStatus.OSQP_SOLVED and m._solver.hip_stats()['kform_nnz'] == 0
