"""Pins the oracle (oracle/osqp_oracle.c) BEFORE anything trusts it (CPU only):
   (a) against the reference C core's golden solutions (src/osqp/tests/solutions/*.npz, stored in the
       fixtures as gold_*), with the reference's own tolerances (src/osqp/tests/conftest.py:14-17),
   (b) iterate-for-iterate against the importable pure-python reference solver (ref_* in the fixtures,
       produced by tests/golden/make_fixtures.py): same iteration count, residual traces to 1e-7 rel."""
import numpy as np
import numpy.testing as npt
import pytest

from oracle import Oracle, SOLVED, PRIMAL_INFEASIBLE, DUAL_INFEASIBLE, MAX_ITER_REACHED, NON_CVX
from util import Fixture

ATOL, RTOL, DEC = 1e-3, 1e-4, 4     # reference's builtin/direct row, conftest.py:14-17


def solve_fixture(name, **over):
    f = Fixture(name)
    o = Oracle().setup(f.P, f.q, f.A, f.l, f.u, **f.oracle_settings(**over))
    upd = {k[4:]: f[k] for k in f.raw.files if k.startswith('upd_')}
    if upd:
        o.update(**upd)
    tp, td = o.set_trace(400)
    x, y, info = o.solve()
    return f, o, x, y, info, tp, td


SOLVED_CASES = ['basic_QP', 'basic_update_q', 'basic_update_l', 'basic_update_u', 'basic_update_bounds',
                'matrices_solve', 'matrices_update_P', 'matrices_update_A', 'matrices_update_P_A',
                'feasibility', 'polish_random_admm', 'config1_random_qp', 'warm_start']


@pytest.mark.parametrize('name', SOLVED_CASES)
def test_oracle_matches_c_core_golden_and_purepy(name):
    f, o, x, y, info, tp, td = solve_fixture(name)
    assert info.status_val == SOLVED
    if f.has('gold_x_val') and name != 'polish_random_admm':
        npt.assert_allclose(x, f['gold_x_val'], rtol=RTOL, atol=ATOL)
        npt.assert_allclose(y, f['gold_y_val'], rtol=RTOL, atol=ATOL)
        npt.assert_almost_equal(info.obj_val, float(f['gold_obj']), decimal=DEC)
    # pure-python reference: same algorithm => same iteration count and residual history
    assert f['ref_status'] == SOLVED
    if name != 'matrices_update_P_A':   # purepy's update_P_A forgets the cost scaling c (_osqp.py:1483 vs :1443)
        assert info.iter == int(f['ref_iter'])
        k = min(len(f['ref_trace_pri']), info.iter, 400)
        npt.assert_allclose(tp[:k], f['ref_trace_pri'][:k], rtol=1e-6, atol=1e-9)
        npt.assert_allclose(td[:k], f['ref_trace_dua'][:k], rtol=1e-6, atol=1e-9)
        npt.assert_allclose(x, f['ref_x'], rtol=1e-7, atol=1e-8)
        npt.assert_allclose(y, f['ref_y'], rtol=1e-7, atol=1e-8)
        npt.assert_allclose(info.obj_val, float(f['ref_obj']), rtol=1e-8, atol=1e-9)


def test_oracle_unconstrained_golden():      # unconstrained_test.py:37-45 (purepy cannot run m=0)
    f, o, x, y, info, _, _ = solve_fixture('unconstrained')
    assert info.status_val == SOLVED
    npt.assert_allclose(x, f['gold_x_val'], rtol=RTOL, atol=ATOL)
    npt.assert_almost_equal(info.obj_val, float(f['gold_obj']), decimal=DEC)


def test_oracle_primal_infeasible():         # primal_infeasibility_test.py:25-58
    f, o, x, y, info, _, _ = solve_fixture('primal_infeasible')
    assert info.status_val == PRIMAL_INFEASIBLE == int(f['ref_status'])
    assert info.iter == int(f['ref_iter'])
    cert = o.cert[:f.m]
    assert np.isfinite(cert).all() and np.linalg.norm(cert) > 0
    # certificate property (Appendix A.6): A' dy ~ 0 and u'(dy)+ + l'(dy)- < 0
    dy = cert / np.linalg.norm(cert, np.inf)
    assert np.abs(f.A.T @ dy).max() < 1e-3
    assert f.u @ np.maximum(dy, 0) + f.l @ np.minimum(dy, 0) < 0


def test_oracle_primal_dual_infeasible():
    f, o, x, y, info, _, _ = solve_fixture('primal_dual_infeasible')
    assert info.status_val in (PRIMAL_INFEASIBLE, DUAL_INFEASIBLE)
    assert info.status_val == int(f['ref_status'])


@pytest.mark.parametrize('name,key', [('dual_infeasible_lp', 'gold_lp_normalized_dual_inf_cert_correct'),
                                      ('dual_infeasible_qp', 'gold_qp_normalized_dual_inf_cert_correct')])
def test_oracle_dual_infeasible(name, key):  # dual_infeasibility_test.py:31-72
    f, o, x, y, info, _, _ = solve_fixture(name)
    assert info.status_val == DUAL_INFEASIBLE == int(f['ref_status'])
    assert info.iter == int(f['ref_iter'])
    cert = o.cert[:f.n]
    npt.assert_allclose(cert / np.linalg.norm(cert), f[key], rtol=1e-5, atol=1e-8)


def test_oracle_non_convex():                # non_convex_test.py:22-56
    f = Fixture('non_convex')
    with pytest.raises(ValueError, match='error 4'):     # OSQP_NONCVX_ERROR at setup (inertia of the KKT factor)
        Oracle().setup(f.P, f.q, f.A, f.l, f.u, **f.oracle_settings(sigma=1e-6))
    o = Oracle().setup(f.P, f.q, f.A, f.l, f.u, **f.oracle_settings(sigma=5.0, max_iter=4000))
    x, y, info = o.solve()
    assert info.status_val == NON_CVX


def test_oracle_natural_vs_amd_ordering():
    f = Fixture('config1_random_qp')
    r = []
    for ordering in (0, 1):
        o = Oracle().setup(f.P, f.q, f.A, f.l, f.u, **f.oracle_settings(ordering=ordering))
        r.append(o.solve())
    npt.assert_allclose(r[0][0], r[1][0], rtol=1e-6, atol=1e-8)
    assert abs(r[0][2].iter - r[1][2].iter) <= 2


def test_oracle_pcg_linsys_matches_direct():
    f = Fixture('config1_random_qp')
    xd, yd, idr = Oracle().setup(f.P, f.q, f.A, f.l, f.u, **f.oracle_settings()).solve()
    xp, yp, ip = Oracle().setup(f.P, f.q, f.A, f.l, f.u, **f.oracle_settings(linsys=1, pcg_tol=1e-12)).solve()
    assert ip.status_val == SOLVED and abs(ip.iter - idr.iter) <= 1
    npt.assert_allclose(xp, xd, rtol=1e-6, atol=1e-7)
    npt.assert_allclose(yp, yd, rtol=1e-6, atol=1e-7)


def test_oracle_warm_start():                # warm_start_test.py:25-57
    f = Fixture('warm_start')
    o = Oracle().setup(f.P, f.q, f.A, f.l, f.u, **f.oracle_settings())
    x, y, info = o.solve()
    assert info.iter == int(f['ref_iter'])
    o.warm_start(x=np.zeros(f.n), y=np.zeros(f.m))
    assert o.solve()[2].iter == info.iter
    o.warm_start(x=x, y=y)
    assert o.solve()[2].iter < 10


@pytest.mark.parametrize('name', ['polish_simple', 'polish_random', 'polish_unconstrained'])
def test_oracle_polish_matches_purepy_and_c_core(name):      # polishing_test.py:32-99; _osqp.py:1710-1828
    """The oracle's polish step (dense LU in place of the reference's sparse LU) lands on the pure-python reference's polished
    point to rounding, and on the C core's golden solution within the reference's own tolerances."""
    f = Fixture(name)
    o = Oracle().setup(f.P, f.q, f.A, f.l, f.u, **f.oracle_settings())
    x0, y0, i0 = o.solve()
    assert i0.status_val == SOLVED
    pri0, dua0 = i0.pri_res, i0.dua_res
    x, y, info, status_polish = o.polish(delta=1e-6, polish_refine_iter=3)
    assert status_polish == 1
    assert info.pri_res <= pri0 and info.dua_res < dua0 and info.dua_res < 1e-9
    npt.assert_allclose(x, f['gold_x_val'], rtol=RTOL, atol=ATOL)
    if f.m:
        npt.assert_allclose(y, f['gold_y_val'], rtol=RTOL, atol=ATOL)
    npt.assert_almost_equal(info.obj_val, float(f['gold_obj']), decimal=DEC)
    if f.has('ref_x'):
        assert int(f['ref_status_polish']) == 1
        npt.assert_allclose(x, f['ref_x'], rtol=0, atol=1e-12 * (1 + np.abs(f['ref_x']).max()))
        npt.assert_allclose(y, f['ref_y'], rtol=0, atol=1e-12 * (1 + np.abs(f['ref_y']).max()))
        npt.assert_allclose(info.obj_val, float(f['ref_obj']), rtol=1e-12)


def test_oracle_sparse_polish_equals_the_dense_restatement():
    """oracle_polish_sparse (the reduced KKT matrix assembled into the direct path's LDL' structure: what makes the oracle's polish usable
    at BASELINE sizes) against oracle_polish (dense LU, pinned to the pure-python reference above): same algorithm, 1e-12."""
    import problems
    for gen in (lambda: problems.banded_qp(1200, window=40), lambda: problems.random_qp(), lambda: problems.portfolio_qp(150, 8),
                lambda: problems.banded_qp(60, m=90, window=12, seed=5)):
        P, q, A, l, u = gen()
        out = []
        for sparse in (False, True):
            o = Oracle().setup(P, q, A, l, u, eps_abs=1e-4, eps_rel=1e-4, max_iter=20000, adaptive_rho_interval=50, check_termination=25)
            _, _, io = o.solve()
            assert io.status_val == SOLVED
            xp, yp, ip, st = o.polish(delta=1e-6, polish_refine_iter=3, sparse=sparse)
            _, _, again = o.solve()                        # the ADMM factorisation is intact afterwards
            assert again.status_val == SOLVED
            out.append((xp, yp, st, ip.obj_val))
        (xa, ya, sa, oa), (xb, yb, sb, ob) = out
        assert sa == sb == 1
        assert np.abs(xa - xb).max() <= 1e-12 * (1 + np.abs(xa).max()) and np.abs(ya - yb).max() <= 1e-12 * (1 + np.abs(ya).max())
        assert abs(oa - ob) <= 1e-12 * (1 + abs(oa))
