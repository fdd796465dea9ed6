"""GPU tier: the polish step of the one-launch (direct) path and of the batch kernel -- the reference's algorithm
(/root/reference/src/osqppurepy/_osqp.py:1710-1828: reduced KKT system of the guessed active set regularised by `delta`,
`polish_refine_iter` refinement steps, normal-cone projection, accept test) run inside the kernel on the banded factor in LDS.
Checked against the oracle's restatement of that step (pinned to the pure-python reference in tests/test_oracle_golden.py)."""
import warnings

import numpy as np
import numpy.testing as npt
import pytest
import scipy.sparse as sp

import osqp_amd
import problems
from oracle import Oracle, SOLVED
from util import Fixture

pytestmark = pytest.mark.gpu
warnings.simplefilter('ignore')


def mpc1(seed=3):
    P, q, A, L, U = problems.mpc_batch(1, seed=seed)
    return P, q, A, L[0], U[0]


def _fixture(name):
    f = Fixture(name)
    Pfull = (f.P + sp.triu(f.P, 1).T).tocsc()                                 # the fixtures hold the upper triangle
    return Pfull, f.q, f.A, f.l, f.u


CASES = {'polish_simple': lambda: _fixture('polish_simple'), 'polish_random': lambda: _fixture('polish_random'),
         'polish_unconstrained': lambda: _fixture('polish_unconstrained'), 'mpc': mpc1,
         'banded150': lambda: problems.banded_qp(150, window=20), 'banded60x90': lambda: problems.banded_qp(60, m=90, window=12, seed=5)}
ST = dict(eps_abs=1e-3, eps_rel=1e-3, max_iter=20000, adaptive_rho_interval=50, check_termination=25)


def both(gen, delta=1e-6, refine=3, **over):
    P, q, A, l, u = gen()
    st = dict(ST); st.update(over)
    m = osqp_amd.OSQP()
    m.setup(P, q, A, l, u, verbose=False, polishing=True, delta=delta, polish_refine_iter=refine, **st)
    r = m.solve()
    o = Oracle().setup(P, q, A, l, u, **st)
    xo, yo, io = o.solve()
    assert io.status_val == SOLVED and r.info.status_val == 1 and r.info.iter == io.iter
    xp, yp, ip, sp_ = o.polish(delta=delta, polish_refine_iter=refine)
    return m, r, (xp, yp, ip, sp_), (P, q, A, l, u)


@pytest.mark.parametrize('case', sorted(CASES))
def test_polish_in_the_kernel_matches_the_oracle(case):
    m, r, (xp, yp, ip, sp_), (P, q, A, l, u) = both(CASES[case])
    assert int(m._solver.hip_stats()['kernel_launches']) == 1                # solve + polish: one launch
    assert r.info.status_polish == sp_ == 1
    npt.assert_allclose(r.x, xp, rtol=0, atol=1e-11 * (1 + np.abs(xp).max()))
    if len(yp):
        npt.assert_allclose(r.y, yp, rtol=0, atol=1e-11 * (1 + np.abs(yp).max()))
    assert abs(r.info.obj_val - ip.obj_val) <= 1e-12 * (1 + abs(ip.obj_val))
    assert r.info.prim_res <= 1e-12 and r.info.dual_res <= 1e-12             # eps = 1e-3 ADMM point -> rounding-level residuals
    k = problems.kkt_certificate(P, q, A, l, u, r.x, r.y)
    assert k['pri'] <= 1e-11 and k['dua'] <= 1e-11
    assert r.info.polish_time > 0 and r.info.run_time >= r.info.solve_time + r.info.polish_time - 1e-12


@pytest.mark.parametrize('delta,refine', [(1e-3, 0), (1e-3, 2), (1e-4, 1), (1e-6, 0)])
def test_delta_and_polish_refine_iter_mean_what_the_reference_says(delta, refine):
    """A coarse regularisation without refinement leaves a visible error; each refinement step removes it -- identically in the
    kernel and in the oracle (same regularised system, same number of steps)."""
    m, r, (xp, yp, ip, sp_), _ = both(CASES['banded150'], delta=delta, refine=refine)
    assert r.info.status_polish == sp_
    npt.assert_allclose(r.x, xp, rtol=0, atol=1e-7 * (1 + np.abs(xp).max()))
    npt.assert_allclose(r.y, yp, rtol=0, atol=1e-6 * (1 + np.abs(yp).max()))
    npt.assert_allclose(r.info.dual_res, ip.dua_res, rtol=0.05, atol=1e-9)
    npt.assert_allclose(r.info.prim_res, ip.pri_res, rtol=0.05, atol=1e-9)


def test_refinement_reduces_the_regularisation_error():
    res = []
    for refine in (0, 1, 3):
        m, r, _, _ = both(CASES['banded150'], delta=1e-3, refine=refine)
        res.append(max(r.info.prim_res, r.info.dual_res))
    assert res[1] < 0.1 * res[0] and res[2] < 0.1 * res[1]


def test_rejected_polish_keeps_the_admm_solution():
    """Accept test of _osqp.py:1786-1793: a polish that does not improve the residuals (here: a regularisation so coarse that
    the polished point is worse than a tight ADMM point) reports status_polish = -1 and returns the ADMM solution untouched."""
    P, q, A, l, u = CASES['banded150']()
    st = dict(ST, eps_abs=1e-9, eps_rel=1e-9)
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, polishing=False, **st)
    r0 = m.solve()
    m2 = osqp_amd.OSQP(); m2.setup(P, q, A, l, u, verbose=False, polishing=True, delta=1e-1, polish_refine_iter=0, **st)
    r = m2.solve()
    o = Oracle().setup(P, q, A, l, u, **st); o.solve()
    assert o.polish(delta=1e-1, polish_refine_iter=0)[3] == -1
    assert r.info.status_polish == -1 and r.info.status_val == 1
    assert np.array_equal(r.x, r0.x) and np.array_equal(r.y, r0.y)
    assert r.info.obj_val == r0.info.obj_val and r.info.prim_res == r0.info.prim_res


def test_batch_polish():
    """Every SOLVED problem of a directly-solved batch is polished in its own workgroup: rec column status_polish, residuals
    of the returned points, and three elements against the oracle."""
    B = 24
    P, q, A, L, U = problems.mpc_batch(B, seed=11)
    s = osqp_amd.OSQP()
    s.setup(P, q, A, L[0], U[0], verbose=False, polishing=True, **ST)
    x, y, rec = s._solver.hip_batch_solve(l=L, u=U)
    F = s._solver.BATCH_FIELDS
    assert rec.shape == (B, len(F))
    assert (rec[:, F.index('status_val')] == 1).all() and (rec[:, F.index('status_polish')] == 1).all()
    assert (rec[:, F.index('polish_time')] > 0).all()
    for b in range(B):
        k = problems.kkt_certificate(P, q, A, L[b], U[b], x[b], y[b])
        assert k['pri'] <= 1e-11 and k['dua'] <= 1e-11
    for b in (0, 7, 23):
        o = Oracle().setup(P, q, A, L[b], U[b], **ST); o.solve()
        xp, yp, ip, sp_ = o.polish()
        assert sp_ == 1
        npt.assert_allclose(x[b], xp, rtol=0, atol=1e-11 * (1 + np.abs(xp).max()))
        npt.assert_allclose(y[b], yp, rtol=0, atol=1e-11 * (1 + np.abs(yp).max()))
    s.update_settings(polishing=False)
    x0, y0, rec0 = s._solver.hip_batch_solve(l=L, u=U)
    assert (rec0[:, F.index('status_polish')] == 0).all()
    k = problems.kkt_certificate(P, q, A, L[0], U[0], x0[0], y0[0])
    assert max(k['pri'], k['dua']) > 1e-6                                     # (the unpolished eps = 1e-3 point, for contrast)


def _rel(a, b):
    return np.abs(a - b).max() / (1 + np.abs(b).max())


@pytest.mark.parametrize('n,window,eps', [(2000, 40, 1e-4), (20000, 40, 1e-4), (100000, 200, 1e-4)])
def test_polish_on_the_pcg_path_matches_the_oracle(n, window, eps):
    """The multi-kernel (PCG) path -- the one BASELINE configs 2-4 take: the reference's refinement recurrence run by the engine's own
    kernels (Engine::polish: proximal method of multipliers = one ADMM iteration with alpha = 1, weight 1 / delta_eff on the active rows
    and the inner system solved to 1e-15) against the oracle's polish (reduced KKT system factorised directly, delta = 1e-6, three
    refinement steps; pinned to the pure-python reference, the sparse variant to the dense one in tests/test_oracle_golden.py).
    Whenever the active-set guess is the optimal one both land on the solution of the same reduced KKT system: 1e-8 in x and y.
    The two sides guess from DIFFERENT ADMM iterates (direct vs inexact inner solves stop at different points); where the oracle's
    guess is off -- its polished residuals then stay at 1e-5, the reference algorithm accepts that as an improvement -- the engine's
    result is certified on its own: KKT residuals of the ORIGINAL problem recomputed on the host at 1e-9.  (At eps = 1e-3 the oracle's
    polish rejects its own result on these problems while the engine's succeeds: tools/polish_pcg_probe.py.)"""
    P, q, A, l, u = problems.banded_qp(n, window=window)
    st = dict(eps_abs=eps, eps_rel=eps, max_iter=20000, adaptive_rho_interval=50, check_termination=25)
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, polishing=True, **st)
    r = m.solve()
    assert int(m._solver.hip_stats()['kernel_launches']) > 1                 # the multi-kernel path
    assert r.info.status_val == 1 and r.info.status_polish == 1
    k = problems.kkt_certificate(P, q, A, l, u, r.x, r.y)
    ax = A @ r.x
    scale_p = 1 + max(np.abs(ax).max(), np.abs(np.clip(ax, l, u)).max())
    scale_d = 1 + max(np.abs(P @ r.x).max(), np.abs(A.T @ r.y).max(), np.abs(q).max())
    assert k['pri'] <= 1e-9 * scale_p and k['dua'] <= 1e-9 * scale_d, k          # (relative to the terms the residuals are differences of)
    assert r.info.prim_res <= 1e-9 * scale_p and r.info.dual_res <= 1e-9 * scale_d
    o = Oracle().setup(P, q, A, l, u, **st)
    xo, yo, io = o.solve()
    assert io.status_val == SOLVED
    xp, yp, ip, sp_ = o.polish(delta=1e-6, polish_refine_iter=3)
    print('n=%d eps=%g: ADMM %d iterations (oracle %d); polish %s in %.1f ms; |dx| %.2e |dy| %.2e; residuals %.1e / %.1e (oracle polish %d: %.1e / %.1e)'
          % (n, eps, r.info.iter, io.iter, r.info.status_polish, 1e3 * r.info.polish_time, _rel(r.x, xp), _rel(r.y, yp), r.info.prim_res, r.info.dual_res, sp_, ip.pri_res, ip.dua_res))
    if sp_ == 1 and ip.pri_res <= 1e-12 and ip.dua_res <= 1e-12:             # the oracle's guess was the optimal active set too
        assert _rel(r.x, xp) < 1e-8 and _rel(r.y, yp) < 1e-8
        assert abs(r.info.obj_val - ip.obj_val) <= 1e-9 * (1 + abs(ip.obj_val))
    else:
        assert n > 2000                                                      # (the small case must be a real comparison)
    if n <= 20000:
        # delta and polish_refine_iter are honoured: a larger delta_eff contracts less per step, more steps repair it -- same fixed point
        m2 = osqp_amd.OSQP(); m2.setup(P, q, A, l, u, verbose=False, polishing=True, delta=1e-2, polish_refine_iter=8, **st)
        r2 = m2.solve()
        assert r2.info.status_polish == 1 and _rel(r2.x, r.x) < 1e-7 and _rel(r2.y, r.y) < 1e-7


def test_polish_after_short_pcg_history_runs_to_the_end():
    """The polish's inner systems (relative tolerance 1e-15) take hundreds of PCG iterations where the ADMM chunks before them took two
    or three: the host must keep feeding the one long iteration (it used to give up after a number of top-ups sized by the prediction,
    the solve came back with a device error and the handle kept the PREVIOUS solve's solution)."""
    seed = 72
    rng = np.random.default_rng(seed)
    P, q, A, l, u = problems.banded_qp(3000, window=60, seed=seed)
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=20000)
    r = m.solve()
    m.update(q=q * (1 + 0.01 * rng.standard_normal(len(q)))); r = m.solve()
    m.update(l=l - 0.05, u=u + 0.05); r = m.solve()
    m.warm_start(x=r.x * 0.9, y=r.y * 0.9); r = m.solve()
    Pt = sp.triu(P, format='csc')
    Px, Ax = Pt.data * (1 + 0.02 * rng.random(Pt.nnz)), A.data * (1 + 0.02 * rng.standard_normal(A.nnz))
    m.update(Px=Px, Ax=Ax)
    m.update_settings(polishing=True)
    r = m.solve(raise_error=True)
    assert r.info.status_polish == 1
    A2 = sp.csc_matrix((Ax, A.indices, A.indptr), shape=A.shape)
    z = A2 @ r.x
    assert max(np.maximum(z - (u + 0.05), 0).max(), np.maximum((l - 0.05) - z, 0).max()) < 1e-9
