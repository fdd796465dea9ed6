"""The approximate-termination branch: when max_iter is reached the reference repeats its termination test with
every tolerance multiplied by 10 and reports *_INACCURATE statuses (/root/reference/src/osqppurepy/_osqp.py:1018-1022 the x10,
:1053-1075 the three outcomes, :1264-1266 the call at max_iter; status names bindings.cpp.in:350-360).

For each of the three outcomes the ORACLE is scanned over max_iter = 1, 2, ... (check_termination = 1) for the window of iteration
limits at which it reports the inaccurate status -- residuals / certificate between eps and 10 eps -- and the engine is run with the
limit in the middle of that window: same status as the oracle, through the host simulator (CPU tier), the one-launch direct path
('hip'), the multi-kernel PCG engine ('hip-pcg') and the batch kernel (GPU tier)."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import osqp_amd
import oracle as O
import problems
from backend_param import BACKENDS, engine
from oracle import Oracle

warnings.simplefilter('ignore')
S = osqp_amd.SolverStatus


def _solved_case():
    P, q, A, l, u = problems.random_qp()                      # BASELINE configs[0]
    return (P, q, A, l, u), dict(eps_abs=1e-5, eps_rel=1e-5, eps_prim_inf=1e-4, eps_dual_inf=1e-4), O.SOLVED_INACCURATE, O.SOLVED


def _primal_infeasible_case():
    n, m = 20, 30                                             # box on x + sum(x) >= 3n: infeasible, certificate converges gradually
    rng = np.random.default_rng(3)
    M = sp.random(n, n, 0.3, random_state=1, format='csc')
    P = (M @ M.T + 1e-2 * sp.eye(n)).tocsc()
    q = rng.standard_normal(n)
    A = sp.vstack([sp.random(m - n - 1, n, 0.4, random_state=2), sp.eye(n), sp.csr_matrix(np.ones((1, n)))]).tocsc()
    l = np.r_[-np.ones(m - n - 1), -np.ones(n), 3.0 * n]
    u = np.r_[np.ones(m - n - 1), np.ones(n), 4.0 * n]
    return (P, q, A, l, u), dict(eps_abs=1e-9, eps_rel=1e-9, eps_prim_inf=1e-4, eps_dual_inf=1e-4), O.PRIMAL_INFEASIBLE_INACCURATE, O.PRIMAL_INFEASIBLE


def _dual_infeasible_case():
    n, m = 20, 30                                             # three variables without curvature, one-sided rows: unbounded below
    rng = np.random.default_rng(2)
    P = sp.diags(np.r_[np.ones(n - 3), np.zeros(3)]).tocsc()
    A = sp.vstack([sp.random(m - 5, n, 0.4, random_state=7), sp.eye(n, format='csr')[:5]]).tocsc()
    l = -np.ones(m)
    u = np.full(m, 1e30)
    q = rng.standard_normal(n)
    return (P, q, A, l, u), dict(eps_abs=1e-9, eps_rel=1e-9, eps_prim_inf=1e-4, eps_dual_inf=1e-4), O.DUAL_INFEASIBLE_INACCURATE, O.DUAL_INFEASIBLE


CASES = {'solved': _solved_case, 'primal_infeasible': _primal_infeasible_case, 'dual_infeasible': _dual_infeasible_case}
_window_cache = {}


def oracle_window(name):
    """(problem, settings, inaccurate status, [K_first, K_last]): the iteration limits at which the oracle ends in the inaccurate status."""
    if name not in _window_cache:
        prob, stg, inacc, exact = CASES[name]()
        win = []
        for K in range(1, 600):
            _, _, info = Oracle().setup(*prob, max_iter=K, check_termination=1, adaptive_rho=0, **stg).solve()
            if info.status_val == inacc:
                win.append(K)
            elif info.status_val == exact:
                break
        assert len(win) >= 5 and win == list(range(win[0], win[-1] + 1)), (name, win)
        _window_cache[name] = (prob, stg, inacc, win)
    return _window_cache[name]


def test_oracle_reports_every_inaccurate_status():
    """the checker itself: each window exists, is contiguous, and ends where the exact status begins"""
    for name in CASES:
        _, _, inacc, win = oracle_window(name)
        assert inacc in (O.SOLVED_INACCURATE, O.PRIMAL_INFEASIBLE_INACCURATE, O.DUAL_INFEASIBLE_INACCURATE)
        assert win[0] > 1


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('name', list(CASES))
def test_inaccurate_status_at_max_iter(backend, name):
    prob, stg, inacc, win = oracle_window(name)
    K = (win[0] + win[-1]) // 2
    with engine(backend):
        m = osqp_amd.OSQP(algebra='hip')
        m.setup(*prob, max_iter=K, check_termination=1, adaptive_rho=False, verbose=False, cg_max_iter=200, cg_tol_fraction=0.01, **stg)
        r = m.solve()
        assert r.info.status_val == inacc, (name, backend, K, win[0], win[-1], r.info.status, r.info.iter)   # (the enums agree: bindings.cpp.in:349-361)
        assert r.info.iter == K
        if name == 'solved':
            assert np.isfinite(r.info.obj_val) and np.all(np.isfinite(r.x))
        elif name == 'primal_infeasible':
            assert r.info.obj_val == np.inf or r.info.obj_val >= 1e30
            assert np.all(np.isfinite(r.prim_inf_cert))
        else:
            assert r.info.obj_val == -np.inf or r.info.obj_val <= -1e30
            assert np.all(np.isfinite(r.dual_inf_cert))
        # one iteration past the window's end the exact status takes over (same handle: update_settings)
        m.update_settings(max_iter=win[-1] + 40)
        m.warm_start(x=np.zeros(len(prob[1])), y=np.zeros(len(prob[3])))
        r2 = m.solve()
        assert r2.info.status_val == inacc - 1, (name, backend, r2.info.status)


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(CASES))
def test_inaccurate_status_in_the_batch_kernel(name):
    """the batch kernel carries its own copy of the x10 pass (batch_hip.hip): every member of a batch ends like the oracle"""
    prob, stg, inacc, win = oracle_window(name)
    P, q, A, l, u = prob
    K = (win[0] + win[-1]) // 2
    m = osqp_amd.OSQP(algebra='hip')
    m.setup(P, q, A, l, u, max_iter=K, check_termination=1, adaptive_rho=False, verbose=False, **stg)
    B = 6
    x, y, rec = m._solver.hip_batch_solve(q=np.tile(q, (B, 1)), nbatch=B)
    assert np.all(rec[:, 0] == inacc), rec[:, 0]
    assert np.all(rec[:, 1] == K)
