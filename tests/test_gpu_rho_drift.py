"""GPU tier: how far the DEFAULT adaptive-rho policy of the PCG path (policy.h ctl_rho_rule: the setting's tolerance spent on a square-root scale + a
persistence test) drifts from the reference's rule (/root/reference/src/osqppurepy/_osqp.py:880-930) in ITERATION COUNTS.  Solutions
agree to tolerance whatever the rule (every parity test); the literal rule is one policy field away and counts like purepy
(test_literal_rho_rule.py).  This test BOUNDS the default's drift: over the golden fixtures and BASELINE configs[0], [1] (n = 20k), [3] (small) the
PCG engine's iteration count stays within [0.6, 1.25] x the oracle's (the reference rule, direct solves; one termination check of slack) and it
never applies more than 2 x the oracle's rho updates + 2 (measured: portfolio 1000 x 20 takes 4 where the oracle takes 1 and needs 425 iterations
instead of 675 -- an update costs this path two small kernels, not a refactorisation, which is why the rule fires earlier).  Both counts are recorded in gpurun_out/parity_deviations.json."""
import os
import warnings

import pytest

import osqp_amd
import problems
from oracle import Oracle, SOLVED
from util import Fixture, record_deviation

pytestmark = pytest.mark.gpu
warnings.simplefilter('ignore')
FIXTURES = ['basic_QP', 'matrices_solve', 'config1_random_qp', 'warm_start', 'polish_random_admm', 'feasibility']


def _pcg_engine(P, q, A, l, u, **st):
    old = os.environ.get('OSQP_HIP_SMALL_DIRECT')
    os.environ['OSQP_HIP_SMALL_DIRECT'] = '0'                    # small problems through the multi-kernel PCG engine too
    try:
        m = osqp_amd.OSQP(algebra='hip')
        m.setup(P, q, A, l, u, verbose=False, **st)
        r = m.solve()
        assert m._solver.hip_stats()['kernel_launches'] > 1
        return r
    finally:
        if old is None:
            os.environ.pop('OSQP_HIP_SMALL_DIRECT', None)
        else:
            os.environ['OSQP_HIP_SMALL_DIRECT'] = old


def _check(case, r, io):
    assert r.info.status_val == osqp_amd.SolverStatus.OSQP_SOLVED and io.status_val == SOLVED, (case, r.info.status, io.status_val)
    ratio = r.info.iter / max(io.iter, 1)
    record_deviation('default_rho_rule_drift', case, iters=r.info.iter, oracle_iters=io.iter, ratio=ratio, rho_updates=r.info.rho_updates, oracle_rho_updates=int(io.rho_updates))
    print('%-32s engine %5d iterations / %d rho updates, oracle %5d / %d, ratio %.2f' % (case, r.info.iter, r.info.rho_updates, io.iter, io.rho_updates, ratio))
    # (iteration counts are multiples of check_termination: one check of slack on the short solves)
    assert 0.6 * io.iter - 25 <= r.info.iter <= 1.25 * io.iter + 25, (case, r.info.iter, io.iter)
    assert r.info.rho_updates <= 2 * int(io.rho_updates) + 2, (case, r.info.rho_updates, io.rho_updates)


@pytest.mark.parametrize('case', FIXTURES)
def test_default_rule_on_the_golden_fixtures(case):
    f = Fixture(case)
    st = f.hip_settings(polishing=False)
    r = _pcg_engine(f.P, f.q, f.A, f.l, f.u, **{k: v for k, v in st.items() if k != 'verbose'})
    xo, yo, io = Oracle().setup(f.P, f.q, f.A, f.l, f.u, **f.oracle_settings()).solve()
    _check(case, r, io)


@pytest.mark.parametrize('name,gen,kw', [('configs[0] random_qp', problems.random_qp, {}),
                                         ('configs[1] banded n=20000', problems.banded_qp, dict(n=20000)),
                                         ('configs[3] portfolio 1000x20', problems.portfolio_qp, dict(na=1000, k=20))])
def test_default_rule_on_the_baseline_generators(name, gen, kw):
    P, q, A, l, u = gen(**kw)
    st = dict(eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, adaptive_rho_interval=50, check_termination=25)
    r = _pcg_engine(P, q, A, l, u, **st)
    xo, yo, io = Oracle().setup(P, q, A, l, u, **st).solve()
    _check(name, r, io)
