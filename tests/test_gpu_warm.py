"""GPU tier: parametric re-solves (update(q | l, u) + solve() from the previous solution: the reference's main use pattern,
src/osqp/nn/torch.py:136-140, interface.py:399-401) -- the engine's warm iteration counts against the ORACLE's on the same sequence of
1 % data changes.  The reference algorithm itself gains little from a warm start on these QPs at eps = 1e-6 (a 1 % change of q moves
the solution by ~1 %: four decades of error to remove instead of six; tools/warm_probe.py prints the oracle's counts: 225-775 warm
against 350 cold on the banded QP), so the bar is the oracle's own warm count, not a fraction of the cold one."""
import warnings

import numpy as np
import pytest

import osqp_amd
import problems
from oracle import Oracle, SOLVED

pytestmark = pytest.mark.gpu
warnings.simplefilter('ignore')


def _mpc():
    P, q, A, L, U = problems.mpc_batch(1, seed=3)
    return P, q, A, L[0], U[0]


@pytest.mark.parametrize('name,gen,pcg_only', [('banded2000', lambda: problems.banded_qp(2000, window=40), False), ('mpc', _mpc, False), ('mpc-pcg', _mpc, True)])
def test_warm_resolves_take_no_more_iterations_than_the_oracles(name, gen, pcg_only):
    P, q, A, l, u = gen()
    st = dict(eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, check_termination=25, adaptive_rho_interval=50)
    o = Oracle().setup(P, q, A, l, u, **st)
    _, _, io = o.solve()
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, warm_starting=True, **st)
    if pcg_only:
        m._solver.set_policy(small_direct=0)
    r = m.solve()
    assert io.status_val == SOLVED and r.info.status_val == 1
    rng = np.random.default_rng(0)
    tot_o = tot_e = 0
    for rep in range(6):
        if rep % 2 == 0:
            q2 = q * (1 + 0.01 * rng.standard_normal(len(q))); o.update(q=q2); m.update(q=q2)
        else:
            d = 0.01 * rng.random(len(l)); o.update(l=l - d, u=u + d); m.update(l=l - d, u=u + d)
        xo, yo, io = o.solve(); r = m.solve()
        assert io.status_val == SOLVED and r.info.status_val == 1
        print('%s re-solve %d: oracle %d iterations, engine %d' % (name, rep, io.iter, r.info.iter))
        assert r.info.iter <= 1.25 * io.iter + 25, (rep, r.info.iter, io.iter)           # (+ one termination-check interval)
        assert np.abs(r.x - xo).max() <= 1e-4 * (1 + np.abs(xo).max())
        tot_o += io.iter; tot_e += r.info.iter
    assert tot_e <= 1.25 * tot_o
