"""GPU tier: independent solver handles driven concurrently from Python threads (the reference's threading contract:
src/osqp/tests/multithread_test.py:38-53, nn/torch.py:200-217; solve() runs with the GIL released)."""
import warnings
from multiprocessing.pool import ThreadPool

import numpy as np
import pytest

import osqp_amd
import problems

pytestmark = pytest.mark.gpu
warnings.simplefilter('ignore')


def _solve(seed):
    P, q, A, l, u = problems.banded_qp(3000, window=60, seed=seed)
    m = osqp_amd.OSQP()
    m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=20000)
    out = []
    for _ in range(2):                       # two solves per handle: graph replay from a worker thread as well
        m.update_settings(warm_starting=False)
        r = m.solve()
        out.append((r.info.status_val, r.info.iter, r.x.copy(), r.y.copy()))
    return out


@pytest.mark.parametrize('graph', ['1', '0'])
def test_concurrent_handles_give_identical_results(graph, monkeypatch):
    """Bitwise the serial results, also with eager launches (OSQP_HIP_GRAPH=0: the form that exposed a racy flag read in front of a
    workgroup barrier -- 4 % of the concurrent solves differed from the serial ones before it was removed; tools/thread_stress.py)."""
    monkeypatch.setenv('OSQP_HIP_GRAPH', graph)
    seeds = list(range(40, 48))
    serial = [_solve(s) for s in seeds]
    for _ in range(4):
        with ThreadPool(4) as pool:
            threaded = pool.map(_solve, seeds)
        for a, b in zip(serial, threaded):
            for (sa, ia, xa, ya), (sb, ib, xb, yb) in zip(a, b):
                assert sa == sb == 1 and ia == ib
                assert np.array_equal(xa, xb) and np.array_equal(ya, yb)


def _api_scenario(seed):
    """setup / solve / update q / update bounds / warm start / update matrices / polish on the multi-kernel path, one-launch solve +
    polish and two batch calls on the small path (tools/thread_stress_api.py runs more rounds of the same)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    out = []
    P, q, A, l, u = problems.banded_qp(3000, window=60, seed=seed)
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=20000)
    r = m.solve(); out.append((r.info.iter, r.x.copy(), r.y.copy()))
    m.update(q=q * (1 + 0.01 * rng.standard_normal(len(q)))); r = m.solve(); out.append((r.info.iter, r.x.copy(), r.y.copy()))
    m.update(l=l - 0.05, u=u + 0.05); r = m.solve(); out.append((r.info.iter, r.x.copy(), r.y.copy()))
    m.warm_start(x=r.x * 0.9, y=r.y * 0.9); r = m.solve(); out.append((r.info.iter, r.x.copy(), r.y.copy()))
    Pt = sp.triu(P, format='csc')
    m.update(Px=Pt.data * (1 + 0.02 * rng.random(Pt.nnz)), Ax=A.data * (1 + 0.02 * rng.standard_normal(A.nnz)))
    m.update_settings(polishing=True); r = m.solve(); out.append((r.info.iter, r.x.copy(), r.y.copy()))
    Pb, qb, Ab, L, U = problems.mpc_batch(16, seed=seed)
    s = osqp_amd.OSQP(); s.setup(Pb, qb, Ab, L[0], U[0], eps_abs=1e-6, eps_rel=1e-6, verbose=False, polishing=True)
    r = s.solve(); out.append((r.info.iter, r.x.copy(), r.y.copy()))
    for _ in range(2):
        x, y, rec = s._solver.hip_batch_solve(l=L, u=U); out.append((int(rec[:, 1].sum()), x.copy(), y.copy()))
    return out


def test_whole_api_sequence_is_deterministic_under_concurrency():
    seeds = list(range(70, 76))
    serial = [_api_scenario(s) for s in seeds]
    for _ in range(2):
        with ThreadPool(4) as pool:
            threaded = pool.map(_api_scenario, seeds)
        for a, b in zip(serial, threaded):
            for (ia, xa, ya), (ib, xb, yb) in zip(a, b):
                assert ia == ib and np.array_equal(xa, xb) and np.array_equal(ya, yb)
