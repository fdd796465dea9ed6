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
