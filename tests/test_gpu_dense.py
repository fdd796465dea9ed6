"""GPU tier: the dense fp64 kernels of the device-factorised Woodbury correction (dense_hip.hip: strided GEMM on v_mfma_f64_16x16x4, block Gauss-Jordan
inverse of an SPD matrix) through the paths that use them -- the row-space and the column-space form of the correction (OSQPHipPolicy::woodbury_dual)
with this engine's own kernels (the default) against the vendor route (OSQPHipPolicy::woodbury_vendor = 1: rocBLAS + rocSOLVER, the A/B switch)
and the oracle; sizes that are no multiple of the 64 x 64 tile or of the 64-column block step."""
import contextlib
import os
import warnings

import numpy as np
import pytest

import osqp_amd
import problems
from oracle import Oracle, SOLVED

pytestmark = pytest.mark.gpu
warnings.simplefilter('ignore')


@contextlib.contextmanager
def _env(**kv):
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _rel(a, b):
    return np.abs(a - b).max() / (1 + np.abs(b).max())


@pytest.mark.parametrize('nf,ns', [(150, 333), (301, 650)])
@pytest.mark.parametrize('dual', ['1', '0'])
def test_own_dense_kernels_equal_the_vendor_route_and_the_oracle(nf, ns, dual):
    P, q, A, l, u = problems.lasso_qp(nf, ns)
    xo, yo, io = Oracle().setup(P, q, A, l, u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000).solve()
    assert io.status_val == SOLVED
    res = {}
    for vendor in ('0', '1'):
        with _env(OSQP_HIP_WOODBURY_DUAL=dual, OSQP_HIP_WOODBURY_VENDOR=vendor):
            m = osqp_amd.OSQP()
            try:
                m.setup(P, q, A, l, u, eps_abs=1e-8, eps_rel=1e-8, verbose=False, max_iter=50000)
            except Exception:                                          # noqa: BLE001 -- (the vendor libraries may be absent on a box: then only the own route is checked)
                assert vendor == '1'
                continue
            r = m.solve(raise_error=True)
            res[vendor] = (r, m._solver.hip_stats(), m._solver.hip_preconditioner())
    r0, s0, name0 = res['0']
    assert s0['woodbury_rows'] == ns and s0['woodbury_direct'] == 1, (s0, name0)
    assert (s0['woodbury_dual_cols'] == nf) == (dual == '1')
    assert 'matrix cores' in name0 and 'vendor' not in name0
    assert _rel(r0.x, xo) < 5e-6 and _rel(r0.y, yo) < 2e-5
    if '1' in res and res['1'][1]['woodbury_rows'] == ns:
        r1 = res['1'][0]
        assert 'vendor' in res['1'][2]
        assert r1.info.iter == r0.info.iter and _rel(r0.x, r1.x) < 1e-7 and _rel(r0.y, r1.y) < 1e-6
