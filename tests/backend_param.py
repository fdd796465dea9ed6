"""Parametrisation of the API-level tests over the engine build they run against:
   'hip'     -- the product library on a real MI355X            (marked gpu; this is the parity tier)
   'hostsim' -- the same host driver linked to the plain-loop device-op simulator (CPU tier: exercises the host logic)
"""
import contextlib

import pytest

import os

# 'hip'      small problems take the one-launch direct path (banded LDL' in LDS, batch_hip.hip) -- the default
# 'hip-pcg'  OSQP_HIP_SMALL_DIRECT=0: the same problems through the multi-kernel PCG engine (pcg_hip.hip), so that the
#            reference's goldens pin BOTH kernels
BACKENDS = [pytest.param('hostsim'), pytest.param('hip', marks=pytest.mark.gpu), pytest.param('hip-pcg', marks=pytest.mark.gpu)]


@contextlib.contextmanager
def engine(backend):
    if backend == 'hostsim':
        from hostsim_util import hostsim
        with hostsim() as h:
            assert h.osqp_hip_backend() == b'hostsim'
            yield h
    else:
        from osqp_amd import _lib
        h = _lib.handle()
        assert h.osqp_hip_backend() == b'hip-gfx950', 'GPU tier must run the HIP library'
        old = os.environ.get('OSQP_HIP_SMALL_DIRECT')
        os.environ['OSQP_HIP_SMALL_DIRECT'] = '0' if backend == 'hip-pcg' else '1'
        try:
            yield h
        finally:
            if old is None:
                os.environ.pop('OSQP_HIP_SMALL_DIRECT', None)
            else:
                os.environ['OSQP_HIP_SMALL_DIRECT'] = old
