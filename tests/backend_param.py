"""Parametrisation of the API-level tests over the engine build they run against:
   'hip'     -- the product library on a real MI355X            (marked gpu; this is the parity tier)
   'hostsim' -- the same host driver linked to the plain-loop device-op simulator (CPU tier: exercises the host logic)
"""
import contextlib

import pytest

BACKENDS = [pytest.param('hostsim'), pytest.param('hip', marks=pytest.mark.gpu)]


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
        yield h
