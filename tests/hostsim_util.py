"""Routes the package's ctypes handle to the host-simulator build of the engine (TEST INFRASTRUCTURE ONLY: lets the
CPU tier test the host driver + front-end; see tests/hostsim_build.py)."""
import ctypes
import contextlib

import hostsim_build


@contextlib.contextmanager
def hostsim():
    from osqp_amd import _lib
    old = _lib._handle
    _lib._handle = _lib._bind(ctypes.CDLL(hostsim_build.build()))
    try:
        yield _lib._handle
    finally:
        _lib._handle = old
