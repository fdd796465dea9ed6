"""CPU tier: the C-ABI library loads and exports every symbol include/osqp_hip.h declares (no compute call)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    h = open(os.path.join(ROOT, 'include', 'osqp_hip.h')).read()
    h = re.sub(r'/\*.*?\*/', '', h, flags=re.S)
    return sorted(set(re.findall(r'\b(osqp_[a-z_]+)\s*\(', h)))


def test_header_and_ctypes_prototypes_agree():
    from osqp_amd import _lib
    assert set(declared_functions()) == set(_lib.PROTOTYPES)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from osqp_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name
    lib.osqp_hip_backend.restype = ctypes.c_char_p
    assert lib.osqp_hip_backend() == b'hip-gfx950'
    lib.osqp_capabilities.restype = ctypes.c_int
    assert lib.osqp_capabilities() == 0x02 | 0x08


def test_settings_defaults_and_struct_layout():
    from osqp_amd import _lib, ext_hip
    s = ext_hip.OSQPSettings()
    _lib._bind(ctypes.CDLL(_lib.LIB_PATH)).osqp_set_default_settings(ctypes.byref(s))
    assert (s.rho, s.sigma, s.alpha, s.scaling, s.max_iter, s.check_termination) == (0.1, 1e-6, 1.6, 10, 4000, 25)
    assert s.linsys_solver == ext_hip.OSQP_INDIRECT_SOLVER and s.cg_precond == ext_hip.OSQP_DIAGONAL_PRECONDITIONER
    assert s.polish_refine_iter == 3 and s.delta == 1e-6 and s.time_limit == 1e10     # last fields: layout check
    assert len(ext_hip.OSQPSettings._fields_) == 29                                   # bindings.cpp.in:409-447


def test_setup_without_gpu_fails_loudly():
    """On a box without a GPU the product must refuse (no CPU fallback): OSQP_ALGEBRA_LOAD_ERROR."""
    import numpy as np
    import scipy.sparse as sp
    import torch
    if torch.cuda.is_available():
        return
    import osqp_amd
    m = osqp_amd.OSQP()
    try:
        m.setup(sp.eye(2, format='csc'), np.zeros(2), sp.eye(2, format='csc'), -np.ones(2), np.ones(2), verbose=False)
    except osqp_amd.OSQPException as e:
        assert e == osqp_amd.SolverError.OSQP_ALGEBRA_LOAD_ERROR
    else:
        raise AssertionError('setup succeeded without a GPU')
