"""Drop-in proof at the C-API boundary (CPU tier, build container only): the reference's OWN pybind11 binding source
(/root/reference/src/bindings.cpp.in, used where it lies -- never copied into the repo) is configured exactly as the
reference's CMakeLists.txt:39-41 does (module name substituted for @OSQP_EXT_MODULE_NAME@), compiled against
include/compat/osqp_api_{functions,types}.h and linked to libosqp_hip.so.  The resulting extension module must expose the
surface the reference front-end consumes with this engine's enum values, defaults and capabilities.
Skipped when /root/reference is not present (e.g. on the GPU box); the build directory is git- and gpurun-ignored."""
import importlib.util
import os
import subprocess
import sys
import sysconfig

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = '/root/reference/src/bindings.cpp.in'
OUT = os.path.join(ROOT, 'tests', '_build', 'dropin')

pytestmark = pytest.mark.skipif(not os.path.exists(SRC), reason='reference tree not present')


@pytest.fixture(scope='module')
def ext():
    import pybind11
    import __graft_entry__ as g
    g.build()
    os.makedirs(OUT, exist_ok=True)
    cpp = os.path.join(OUT, 'bindings.cpp')
    with open(SRC) as f:
        text = f.read().replace('@OSQP_EXT_MODULE_NAME@', 'osqp_hip_ext')       # configure_file(), CMakeLists.txt:39-40
    with open(cpp, 'w') as f:
        f.write(text)
    so = os.path.join(OUT, 'osqp_hip_ext' + sysconfig.get_config_var('EXT_SUFFIX'))
    libdir = os.path.join(ROOT, 'osqp-python_amd', 'osqp_amd')
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-shared', '-fPIC', '-fvisibility=hidden', cpp, '-o', so,
                           '-I', os.path.join(ROOT, 'include', 'compat'), '-I', pybind11.get_include(), '-I', sysconfig.get_paths()['include'],
                           os.path.join(libdir, 'libosqp_hip.so'), '-Wl,-rpath,' + libdir])
    from osqp_amd import _lib
    _lib.handle()                                        # torch's HIP runtime first
    spec = importlib.util.spec_from_file_location('osqp_hip_ext', so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_reference_binding_compiles_and_exposes_the_ext_surface(ext):
    assert ext.OSQP_USE_FLOAT == 0 and ext.OSQP_USE_LONG == 0 and ext.OSQP_INFTY == 1e30                 # bindings.cpp.in:327-340
    assert int(ext.osqp_status_type.OSQP_SOLVED) == 1 and int(ext.osqp_status_type.OSQP_UNSOLVED) == 11   # :349-361
    assert int(ext.osqp_error_type.OSQP_FUNC_NOT_IMPLEMENTED) == 10                                       # :364-375
    assert ext.OSQP_INDIRECT_SOLVER == ext.osqp_linsys_solver_type.OSQP_INDIRECT_SOLVER                   # export_values, :346
    assert ext.osqp_capabilities() == int(ext.osqp_capabilities_type.OSQP_CAPABILITY_INDIRECT_SOLVER) | int(ext.osqp_capabilities_type.OSQP_CAPABILITY_UPDATE_MATRICES)
    s = ext.OSQPSettings(); ext.osqp_set_default_settings(s)                                             # :405-449
    assert (s.rho, s.sigma, s.alpha, s.scaling, s.max_iter, s.check_termination, s.cg_max_iter) == (0.1, 1e-6, 1.6, 10, 4000, 25, 50)
    assert s.linsys_solver == ext.osqp_linsys_solver_type.OSQP_INDIRECT_SOLVER and s.cg_precond == ext.OSQP_DIAGONAL_PRECONDITIONER
    assert s.polish_refine_iter == 3 and s.time_limit == 1e10
    names = [k for k in ext.OSQPSettings.__dict__ if not k.startswith('_')]                              # what interface.py:318-322 enumerates
    assert len(names) == 29
    A = ext.CSC(sp.csc_matrix(np.array([[1.0, 2.0], [0.0, 3.0]])))                                       # :12-62
    assert (A.m, A.n, A.nzmax, A.nz) == (2, 2, 3, -1) and list(A.p) == [0, 1, 3]
    d = ext.OSQPCodegenDefines(); ext.osqp_set_default_codegen_defines(d)
    assert d.embedded_mode == 1


def test_reference_binding_drives_the_engine(ext):
    """Through the reference's binding: setup -> (GPU present) solve, else the engine's loud refusal as ValueError('7')."""
    import torch
    P = ext.CSC(sp.csc_matrix(np.array([[4.0, 1.0], [0.0, 2.0]])))
    A = ext.CSC(sp.csc_matrix(np.array([[1.0, 1.0], [1.0, 0.0], [0.0, 1.0]])))
    q, l, u = np.array([1.0, 1.0]), np.array([1.0, 0.0, 0.0]), np.array([1.0, 0.7, 0.7])
    s = ext.OSQPSettings(); ext.osqp_set_default_settings(s); s.verbose = 0; s.eps_abs = s.eps_rel = 1e-7
    if not torch.cuda.is_available():
        with pytest.raises(ValueError, match='7'):                     # OSQP_ALGEBRA_LOAD_ERROR via py::value_error, :153-156
            ext.OSQPSolver(P, q, A, l, u, 3, 2, s)
        return
    solver = ext.OSQPSolver(P, q, A, l, u, 3, 2, s)
    solver.solve()
    assert solver.info.status_val == 1
    np.testing.assert_allclose(solver.solution.x, [0.3, 0.7], atol=1e-5)
