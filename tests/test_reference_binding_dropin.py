"""Drop-in proof at the C-API boundary (CPU tier, build container only): the reference's OWN pybind11 binding source
(/root/reference/src/bindings.cpp.in, used where it lies -- never copied into the repo) is configured exactly as the
reference's CMakeLists.txt:39-41 does (module name substituted for @OSQP_EXT_MODULE_NAME@), compiled against
include/compat/osqp_api_{functions,types}.h and linked to libosqp_hip.so.  The resulting extension module must expose the
surface the reference front-end consumes with this engine's enum values, defaults and capabilities.
Skipped when /root/reference is not present (e.g. on the GPU box).  The configured source and the built module live in a pytest temporary
directory (tmp_path_factory): nothing derived from the reference's source is ever written under the repository."""
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

pytestmark = pytest.mark.skipif(not os.path.exists(SRC), reason='reference tree not present')


def _build_binding(modname, lib, OUT):
    """configure_file() + compile of the reference's binding source against include/compat, linked to `lib`; everything goes to OUT (a temp dir)"""
    import pybind11
    OUT = str(OUT)
    cpp = os.path.join(OUT, modname + '.cpp')
    with open(SRC) as f:
        text = f.read().replace('@OSQP_EXT_MODULE_NAME@', modname)       # configure_file(), CMakeLists.txt:39-40
    with open(cpp, 'w') as f:
        f.write(text)
    so = os.path.join(OUT, modname + sysconfig.get_config_var('EXT_SUFFIX'))
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-shared', '-fPIC', '-fvisibility=hidden', cpp, '-o', so,
                           '-I', os.path.join(ROOT, 'include', 'compat'), '-I', pybind11.get_include(), '-I', sysconfig.get_paths()['include'],
                           lib, '-Wl,-rpath,' + os.path.dirname(lib)])
    spec = importlib.util.spec_from_file_location(modname, so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope='module')
def ext(tmp_path_factory):
    import __graft_entry__ as g
    g.build()
    from osqp_amd import _lib
    _lib.handle()                                        # torch's HIP runtime first
    return _build_binding('osqp_hip_ext', os.path.join(ROOT, 'osqp-python_amd', 'osqp_amd', 'libosqp_hip.so'), tmp_path_factory.mktemp('dropin_hip'))


@pytest.fixture(scope='module')
def ext_hostsim(tmp_path_factory):
    """The same binding source linked against the HOST-SIMULATOR build of the engine (tests/hostsim_build.py: the product's host driver
    + api.cpp over plain-loop device ops): the whole call sequence of bindings.cpp.in:153-281 can then be executed in a container
    without a GPU.  Test infrastructure; nothing of it travels or ships."""
    import hostsim_build
    return _build_binding('osqp_hostsim_ext', hostsim_build.build(), tmp_path_factory.mktemp('dropin_hostsim'))


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


def test_reference_binding_drives_a_whole_call_sequence_on_the_host_simulator(ext_hostsim):
    """setup -> solve -> update_data_vec -> solve -> update_data_mat (by index) -> solve -> warm_start -> solve -> update_settings ->
    update_rho -> solve, all through the reference's own binding (bindings.cpp.in:153 osqp_setup, :198 osqp_solve, :237 osqp_update_data_vec,
    :280 osqp_update_data_mat, :193 osqp_warm_start, :204 osqp_update_settings, :213 osqp_update_rho), against the SAME sequence through
    this repo's ctypes front-end on the same library: x, y, iteration counts and objective must agree bit for bit at every stage."""
    import osqp_amd
    from hostsim_util import hostsim
    from util import Fixture
    ext = ext_hostsim
    f = Fixture('matrices_update_P_A')                      # update_matrices_test.py: n = 5, m = 8, new P / A values in the fixture
    Pu = sp.triu(f.P, format='csc'); Pu.sort_indices()
    A = f.A.copy(); A.sort_indices()
    n, m = f.n, f.m
    rng = np.random.default_rng(0)
    q2 = f.q + 0.1 * rng.standard_normal(n)
    l2, u2 = f.l - 0.05, f.u + 0.05
    Px_idx = np.arange(0, Pu.nnz, 2, dtype=np.int32); Px_new = (Pu.data[Px_idx] * 1.1).copy()
    Ax_idx = np.arange(1, A.nnz, 3, dtype=np.int32); Ax_new = (A.data[Ax_idx] * 0.9).copy()
    stg = dict(eps_abs=1e-7, eps_rel=1e-7, max_iter=4000, check_termination=1, adaptive_rho_interval=25, scaling=10)

    # ---- the reference binding
    s = ext.OSQPSettings(); ext.osqp_set_default_settings(s); s.verbose = 0
    for k, v in stg.items():
        setattr(s, k, v)
    Pc, Ac = ext.CSC(Pu), ext.CSC(A)
    solver = ext.OSQPSolver(Pc, np.ascontiguousarray(f.q), Ac, np.ascontiguousarray(f.l), np.ascontiguousarray(f.u), m, n, s)
    stages = []

    def snap(tag):
        assert solver.solve() == 0
        info, sol = solver.info, solver.solution
        stages.append((tag, np.array(sol.x), np.array(sol.y), int(info.iter), float(info.obj_val), int(info.status_val)))
    snap('setup')
    assert solver.update_data_vec(q2, l2, u2) == 0; snap('vec')
    assert solver.update_data_mat(Px_new, Px_idx, Ax_new, Ax_idx) == 0; snap('mat')
    assert solver.warm_start(np.zeros(n), None) == 0; snap('warm')
    s2 = solver.get_settings(); s2.alpha = 1.4; s2.max_iter = 3000
    assert solver.update_settings(s2) == 0
    assert solver.update_rho(0.3) == 0; snap('settings')
    with pytest.raises(ValueError, match='2'):              # OSQP_SETTINGS_VALIDATION_ERROR through py::value_error (:204-209)
        bad = ext.OSQPSettings(); ext.osqp_set_default_settings(bad); bad.alpha = 3.0
        solver.update_settings(bad)

    # ---- the same sequence through this repo's front-end (ctypes) on the same library
    with hostsim():
        mdl = osqp_amd.OSQP(algebra='hip')
        mdl.setup(Pu, f.q, A, f.l, f.u, verbose=False, **stg)
        mine = []

        def snap2(tag):
            r = mdl.solve()
            mine.append((tag, r.x.copy(), r.y.copy(), int(r.info.iter), float(r.info.obj_val), int(r.info.status_val)))
        snap2('setup')
        mdl.update(q=q2, l=l2, u=u2); snap2('vec')
        mdl.update(Px=Px_new, Px_idx=Px_idx, Ax=Ax_new, Ax_idx=Ax_idx); snap2('mat')
        mdl.warm_start(x=np.zeros(n)); snap2('warm')
        mdl.update_settings(alpha=1.4, max_iter=3000, rho=0.3); snap2('settings')
    for a, b in zip(stages, mine):
        assert a[0] == b[0] and a[5] == 1 and b[5] == 1, (a[0], a[5], b[5])
        assert a[3] == b[3], (a[0], a[3], b[3])
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[4] == b[4], a[0]
    assert len({st[3] for st in stages}) > 1                # (the stages really are different solves)
