"""Loads libosqp_hip.so (the C-ABI engine built from ../csrc by ../Makefile) and declares its prototypes.

There is deliberately NO fallback: if the HIP library is missing the import fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OSQP_HIP_LIBRARY selects another build of the same engine (e.g. the diagnostic libosqp_hip_trace.so of tools/ktrace.py)
LIB_PATH = os.environ.get('OSQP_HIP_LIBRARY') or os.path.join(_HERE, 'libosqp_hip.so')

c_int_p = C.POINTER(C.c_int)
c_double_p = C.POINTER(C.c_double)


class CscStruct(C.Structure):          # OSQPCscMatrix, include/osqp_hip.h
    _fields_ = [('m', C.c_int), ('n', C.c_int), ('p', c_int_p), ('i', c_int_p), ('x', c_double_p),
                ('nzmax', C.c_int), ('nz', C.c_int)]


SETTINGS_FIELDS = [            # OSQPSettings, same order as include/osqp_hip.h (= bindings.cpp.in:409-447)
    ('device', C.c_int), ('linsys_solver', C.c_int), ('verbose', C.c_int), ('warm_starting', C.c_int),
    ('scaling', C.c_int), ('polishing', C.c_int), ('rho', C.c_double), ('rho_is_vec', C.c_int),
    ('sigma', C.c_double), ('alpha', C.c_double), ('cg_max_iter', C.c_int), ('cg_tol_reduction', C.c_int),
    ('cg_tol_fraction', C.c_double), ('cg_precond', C.c_int), ('adaptive_rho', C.c_int),
    ('adaptive_rho_interval', C.c_int), ('adaptive_rho_fraction', C.c_double), ('adaptive_rho_tolerance', C.c_double),
    ('max_iter', C.c_int), ('eps_abs', C.c_double), ('eps_rel', C.c_double), ('eps_prim_inf', C.c_double),
    ('eps_dual_inf', C.c_double), ('scaled_termination', C.c_int), ('check_termination', C.c_int),
    ('check_dualgap', C.c_int), ('time_limit', C.c_double), ('delta', C.c_double), ('polish_refine_iter', C.c_int)]


class SettingsStruct(C.Structure):
    _fields_ = SETTINGS_FIELDS


INFO_FIELDS = [               # OSQPInfo (bindings.cpp.in:473-492)
    ('status', C.c_char * 32), ('status_val', C.c_int), ('status_polish', C.c_int), ('obj_val', C.c_double),
    ('dual_obj_val', C.c_double), ('prim_res', C.c_double), ('dual_res', C.c_double), ('duality_gap', C.c_double),
    ('iter', C.c_int), ('rho_updates', C.c_int), ('rho_estimate', C.c_double), ('setup_time', C.c_double),
    ('solve_time', C.c_double), ('update_time', C.c_double), ('polish_time', C.c_double), ('run_time', C.c_double),
    ('primdual_int', C.c_double), ('rel_kkt_error', C.c_double)]


class InfoStruct(C.Structure):
    _fields_ = INFO_FIELDS


class SolutionStruct(C.Structure):
    _fields_ = [('x', c_double_p), ('y', c_double_p), ('prim_inf_cert', c_double_p), ('dual_inf_cert', c_double_p)]


class SolverStruct(C.Structure):
    _fields_ = [('settings', C.POINTER(SettingsStruct)), ('solution', C.POINTER(SolutionStruct)),
                ('info', C.POINTER(InfoStruct)), ('work', C.c_void_p)]


class StatsStruct(C.Structure):
    _fields_ = [(k, C.c_double) for k in ('pcg_iters_total', 'pcg_iters_max', 'pcg_unconverged', 'kernel_launches',
                                          'graph_launches', 'gpu_solve_ms', 'nnzA', 'nnzB', 'pcg_fused', 'batch_direct_bw',
                                          'cg_cap_escalations', 'windowed_blocks', 'row_blocks', 'slot_topups', 'f1_replicas', 'woodbury_rows', 'woodbury_direct', 'preconditioner', 'woodbury_factorisations', 'woodbury_factor_ms', 'reordered', 'reorder_ms', 'woodbury_cache_hits', 'f1_far_columns', 'woodbury_dual_cols', 'woodbury_fused_iteration', 'woodbury_one_launch', 'kform_nnz', 'batch_wave_split')]


class PolicyStruct(C.Structure):       # OSQPHipPolicy, include/osqp_hip.h (same order)
    _fields_ = ([(k, C.c_int) for k in ('graph', 'slots', 'pcg_fused', 'f1', 'window', 'woodbury', 'woodbury_direct', 'woodbury_large', 'device_driven', 'small_direct', 'batch_reorder', 'batch_variant')] +
                [('extrap', C.c_double), ('rho_eq_factor', C.c_double), ('rho_window', C.c_int), ('rho_window_tol', C.c_double), ('rho_persist', C.c_int),
                 ('rho_tol_exp', C.c_double), ('budget_tolerate', C.c_double), ('budget_sigma', C.c_double), ('budget_slack', C.c_int), ('budget_full', C.c_int),
                 ('cg_escalate', C.c_int), ('stall', C.c_int), ('polish_delta_floor', C.c_double), ('polish_pcg_tol', C.c_double), ('slot_poll', C.c_int), ('poll_low', C.c_int), ('poll_first', C.c_double),
                 ('poll_frac', C.c_double), ('poll_wait', C.c_double), ('finish_pairs', C.c_int), ('poll_sleep_us', C.c_int),
                 ('slot_log', C.c_int), ('setup_timing', C.c_int), ('batch_timing', C.c_int), ('woodbury_log', C.c_int), ('woodbury_direct_tol', C.c_double), ('woodbury_fused', C.c_int), ('debug_fail_refactor', C.c_int), ('reorder', C.c_int), ('woodbury_cache', C.c_int), ('kform', C.c_int), ('woodbury_dual', C.c_int), ('woodbury_vendor', C.c_int), ('batch_wave', C.c_int)])


SolverP = C.POINTER(SolverStruct)


class LinSysStruct(C.Structure):      # OSQPHipLinSysSolver, include/osqp_hip.h (the LinSysSolver slot)
    pass


LinSysP = C.POINTER(LinSysStruct)
LinSysStruct._fields_ = [
    ('type', C.c_int),
    ('name', C.CFUNCTYPE(C.c_char_p, LinSysP)),
    ('solve', C.CFUNCTYPE(C.c_int, LinSysP, c_double_p, C.c_int)),
    ('update_settings', C.CFUNCTYPE(None, LinSysP, C.POINTER(SettingsStruct))),
    ('warm_start', C.CFUNCTYPE(None, LinSysP, c_double_p)),
    ('adjoint_derivative', C.CFUNCTYPE(C.c_int, LinSysP)),
    ('free', C.CFUNCTYPE(None, LinSysP)),
    ('update_matrices', C.CFUNCTYPE(C.c_int, LinSysP, C.POINTER(CscStruct), c_int_p, C.c_int, C.POINTER(CscStruct), c_int_p, C.c_int)),
    ('update_rho_vec', C.CFUNCTYPE(C.c_int, LinSysP, c_double_p, C.c_double)),
    ('nthreads', C.c_int), ('pcg_iters', C.c_int), ('impl', C.c_void_p)]

# every symbol include/osqp_hip.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    'osqp_capabilities': (C.c_int, []),
    'osqp_set_default_settings': (None, [C.POINTER(SettingsStruct)]),
    'osqp_version': (C.c_char_p, []),
    'osqp_setup': (C.c_int, [C.POINTER(SolverP), C.POINTER(CscStruct), c_double_p, C.POINTER(CscStruct), c_double_p,
                             c_double_p, C.c_int, C.c_int, C.POINTER(SettingsStruct)]),
    'osqp_solve': (C.c_int, [SolverP]),
    'osqp_cleanup': (C.c_int, [SolverP]),
    'osqp_warm_start': (C.c_int, [SolverP, c_double_p, c_double_p]),
    'osqp_cold_start': (C.c_int, [SolverP]),
    'osqp_update_data_vec': (C.c_int, [SolverP, c_double_p, c_double_p, c_double_p]),
    'osqp_update_data_mat': (C.c_int, [SolverP, c_double_p, c_int_p, C.c_int, c_double_p, c_int_p, C.c_int]),
    'osqp_update_settings': (C.c_int, [SolverP, C.POINTER(SettingsStruct)]),
    'osqp_update_rho': (C.c_int, [SolverP, C.c_double]),
    'osqp_get_dimensions': (None, [SolverP, c_int_p, c_int_p]),
    'osqp_adjoint_derivative_compute': (C.c_int, [SolverP, c_double_p, c_double_p]),
    'osqp_adjoint_derivative_get_mat': (C.c_int, [SolverP, C.POINTER(CscStruct), C.POINTER(CscStruct)]),
    'osqp_adjoint_derivative_get_vec': (C.c_int, [SolverP, c_double_p, c_double_p, c_double_p]),
    'osqp_codegen': (C.c_int, [SolverP, C.c_char_p, C.c_char_p, C.c_void_p]),
    'osqp_set_default_codegen_defines': (None, [C.c_void_p]),
    'osqp_hip_set_default_print': (None, [C.c_void_p, C.c_void_p]),
    'osqp_hip_set_print': (C.c_int, [SolverP, C.c_void_p, C.c_void_p]),
    'osqp_hip_get_stats': (C.c_int, [SolverP, C.POINTER(StatsStruct)]),
    'osqp_hip_time_kernel': (C.c_int, [SolverP, C.c_int, C.c_int, c_double_p]),
    'osqp_hip_trace_read': (C.c_int, [SolverP, C.POINTER(C.c_ulonglong), C.c_int]),
    'osqp_hip_test_spmv': (C.c_int, [SolverP, C.c_int, c_double_p, c_double_p]),
    'osqp_hip_set_rho_eq_factor': (C.c_int, [SolverP, C.c_double]),
    'osqp_hip_default_policy': (None, [C.POINTER(PolicyStruct)]),
    'osqp_hip_set_default_policy': (None, [C.POINTER(PolicyStruct)]),
    'osqp_hip_set_policy': (C.c_int, [SolverP, C.POINTER(PolicyStruct)]),
    'osqp_hip_get_policy': (C.c_int, [SolverP, C.POINTER(PolicyStruct)]),
    'osqp_hip_batch_solve': (C.c_int, [SolverP, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int]),
    'osqp_hip_batch_solve_device': (C.c_int, [SolverP, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'osqp_hip_batch_solve_mat': (C.c_int, [SolverP, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int]),
    'osqp_hip_batch_solve_mat_device': (C.c_int, [SolverP, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'osqp_hip_update_data_vec_device': (C.c_int, [SolverP, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'osqp_hip_warm_start_device': (C.c_int, [SolverP, C.c_void_p, C.c_void_p, C.c_void_p]),
    'osqp_hip_get_scaling': (C.c_int, [SolverP, c_double_p, c_double_p, c_double_p]),
    'osqp_hip_get_reordering': (C.c_int, [SolverP, c_int_p, c_int_p]),
    'osqp_hip_backend': (C.c_char_p, []),
    'osqp_hip_linsys_init': (C.c_int, [C.POINTER(LinSysP), C.POINTER(CscStruct), C.POINTER(CscStruct), c_double_p,
                                       C.POINTER(SettingsStruct), c_double_p, c_double_p, C.c_int]),
}


PRINT_FN = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)      # osqp_hip_print_fn


def _py_print(text, _user):
    """Where `verbose` output of every solver handle goes: sys.stdout, like the reference's c_print = PySys_WriteStdout under the GIL
    (/root/reference/cmake/printing.h:2-7).  Called from inside osqp_setup / osqp_solve, which ctypes runs with the GIL released -- a ctypes
    callback takes it for the duration of the call."""
    import sys
    try:
        sys.stdout.write(text.decode('utf-8', 'replace'))
    except Exception:                      # noqa: BLE001 -- a closed / replaced stdout must not take the solve down
        pass


_print_cb = PRINT_FN(_py_print)            # (module-level: must outlive every handle of the library)


def _bind(lib):
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    lib.osqp_hip_set_default_print(C.cast(_print_cb, C.c_void_p), None)
    return lib


_handle = None


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7) and
    libhsa-runtime64.so; if /opt/rocm's copy is already mapped when torch is imported, torch maps its second copy and the
    two HSA runtimes fight over /dev/kfd (hipGetDeviceCount fails in whichever initialises second).  Loading torch's
    runtime FIRST makes the dynamic linker satisfy libosqp_hip.so's NEEDED libamdhip64.so.7 by SONAME with the copy that
    is already mapped, so the engine, torch.cuda and torch.distributed (RCCL) share one runtime and one set of streams.
    Without torch installed the engine binds to /opt/rocm/lib (its RUNPATH)."""
    import importlib.util
    import sys
    if 'torch' in sys.modules or importlib.util.find_spec('torch') is None:
        return
    import torch  # noqa: F401


def handle():
    """The loaded engine library.  Raises ImportError when it has not been built."""
    global _handle
    if _handle is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                'libosqp_hip.so not found at %s -- build the HIP engine first (python -c "import __graft_entry__ as g; '
                'g.build()" or make -C osqp-python_amd).  There is no CPU fallback.' % LIB_PATH)
        _preload_torch_hip_runtime()
        _handle = _bind(C.CDLL(LIB_PATH))
    return _handle
