"""'Extension module' of the hip algebra: the object surface that /root/reference/src/bindings.cpp.in gives the
reference front-end (constants, enums, CSC, OSQPSettings, OSQPInfo, OSQPSolution, OSQPSolver ...), re-authored as a
thin ctypes shim over the C ABI of libosqp_hip.so (include/osqp_hip.h).  Checklist: SURVEY.md Appendix B.
"""
import ctypes as C
from enum import IntEnum

import numpy as np
import scipy.sparse as spa

from . import _lib

OSQP_USE_FLOAT = 0          # bindings.cpp.in:327-331
OSQP_USE_LONG = 0           # bindings.cpp.in:333-337
OSQP_INFTY = 1e30           # bindings.cpp.in:340


class osqp_linsys_solver_type(IntEnum):      # bindings.cpp.in:343-346
    OSQP_DIRECT_SOLVER = 1
    OSQP_INDIRECT_SOLVER = 2


class osqp_status_type(IntEnum):             # bindings.cpp.in:349-361
    OSQP_SOLVED = 1
    OSQP_SOLVED_INACCURATE = 2
    OSQP_PRIMAL_INFEASIBLE = 3
    OSQP_PRIMAL_INFEASIBLE_INACCURATE = 4
    OSQP_DUAL_INFEASIBLE = 5
    OSQP_DUAL_INFEASIBLE_INACCURATE = 6
    OSQP_MAX_ITER_REACHED = 7
    OSQP_TIME_LIMIT_REACHED = 8
    OSQP_NON_CVX = 9
    OSQP_SIGINT = 10
    OSQP_UNSOLVED = 11


class osqp_error_type(IntEnum):              # bindings.cpp.in:364-375
    OSQP_NO_ERROR = 0
    OSQP_DATA_VALIDATION_ERROR = 1
    OSQP_SETTINGS_VALIDATION_ERROR = 2
    OSQP_LINSYS_SOLVER_INIT_ERROR = 3
    OSQP_NONCVX_ERROR = 4
    OSQP_MEM_ALLOC_ERROR = 5
    OSQP_WORKSPACE_NOT_INIT_ERROR = 6
    OSQP_ALGEBRA_LOAD_ERROR = 7
    OSQP_CODEGEN_DEFINES_ERROR = 8
    OSQP_DATA_NOT_INITIALIZED = 9
    OSQP_FUNC_NOT_IMPLEMENTED = 10


class osqp_precond_type(IntEnum):            # bindings.cpp.in:378-381
    OSQP_NO_PRECONDITIONER = 0
    OSQP_DIAGONAL_PRECONDITIONER = 1


class osqp_capabilities_type(IntEnum):       # bindings.cpp.in:395-400
    OSQP_CAPABILITY_DIRECT_SOLVER = 0x01
    OSQP_CAPABILITY_INDIRECT_SOLVER = 0x02
    OSQP_CAPABILITY_CODEGEN = 0x04
    OSQP_CAPABILITY_UPDATE_MATRICES = 0x08
    OSQP_CAPABILITY_DERIVATIVES = 0x10


# .export_values() on the first, second and fourth enum (bindings.cpp.in:346,361,381)
for _e in (osqp_linsys_solver_type, osqp_status_type, osqp_precond_type):
    globals().update(_e.__members__)


def _ptr(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


class CSC:
    """bindings.cpp.in:12-62: zero-copy int32/f64 views of a SciPy CSC matrix."""

    def __init__(self, A):
        if not spa.isspmatrix_csc(A):
            A = spa.csc_matrix(A)
        self.m, self.n = int(A.shape[0]), int(A.shape[1])
        self.p = np.ascontiguousarray(A.indptr, dtype=np.int32)
        self.i = np.ascontiguousarray(A.indices, dtype=np.int32)
        self.x = np.ascontiguousarray(A.data, dtype=np.float64)
        self.nzmax = int(A.nnz)
        self.nz = -1

    def _struct(self):
        return _lib.CscStruct(self.m, self.n, _ptr(self.p, _lib.c_int_p), _ptr(self.i, _lib.c_int_p),
                              _ptr(self.x, _lib.c_double_p), self.nzmax, self.nz)


class OSQPSettings(_lib.SettingsStruct):
    """bindings.cpp.in:405-447.  The 29 fields are class attributes (ctypes descriptors), which is what the front-end
    enumerates to discover legal keyword settings (interface.py:318-322)."""
    pass


def osqp_set_default_settings(settings):       # bindings.cpp.in:449
    _lib.handle().osqp_set_default_settings(C.byref(settings))


def osqp_capabilities():                       # bindings.cpp.in:402
    return int(_lib.handle().osqp_capabilities())


def _info_property(name, typ, writable):
    def get(self):
        v = getattr(self._s.contents, name)
        return v.decode() if isinstance(v, bytes) else v

    def set_(self, v):
        setattr(self._s.contents, name, v)

    return property(get, set_ if writable else None)


class OSQPInfo:
    """bindings.cpp.in:473-492 (obj_val and dual_obj_val are read-write, :478-479)."""

    def __init__(self, struct_ptr):
        self._s = struct_ptr


for _n, _t in _lib.INFO_FIELDS:
    setattr(OSQPInfo, _n, _info_property(_n, _t, _n in ('obj_val', 'dual_obj_val')))


class OSQPSolution:
    """bindings.cpp.in:64-105: each access returns a fresh copy of the solver's host array."""

    def __init__(self, struct_ptr, m, n):
        self._s, self._m, self._n = struct_ptr, m, n

    def _arr(self, p, k):
        return np.ctypeslib.as_array(p, shape=(k,)).copy() if k > 0 else np.zeros(0)

    x = property(lambda self: self._arr(self._s.contents.x, self._n))
    y = property(lambda self: self._arr(self._s.contents.y, self._m))
    prim_inf_cert = property(lambda self: self._arr(self._s.contents.prim_inf_cert, self._m))
    dual_inf_cert = property(lambda self: self._arr(self._s.contents.dual_inf_cert, self._n))


def _vec(a, dtype=np.float64):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


class OSQPSolver:
    """bindings.cpp.in:107-323, :495-512."""

    def __init__(self, P, q, A, l, u, m, n, settings):
        self._lib = _lib.handle()
        self._p = _lib.SolverP()
        self.m, self.n = int(m), int(n)
        for a in (q, l, u):                              # .noconvert() at bindings.cpp.in:497
            if not (isinstance(a, np.ndarray) and a.dtype == np.float64):
                raise TypeError('q, l, u must be float64 numpy arrays')
        q, l, u = _vec(q), _vec(l), _vec(u)
        Ps, As = P._struct(), A._struct()
        status = self._lib.osqp_setup(C.byref(self._p), C.byref(Ps), _ptr(q, _lib.c_double_p), C.byref(As),
                                      _ptr(l, _lib.c_double_p), _ptr(u, _lib.c_double_p), self.m, self.n, C.byref(settings))
        if status:
            self._p = None
            raise ValueError(str(status))                # bindings.cpp.in:153-156
        self._solution = OSQPSolution(self._p.contents.solution, self.m, self.n)
        self._info = OSQPInfo(self._p.contents.info)

    def __del__(self):
        if getattr(self, '_p', None):
            self._lib.osqp_cleanup(self._p)              # bindings.cpp.in:159-161
            self._p = None

    solution = property(lambda self: self._solution)
    info = property(lambda self: self._info)

    def get_settings(self):
        s = OSQPSettings()
        C.memmove(C.byref(s), self._p.contents.settings, C.sizeof(s))
        return s

    def solve(self):                                     # ctypes drops the GIL for the call (cf. bindings.cpp.in:196-201)
        status = self._lib.osqp_solve(self._p)
        if status:                                       # a device / allocation failure inside the solve: never a silent stale result
            raise RuntimeError('osqp_solve failed with osqp_error_type %d' % status)
        return status

    def warm_start(self, x=None, y=None):
        x, y = _vec(x), _vec(y)
        return self._lib.osqp_warm_start(self._p, _ptr(x, _lib.c_double_p), _ptr(y, _lib.c_double_p))

    def update_data_vec(self, q=None, l=None, u=None):
        q, l, u = _vec(q), _vec(l), _vec(u)
        return self._lib.osqp_update_data_vec(self._p, _ptr(q, _lib.c_double_p), _ptr(l, _lib.c_double_p), _ptr(u, _lib.c_double_p))

    def hip_update_data_vec_device(self, q_ptr=None, l_ptr=None, u_ptr=None, stream=None):
        """osqp_hip_update_data_vec_device: raw device addresses (int or None = unchanged) of UNSCALED float64 vectors on this
        solver's GPU, e.g. torch_tensor.data_ptr(); ordered after the work queued on `stream` (hipStream_t handle as int)."""
        return self._lib.osqp_hip_update_data_vec_device(self._p, q_ptr, l_ptr, u_ptr, stream)

    def hip_warm_start_device(self, x_ptr=None, y_ptr=None, stream=None):
        return self._lib.osqp_hip_warm_start_device(self._p, x_ptr, y_ptr, stream)

    def update_data_mat(self, P_x=None, P_i=None, A_x=None, A_i=None):        # bindings.cpp.in:240-281
        P_x, A_x = _vec(P_x), _vec(A_x)
        P_i, A_i = _vec(P_i, np.int32), _vec(A_i, np.int32)
        P_n = len(P_i) if P_i is not None else (len(P_x) if P_x is not None else 0)
        A_n = len(A_i) if A_i is not None else (len(A_x) if A_x is not None else 0)
        return self._lib.osqp_update_data_mat(self._p, _ptr(P_x, _lib.c_double_p), _ptr(P_i, _lib.c_int_p), P_n,
                                              _ptr(A_x, _lib.c_double_p), _ptr(A_i, _lib.c_int_p), A_n)

    def update_settings(self, settings):
        status = self._lib.osqp_update_settings(self._p, C.byref(settings))
        if status:
            raise ValueError(str(status))                # bindings.cpp.in:204-209
        return status

    def update_rho(self, rho_new):
        return self._lib.osqp_update_rho(self._p, float(rho_new))

    # ---- out of scope (derivatives / codegen): the engine reports OSQP_FUNC_NOT_IMPLEMENTED ----
    def adjoint_derivative_compute(self, dx=None, dy=None):
        return self._lib.osqp_adjoint_derivative_compute(self._p, None, None)

    def codegen(self, output_dir, file_prefix, defines):
        return self._lib.osqp_codegen(self._p, None, None, None)

    # ---- engine extensions ----
    PRECONDITIONERS = {0: 'none', 1: 'jacobi', 2: 'jacobi + woodbury (dense rows, system inverted in one workgroup\'s LDS)',
                       3: 'jacobi + woodbury (dense rows, dense system formed and inverted on the fp64 matrix cores)'}      # OSQP_HIP_PRECOND_*

    def hip_stats(self):
        s = _lib.StatsStruct()
        self._lib.osqp_hip_get_stats(self._p, C.byref(s))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def hip_preconditioner(self):
        """What this handle's PCG is preconditioned with right now, in words (OSQPHipStats::preconditioner)."""
        st = self.hip_stats()
        name = self.PRECONDITIONERS[int(st['preconditioner'])]
        if int(st['preconditioner']) == 3:
            cd = int(st.get('woodbury_dual_cols', 0))
            name += (', column space (%d x %d)' % (cd, cd)) if cd else ', row space'
            if self.get_policy().get('woodbury_vendor'):
                name += ' -- vendor route (rocBLAS / rocSOLVER)'
        return name

    def get_policy(self):
        """This handle's engine policy (include/osqp_hip.h OSQPHipPolicy) as a dict."""
        p = _lib.PolicyStruct()
        self._lib.osqp_hip_get_policy(self._p, C.byref(p))
        return {k: getattr(p, k) for k, _ in p._fields_}

    def set_policy(self, **fields):
        """Change fields of this handle's policy, e.g. set_policy(small_direct=0, rho_window=0).  [setup] fields keep their value."""
        p = _lib.PolicyStruct()
        self._lib.osqp_hip_get_policy(self._p, C.byref(p))
        for k, v in fields.items():
            if k not in dict(p._fields_):
                raise ValueError('unknown policy field %r' % k)
            setattr(p, k, v)
        st = self._lib.osqp_hip_set_policy(self._p, C.byref(p))
        if st:
            raise ValueError(str(st))

    def hip_time_kernel(self, which, reps=50):
        ms = C.c_double()
        st = self._lib.osqp_hip_time_kernel(self._p, int(which), int(reps), C.byref(ms))
        if st:
            raise ValueError(str(st))
        return ms.value

    def hip_trace_read(self, count=1024 * 16):
        buf = (C.c_ulonglong * count)()
        st = self._lib.osqp_hip_trace_read(self._p, buf, count)
        if st:
            raise ValueError(str(st))
        return np.frombuffer(buf, dtype=np.uint64).copy()

    def hip_test_spmv(self, which, vec):
        vec = _vec(vec)
        out = np.empty(self.m if which == 0 else self.n)
        self._lib.osqp_hip_test_spmv(self._p, int(which), _ptr(vec, _lib.c_double_p), _ptr(out, _lib.c_double_p))
        return out

    def hip_set_rho_eq_factor(self, factor):
        return self._lib.osqp_hip_set_rho_eq_factor(self._p, float(factor))

    BATCH_FIELDS = ('status_val', 'iter', 'obj_val', 'prim_res', 'dual_res', 'rho', 'rho_updates', 'pcg_iters', 'status_polish', 'polish_time', 'rho_estimate', 'reserved')
    BATCH_REC = len(BATCH_FIELDS)        # OSQP_HIP_BATCH_REC

    def hip_batch_solve(self, q=None, l=None, u=None, x0=None, y0=None, nbatch=None, Px=None, Ax=None):
        """Solve a batch of QPs sharing this solver's P, A and settings (osqp_hip_batch_solve).  q: (B, n), l/u: (B, m).
        Px (B, nnz(triu P)) / Ax (B, nnz(A)): per-problem matrix values in the CSC order given at setup (osqp_hip_batch_solve_mat: the reference's
        per-element P_val / A_val, nn/torch.py:128-157) -- still one launch.  Returns x (B, n), y (B, m), rec (B, BATCH_REC) with columns BATCH_FIELDS."""
        arrs = [a for a in (q, l, u, x0, y0, Px, Ax) if a is not None]
        B = int(nbatch) if nbatch is not None else int(np.asarray(arrs[0]).shape[0])
        q, l, u = (None if a is None else np.ascontiguousarray(a, dtype=np.float64).reshape(B, -1) for a in (q, l, u))
        warm = x0 is not None or y0 is not None      # a missing one starts from zero, like warm_start(x=None) / (y=None) of a single solver
        # (l, u are clamped to +-OSQP_INFTY inside the kernel, like interface.py:334-337)
        x = np.zeros((B, self.n)) if x0 is None else np.ascontiguousarray(x0, dtype=np.float64).reshape(B, self.n).copy()
        y = np.zeros((B, self.m)) if y0 is None else np.ascontiguousarray(y0, dtype=np.float64).reshape(B, self.m).copy()
        rec = np.zeros((B, self.BATCH_REC))
        if Px is not None or Ax is not None:
            Px, Ax = (None if a is None else np.ascontiguousarray(a, dtype=np.float64).reshape(B, -1) for a in (Px, Ax))
            st = self._lib.osqp_hip_batch_solve_mat(self._p, B, _ptr(Px, _lib.c_double_p), _ptr(Ax, _lib.c_double_p), _ptr(q, _lib.c_double_p), _ptr(l, _lib.c_double_p),
                                                    _ptr(u, _lib.c_double_p), _ptr(x, _lib.c_double_p), _ptr(y, _lib.c_double_p), _ptr(rec, _lib.c_double_p), int(warm))
            if st:
                raise self._batch_error(st)
            return x, y, rec
        st = self._lib.osqp_hip_batch_solve(self._p, B, _ptr(q, _lib.c_double_p), _ptr(l, _lib.c_double_p), _ptr(u, _lib.c_double_p),
                                            _ptr(x, _lib.c_double_p), _ptr(y, _lib.c_double_p), _ptr(rec, _lib.c_double_p), int(warm))
        if st:
            raise self._batch_error(st)
        return x, y, rec

    def hip_batch_solve_device(self, nbatch, q_ptr, l_ptr, u_ptr, x_ptr, y_ptr, rec_ptr, warm=False, stream=None, Px_ptr=None, Ax_ptr=None):
        """osqp_hip_batch_solve_device: raw device addresses (int or None) of float64 arrays laid out as in hip_batch_solve;
        rec: (B, BATCH_REC).  Enqueued on `stream` (hipStream_t handle as int; None: the solver's stream, synchronous).
        Px_ptr / Ax_ptr: per-problem matrix values on the device (osqp_hip_batch_solve_mat_device)."""
        if Px_ptr is not None or Ax_ptr is not None:
            st = self._lib.osqp_hip_batch_solve_mat_device(self._p, int(nbatch), Px_ptr, Ax_ptr, q_ptr, l_ptr, u_ptr, x_ptr, y_ptr, rec_ptr, int(bool(warm)), stream)
        else:
            st = self._lib.osqp_hip_batch_solve_device(self._p, int(nbatch), q_ptr, l_ptr, u_ptr, x_ptr, y_ptr, rec_ptr, int(bool(warm)), stream)
        if st:
            raise self._batch_error(st)

    def _batch_error(self, st):
        """ValueError(str(code)) as the pybind layer raises for a failed call (callers compare str(e) with the code); `.reason` says why
        the batch kernel declined: OSQP_FUNC_NOT_IMPLEMENTED = the QP does not fit one workgroup's LDS, or this handle works on a
        reordered copy of the problem (OSQPHipPolicy::reorder = 2; the automatic mode never reorders a problem the batch kernel takes)."""
        e = ValueError(str(st))
        e.code = int(st)
        e.reason = ('the handle works on a reordered copy of the problem (OSQPHipPolicy::reorder): the batch kernel takes the caller\'s numbering only'
                    if self.hip_stats().get('reordered') else 'the batch kernel declined (code %d): the QP does not fit one workgroup\'s LDS, or an argument is invalid' % int(st))
        return e

    def hip_scaling(self):
        D, E, c = np.empty(self.n), np.empty(self.m), C.c_double()
        self._lib.osqp_hip_get_scaling(self._p, _ptr(D, _lib.c_double_p), _ptr(E, _lib.c_double_p), C.byref(c))
        return D, E, c.value
