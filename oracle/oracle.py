"""ctypes wrapper around oracle/_build/liboracle.so  (TEST INFRASTRUCTURE ONLY).

The oracle is the CPU restatement of the reference algorithm
(/root/reference/src/osqppurepy/_osqp.py, see osqp_oracle.c for per-function citations).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, '_build', 'liboracle.so')

# status values (v1.0.0 enum order, bindings.cpp.in:349-361)
SOLVED, SOLVED_INACCURATE, PRIMAL_INFEASIBLE, PRIMAL_INFEASIBLE_INACCURATE, DUAL_INFEASIBLE, \
    DUAL_INFEASIBLE_INACCURATE, MAX_ITER_REACHED, TIME_LIMIT_REACHED, NON_CVX, SIGINT, UNSOLVED = range(1, 12)
INFTY = 1e30


class Settings(C.Structure):
    _fields_ = [(k, C.c_double) for k in ('rho', 'sigma', 'alpha', 'eps_abs', 'eps_rel', 'eps_prim_inf',
                                           'eps_dual_inf', 'adaptive_rho_tolerance', 'pcg_tol')] + \
               [(k, C.c_int) for k in ('scaling', 'max_iter', 'scaled_termination', 'check_termination',
                                       'warm_start', 'adaptive_rho', 'adaptive_rho_interval', 'linsys',
                                       'ordering', 'pcg_max_iter', 'c_core_warm_start')]


class Info(C.Structure):
    _fields_ = [('iter', C.c_int), ('status_val', C.c_int), ('rho_updates', C.c_int)] + \
               [(k, C.c_double) for k in ('obj_val', 'pri_res', 'dua_res', 'rho_estimate', 'setup_time',
                                          'solve_time', 'lnz', 'pcg_iters')]


def build(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, 'osqp_oracle.c')):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B'])
    return _LIB


def build_native():
    """The same source compiled `-O3 -march=native` FOR THE HOST IT RUNS ON (the CPU baseline's flags) -- used only by
    the cpu_baseline legs of bench.py / bench_batch.py, which call use_native() in their timing processes.  The file name carries a hash
    of this host's CPU flags: a build that travelled from another machine is never loaded (its instructions may not exist here)."""
    import hashlib
    try:
        with open('/proc/cpuinfo') as f:
            flags = next((ln for ln in f if ln.startswith('flags')), '')
    except OSError:
        flags = ''
    out = os.path.join(_HERE, '_build', 'liboracle_native_%s.so' % hashlib.sha1(flags.encode()).hexdigest()[:10])
    src = os.path.join(_HERE, 'osqp_oracle.c')
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call([os.environ.get('CC', 'gcc'), '-O3', '-march=native', '-fno-fast-math', '-shared', '-fPIC', '-o', out, src, '-lm'])
    return out


_lib = None
_native = False


def use_native(on=True):
    """Route this PROCESS's oracle calls to the -march=native build (call before the first Oracle())."""
    global _native, _lib
    if bool(on) != _native:
        _native, _lib = bool(on), None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_native() if _native else build())
        _lib.oracle_setup.restype = C.c_void_p
        _lib.oracle_solve.argtypes = [C.c_void_p] + [C.c_void_p] * 3 + [C.POINTER(Info)]
        for f in ('oracle_update_lin_cost', 'oracle_update_bounds', 'oracle_update_matrices', 'oracle_warm_start',
                  'oracle_free', 'oracle_set_trace', 'oracle_update_settings', 'oracle_get_scaling'):
            getattr(_lib, f).argtypes = None
        _lib.oracle_update_rho.argtypes = [C.c_void_p, C.c_double]
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    """Mirrors the setup/solve/update/warm_start surface of osqppurepy.OSQP (interface.py:18-292)."""

    def __init__(self):
        self._h = None

    def setup(self, P, q, A, l, u, **stg):
        L = lib()
        n = len(q)
        m = A.shape[0] if A is not None else 0
        if P is None:
            P = sp.csc_matrix((n, n))
        if A is None:
            A = sp.csc_matrix((0, n)); l = np.zeros(0); u = np.zeros(0)
        P = sp.triu(sp.csc_matrix(P), format='csc'); P.sort_indices()
        A = sp.csc_matrix(A); A.sort_indices()
        self.n, self.m = n, m
        self._keep = [np.ascontiguousarray(P.indptr, np.int32), np.ascontiguousarray(P.indices, np.int32), _f64(P.data),
                      _f64(q), np.ascontiguousarray(A.indptr, np.int32), np.ascontiguousarray(A.indices, np.int32),
                      _f64(A.data), _f64(np.maximum(l, -INFTY)), _f64(np.minimum(u, INFTY))]
        s = Settings(); L.oracle_default_settings(C.byref(s))
        for k, v in stg.items():
            if not hasattr(s, k):
                raise ValueError('unknown oracle setting ' + k)
            setattr(s, k, type(getattr(s, k))(v))
        self.settings = s
        err = C.c_int(0)
        L.oracle_setup.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 9 + [C.POINTER(Settings), C.POINTER(C.c_int)]
        self._h = L.oracle_setup(n, m, *[_dp(a) for a in self._keep], C.byref(s), C.byref(err))
        self.setup_error = err.value
        if err.value:
            raise ValueError('oracle setup error %d' % err.value)
        return self

    def set_trace(self, cap):
        self._tp = np.full(cap, np.nan); self._td = np.full(cap, np.nan)
        lib().oracle_set_trace(C.c_void_p(self._h), _dp(self._tp), _dp(self._td), C.c_int(cap))
        return self._tp, self._td

    def solve(self):
        x = np.empty(self.n); y = np.empty(self.m); cert = np.zeros(max(self.n, self.m, 1)); info = Info()
        lib().oracle_solve(C.c_void_p(self._h), _dp(x), _dp(y), _dp(cert), C.byref(info))
        self.cert = cert
        return x, y, info

    def polish(self, delta=1e-6, polish_refine_iter=3, sparse=None):
        """The reference's polish step on the iterates the last solve() left (status must be SOLVED): returns x, y, info, status_polish.
        sparse: factorise the reduced KKT matrix with the sparse LDL' of the direct path (oracle_polish_sparse) instead of the dense LU
        (default: from n + m > 4000 on -- the dense LU is O(N^3)); both restate the same algorithm, tests/test_oracle_golden.py checks
        one against the other."""
        x = np.empty(self.n); y = np.empty(self.m); info = Info()
        L = lib()
        if sparse is None:
            sparse = self.n + self.m > 4000 and self.settings.linsys == 0
        fn = L.oracle_polish_sparse if sparse else L.oracle_polish
        fn.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(Info)]
        st = fn(C.c_void_p(self._h), float(delta), int(polish_refine_iter), _dp(x), _dp(y), C.byref(info))
        return x, y, info, st

    def update(self, q=None, l=None, u=None, Px=None, Ax=None):
        L = lib(); h = C.c_void_p(self._h)
        if q is not None:
            self._keep[3] = _f64(q); L.oracle_update_lin_cost(h, _dp(self._keep[3]))
        if l is not None or u is not None:
            if l is not None:
                self._keep[7] = _f64(np.maximum(l, -INFTY))
            if u is not None:
                self._keep[8] = _f64(np.minimum(u, INFTY))
            if L.oracle_update_bounds(h, _dp(self._keep[7]), _dp(self._keep[8])):
                raise ValueError('lower bound must not exceed upper bound')
        if Px is not None or Ax is not None:
            L.oracle_update_matrices(h, _dp(_f64(Px)), _dp(_f64(Ax)))

    def warm_start(self, x=None, y=None):
        lib().oracle_warm_start(C.c_void_p(self._h), _dp(_f64(x)), _dp(_f64(y)))

    def update_rho(self, rho):
        lib().oracle_update_rho(C.c_void_p(self._h), float(rho))

    def update_settings(self, **kw):
        for k, v in kw.items():
            setattr(self.settings, k, type(getattr(self.settings, k))(v))
        lib().oracle_update_settings(C.c_void_p(self._h), C.byref(self.settings))

    def scaling(self):
        D = np.empty(self.n); E = np.empty(self.m); c = C.c_double()
        lib().oracle_get_scaling(C.c_void_p(self._h), _dp(D), _dp(E), C.byref(c))
        return D, E, c.value

    def __del__(self):
        if self._h is not None and _lib is not None:
            _lib.oracle_free(C.c_void_p(self._h)); self._h = None
