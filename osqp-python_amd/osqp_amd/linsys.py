"""The LinSysSolver slot of the engine (include/osqp_hip.h: OSQPHipLinSysSolver) from Python: the reduced-KKT PCG as a
stand-alone linear solver with the meaning of the reference's ``linsys_solver.solve`` (osqppurepy/_osqp.py:286-311):

    b = [rhs_x; rhs_z]  ->  [x~; z~],   [[P + sigma I, A'], [A, -diag(1/rho)]] [x~; nu] = b,   z~ = rhs_z + nu / rho

A C host binds the same table of function pointers directly."""
import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib
from .ext_hip import CSC, OSQPSettings


class LinSysSolver:
    def __init__(self, P, A, rho_vec, sigma=1e-6, polishing=False, scaled_residuals=None, **settings):
        """P: full symmetric or upper-triangular (n x n), A: m x n, rho_vec: m positive weights -- all ALREADY SCALED.
        scaled_residuals: optional float64 array [prim_res, dual_res] the caller keeps up to date (the PCG tolerance follows
        cg_tol_fraction * dual_res, as inside the ADMM engine)."""
        self._lib = _lib.handle()
        self._P = CSC(sp.triu(sp.csc_matrix(P), format='csc')); self._Pc = self._P._struct()
        self._A = CSC(sp.csc_matrix(A)); self._Ac = self._A._struct()
        self.n, self.m = self._A.n, self._A.m
        st = OSQPSettings()
        self._lib.osqp_set_default_settings(C.byref(st))
        st.sigma = sigma
        for k, v in settings.items():
            setattr(st, k, v)
        self._st = st
        self._res = scaled_residuals
        rho = np.ascontiguousarray(rho_vec, dtype=np.float64)
        pr = self._res.ctypes.data_as(_lib.c_double_p) if self._res is not None else None
        du = C.cast(C.addressof(pr.contents) + 8, _lib.c_double_p) if self._res is not None else None
        self._p = _lib.LinSysP()
        err = self._lib.osqp_hip_linsys_init(C.byref(self._p), C.byref(self._Pc), C.byref(self._Ac), rho.ctypes.data_as(_lib.c_double_p),
                                             C.byref(st), pr, du, int(bool(polishing)))
        if err:
            raise ValueError(str(err))
        self._t = self._p.contents

    @property
    def name(self):
        return self._t.name(self._p).decode()

    @property
    def pcg_iters(self):
        return self._p.contents.pcg_iters

    def solve(self, b, admm_iter=1):
        b = np.array(b, dtype=np.float64)
        assert b.shape == (self.n + self.m,)
        err = self._t.solve(self._p, b.ctypes.data_as(_lib.c_double_p), int(admm_iter))
        if err:
            raise ValueError(str(err))
        return b

    def warm_start(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        self._t.warm_start(self._p, x.ctypes.data_as(_lib.c_double_p))

    def update_rho_vec(self, rho_vec, rho_sc=0.0):
        rho = np.ascontiguousarray(rho_vec, dtype=np.float64)
        err = self._t.update_rho_vec(self._p, rho.ctypes.data_as(_lib.c_double_p), float(rho_sc))
        if err:
            raise ValueError(str(err))

    def update_matrices(self, P=None, A=None):
        """Same sparsity pattern, new values."""
        if P is not None:
            self._P = CSC(sp.triu(sp.csc_matrix(P), format='csc')); self._Pc = self._P._struct()
        if A is not None:
            self._A = CSC(sp.csc_matrix(A)); self._Ac = self._A._struct()
        err = self._t.update_matrices(self._p, C.byref(self._Pc) if P is not None else None, None, 0,
                                      C.byref(self._Ac) if A is not None else None, None, 0)
        if err:
            raise ValueError(str(err))

    def update_settings(self, **kw):
        for k, v in kw.items():
            setattr(self._st, k, v)
        self._t.update_settings(self._p, C.byref(self._st))

    def adjoint_derivative(self):
        return self._t.adjoint_derivative(self._p)

    def free(self):
        if self._p:
            self._t.free(self._p)
            self._p = _lib.LinSysP()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
