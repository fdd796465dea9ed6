"""Front-end of the hip algebra: mirrors the public API of the reference's /root/reference/src/osqp/interface.py
(class OSQP :120-434, helpers :28-141) for the one path this engine replaces: setup / solve / update /
update_settings / warm_start.  Code generation and adjoint derivatives (:436-598) are out of scope.
"""
import importlib
import os
import warnings
from enum import IntEnum
from types import SimpleNamespace

import numpy as np
import scipy.sparse as spa

_ALGEBRAS = ('hip',)                                   # cf. interface.py:14-24
_ALGEBRA_MODULES = {'hip': 'osqp_amd.ext_hip'}
OSQP_ALGEBRA_BACKEND = os.environ.get('OSQP_ALGEBRA_BACKEND')


def algebra_available(algebra):                        # interface.py:28-37
    assert algebra in _ALGEBRAS, f'Unknown algebra {algebra}'
    try:
        m = importlib.import_module(_ALGEBRA_MODULES[algebra])
        m._lib.handle()
    except ImportError:
        return False
    return True


def algebras_available():
    return [a for a in _ALGEBRAS if algebra_available(a)]


def default_algebra():                                 # interface.py:44-50
    if OSQP_ALGEBRA_BACKEND is not None:
        return OSQP_ALGEBRA_BACKEND
    for a in _ALGEBRAS:
        if algebra_available(a):
            return a
    raise RuntimeError('No algebra backend available!')


def default_algebra_module():
    return importlib.import_module(_ALGEBRA_MODULES['hip'])


def constant(which, algebra='hip'):                    # interface.py:62-89
    m = importlib.import_module(_ALGEBRA_MODULES[algebra])
    c = getattr(m, which, None)
    if which in m.osqp_status_type.__members__:
        warnings.warn('Direct access to osqp status values will be deprecated. Please use the SolverStatus enum instead.',
                      PendingDeprecationWarning)
    if isinstance(c, IntEnum):
        return c.value
    if isinstance(c, (int, float, str)):
        return c
    if which == 'OSQP_NAN':
        return np.nan
    raise RuntimeError(f'Unknown constant {which}')


def _enum(name, binding_enum):                         # interface.py:92-103
    return IntEnum(name, [(v.name, v.value) for v in binding_enum.__members__.values()])


_ext = default_algebra_module()
SolverStatus = _enum('SolverStatus', _ext.osqp_status_type)
SolverError = _enum('SolverError', _ext.osqp_error_type)


class OSQPException(Exception):                        # interface.py:106-117
    def __init__(self, error_code=None):
        if error_code:
            self.args = (error_code,)

    def __eq__(self, error_code):
        return len(self.args) > 0 and self.args[0] == error_code

    __hash__ = Exception.__hash__


class OSQP:
    @classmethod
    def raises_error(cls, fn, *args, **kwargs):        # interface.py:128-141
        try:
            return fn(*args, **kwargs)
        except ValueError as e:
            code = None
            if e.args:
                try:
                    code = int(e.args[0])
                except ValueError:
                    pass
            raise OSQPException(code)

    def __init__(self, *args, **kwargs):
        self.m = None
        self.n = None
        self.algebra = kwargs.pop('algebra') if 'algebra' in kwargs else default_algebra()
        if self.algebra not in _ALGEBRAS or not algebra_available(self.algebra):
            raise RuntimeError(f'Algebra {self.algebra} not available')
        self.ext = importlib.import_module(_ALGEBRA_MODULES[self.algebra])
        self._dtype = np.float32 if self.ext.OSQP_USE_FLOAT == 1 else np.float64
        self._itype = np.int64 if self.ext.OSQP_USE_LONG == 1 else np.int32
        self._solver = None
        self.settings = None

    def __str__(self):
        if self._solver is None:
            return f'Uninitialized OSQP with algebra={self.algebra}'
        return f'OSQP with algebra={self.algebra} ({self.solver_type})'

    # ---- problem inference: interface.py:165-240 ----
    def _infer_mnpqalu(self, P=None, q=None, A=None, l=None, u=None):
        if P is None:
            if q is not None:
                n = len(q)
            elif A is not None:
                n = A.shape[1]
            else:
                raise ValueError('The problem does not have any variables')
        else:
            n = P.shape[0]
        m = 0 if A is None else A.shape[0]
        if A is None:
            assert (l is None) and (u is None), 'If A is unspecified, leave l/u unspecified too.'
        else:
            assert (l is not None) or (u is not None), 'If A is specified, specify at least one of l/u.'
            if l is None:
                l = -np.inf * np.ones(m)
            if u is None:
                u = np.inf * np.ones(m)
        if P is None:
            P = spa.csc_matrix((n, n), dtype=self._dtype)
        if q is None:
            q = np.zeros(n)
        if A is None:
            A = spa.csc_matrix((0, n), dtype=self._dtype)
            l = np.zeros(0)
            u = np.zeros(0)
        assert len(q) == n, 'Incorrect dimension of q'
        assert len(l) == m, 'Incorrect dimension of l'
        assert len(u) == m, 'Incorrect dimension of u'
        if not spa.issparse(P) and isinstance(P, np.ndarray) and P.ndim == 2:
            raise TypeError('P is required to be a sparse matrix')
        if not spa.issparse(A) and isinstance(A, np.ndarray) and A.ndim == 2:
            raise TypeError('A is required to be a sparse matrix')
        if spa.tril(P, -1).data.size > 0:              # :221-222 keep the upper triangle
            P = spa.triu(P, format='csc')
        if not spa.isspmatrix_csc(P):
            warnings.warn('Converting sparse P to a CSC matrix. This may take a while...')
            P = P.tocsc()
        if not spa.isspmatrix_csc(A):
            warnings.warn('Converting sparse A to a CSC matrix. This may take a while...')
            A = A.tocsc()
        if not P.has_sorted_indices:
            P.sort_indices()
        if not A.has_sorted_indices:
            A.sort_indices()
        u = np.minimum(u, self.constant('OSQP_INFTY'))  # :237-238
        l = np.maximum(l, -self.constant('OSQP_INFTY'))
        return m, n, P, q, A, l, u

    # ---- properties: interface.py:242-264 ----
    @property
    def capabilities(self):
        return int(self.ext.osqp_capabilities())

    def has_capability(self, capability: str):
        try:
            cap = int(self.ext.osqp_capabilities_type.__members__[capability])
        except KeyError:
            raise RuntimeError(f'Unrecognized capability {capability}')
        return (self.capabilities & cap) != 0

    @property
    def solver_type(self):
        return 'direct' if self.settings.linsys_solver == self.ext.osqp_linsys_solver_type.OSQP_DIRECT_SOLVER else 'indirect'

    @property
    def cg_preconditioner(self):
        return 'diagonal' if self.settings.cg_precond == self.ext.OSQP_DIAGONAL_PRECONDITIONER else None

    def constant(self, which):
        return constant(which, algebra=self.algebra)

    # ---- settings: interface.py:280-328 ----
    def update_settings(self, **kwargs):
        assert self.settings is not None
        for old, new in {'polish': 'polishing', 'warm_start': 'warm_starting'}.items():
            if old in kwargs:
                warnings.warn(f'"{old}" is deprecated. Please use "{new}" instead.', DeprecationWarning)
                kwargs[new] = kwargs.pop(old)
        changed = False
        if 'rho' in kwargs and self._solver is not None:
            self._solver.update_rho(kwargs.pop('rho'))
        if 'solver_type' in kwargs:
            value = kwargs.pop('solver_type')
            assert value in ('direct', 'indirect')
            self.settings.linsys_solver = (self.ext.osqp_linsys_solver_type.OSQP_DIRECT_SOLVER if value == 'direct'
                                           else self.ext.osqp_linsys_solver_type.OSQP_INDIRECT_SOLVER)
            changed = True
        if 'cg_preconditioner' in kwargs:
            value = kwargs.pop('cg_preconditioner')
            assert value in (None, 'diagonal')
            self.settings.cg_precond = (self.ext.OSQP_DIAGONAL_PRECONDITIONER if value == 'diagonal'
                                        else self.ext.OSQP_NO_PRECONDITIONER)
            changed = True
        for k, _ in self.ext.OSQPSettings._fields_:
            if k in kwargs:
                setattr(self.settings, k, kwargs.pop(k))
                changed = True
        if kwargs:
            raise ValueError(f'Unrecognized settings {list(kwargs.keys())}')
        if changed and self._solver is not None:
            self.raises_error(self._solver.update_settings, self.settings)

    # ---- data updates: interface.py:330-347 ----
    def update(self, **kwargs):
        q, l, u = kwargs.get('q'), kwargs.get('l'), kwargs.get('u')
        if l is not None:
            l = np.maximum(l, -self.constant('OSQP_INFTY'))
        if u is not None:
            u = np.minimum(u, self.constant('OSQP_INFTY'))
        if q is not None or l is not None or u is not None:
            self._solver.update_data_vec(q=q, l=l, u=u)
        if any(k in kwargs for k in ('Px', 'Px_idx', 'Ax', 'Ax_idx')):
            self._solver.update_data_mat(P_x=kwargs.get('Px'), P_i=kwargs.get('Px_idx'),
                                         A_x=kwargs.get('Ax'), A_i=kwargs.get('Ax_idx'))

    # ---- setup / warm start / solve: interface.py:370-434 ----
    def setup(self, P, q, A, l, u, **settings):
        m, n, P, q, A, l, u = self._infer_mnpqalu(P=P, q=q, A=A, l=l, u=u)
        self.m, self.n = m, n
        P = self.ext.CSC(P.astype(self._dtype))
        q = np.asarray(q).astype(self._dtype)
        A = self.ext.CSC(A.astype(self._dtype))
        l = np.asarray(l).astype(self._dtype)
        u = np.asarray(u).astype(self._dtype)
        self.settings = self.ext.OSQPSettings()
        self.ext.osqp_set_default_settings(self.settings)
        self._solver = None
        self.update_settings(**settings)
        self._solver = self.raises_error(self.ext.OSQPSolver, P, q, A, l, u, self.m, self.n, self.settings)
        if 'rho' in settings:                          # :396-397
            self._solver.update_rho(settings['rho'])

    def warm_start(self, x=None, y=None):
        return self._solver.warm_start(x, y)

    def solve(self, raise_error=None):
        if raise_error is None:
            warnings.warn('The default value of raise_error will change to True in the future.', PendingDeprecationWarning)
            raise_error = False
        self._solver.solve()
        info = self._solver.info
        if info.status_val == SolverStatus.OSQP_NON_CVX:   # :414-415
            info.obj_val = np.nan
        if info.status_val != SolverStatus.OSQP_SOLVED and raise_error:
            raise OSQPException(info.status_val)
        _info = SimpleNamespace(**{k: getattr(info, k) for k in info.__class__.__dict__ if not k.startswith('_')})
        return SimpleNamespace(x=self._solver.solution.x, y=self._solver.solution.y,
                               prim_inf_cert=self._solver.solution.prim_inf_cert,
                               dual_inf_cert=self._solver.solution.dual_inf_cert, info=_info)

    # ---- out of scope ----
    def codegen(self, *a, **k):
        raise NotImplementedError('code generation is out of scope of the MI355X engine (SURVEY.md §2 row 7)')

    def adjoint_derivative_compute(self, *a, **k):
        raise NotImplementedError('adjoint derivatives are out of scope of the MI355X engine (SURVEY.md §2 row 6)')

    adjoint_derivative_get_mat = adjoint_derivative_get_vec = adjoint_derivative_compute
