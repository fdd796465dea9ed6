"""Python front-end of the MI355X engine.

Public surface = the part of the reference's ``osqp.interface`` that drives the hot path (``OSQP.setup / solve / update /
update_settings / warm_start``, the module helpers and the two status enums; /root/reference/src/osqp/interface.py:28-141,
:120-434).  Names, keyword arguments, exception types and the numeric error mapping are the reference's -- they are what
``tests/test_reference_suite.py::test_frontend_contract`` pins -- the implementation is this repository's own: one
normalisation pipeline for the problem data, one table for the settings that are not plain ``OSQPSettings`` fields.
Code generation and the adjoint-derivative entry points are out of scope and say so when called.
"""
import functools
import importlib as _importlib
import os as _os
import warnings as _warnings
from enum import IntEnum as _IntEnum
from types import SimpleNamespace as _Namespace

import numpy as np
from scipy import sparse as spa

# algebra name -> extension module that binds the C ABI (the reference keeps the same kind of table, interface.py:14-24)
_BACKENDS = {'hip': 'osqp_amd.ext_hip'}
OSQP_ALGEBRA_BACKEND = _os.environ.get('OSQP_ALGEBRA_BACKEND')


@functools.lru_cache(maxsize=None)
def _backend(algebra):
    if algebra not in _BACKENDS:
        raise AssertionError(f'Unknown algebra {algebra}')
    return _importlib.import_module(_BACKENDS[algebra])


def algebra_available(algebra):
    """True when the algebra's extension module imports and its native library loads."""
    try:
        _backend(algebra)._lib.handle()
    except (ImportError, OSError):
        return False
    return True


def algebras_available():
    return [name for name in _BACKENDS if algebra_available(name)]


def default_algebra():
    """``$OSQP_ALGEBRA_BACKEND`` if set, else the first algebra that loads."""
    if OSQP_ALGEBRA_BACKEND:
        return str(OSQP_ALGEBRA_BACKEND)
    usable = algebras_available()
    if not usable:
        raise RuntimeError('No algebra backend available!')
    return usable[0]


def constant(which, algebra='hip'):
    """Value of a named constant of the extension module (``OSQP_INFTY``, a status or error name, ...)."""
    ext = _backend(algebra)
    if which in ext.osqp_status_type.__members__:
        _warnings.warn('Direct access to osqp status values will be deprecated. Please use the SolverStatus enum instead.',
                      PendingDeprecationWarning)
    if which == 'OSQP_NAN' and not hasattr(ext, which):
        return np.nan
    value = getattr(ext, which, None)
    if isinstance(value, _IntEnum):
        return int(value)
    if isinstance(value, (int, float, str)):
        return value
    raise RuntimeError(f'Unknown constant {which}')


def _mirror_enum(name, source):
    return _IntEnum(name, {member.name: int(member) for member in source})


SolverStatus = _mirror_enum('SolverStatus', _backend('hip').osqp_status_type)
SolverError = _mirror_enum('SolverError', _backend('hip').osqp_error_type)


class OSQPException(Exception):
    """Raised for a failed setup / settings update (``SolverError`` code) or, with ``raise_error``, an unsolved problem
    (``SolverStatus`` code).  Compares equal to its integer code, as the reference's exception does."""

    def __init__(self, error_code=None):
        super().__init__(*(() if not error_code else (error_code,)))

    @property
    def code(self):
        return self.args[0] if self.args else None

    def __eq__(self, other):
        return self.code is not None and self.code == other

    __hash__ = Exception.__hash__


def _require(condition, message):
    if not condition:
        raise AssertionError(message)


def _sparse_csc(matrix, label, dtype, upper_only=False):
    """CSC copy of a scipy sparse matrix with sorted indices (its upper triangle if ``upper_only``); dense 2-D arrays are
    rejected like the reference does."""
    if isinstance(matrix, np.ndarray) and matrix.ndim == 2:
        raise TypeError(f'{label} is required to be a sparse matrix')
    if upper_only and spa.tril(matrix, -1).nnz:
        matrix = spa.triu(matrix, format='csc')
    if not spa.isspmatrix_csc(matrix):
        _warnings.warn(f'Converting sparse {label} to a CSC matrix. This may take a while...')
        matrix = matrix.tocsc()
    if not matrix.has_sorted_indices:
        matrix.sort_indices()
    return matrix.astype(dtype)


# keyword settings that are not OSQPSettings fields: old spellings, and the two derived from an enum-valued field
_RENAMED = {'polish': 'polishing', 'warm_start': 'warm_starting'}
_ENUM_SETTINGS = {
    # keyword            field            {keyword value: enum member name}
    'solver_type': ('linsys_solver', {'direct': 'OSQP_DIRECT_SOLVER', 'indirect': 'OSQP_INDIRECT_SOLVER'}),
    'cg_preconditioner': ('cg_precond', {None: 'OSQP_NO_PRECONDITIONER', 'diagonal': 'OSQP_DIAGONAL_PRECONDITIONER'}),
}


class OSQP:
    def __init__(self, *args, algebra=None, **kwargs):
        self.algebra = default_algebra() if algebra is None else algebra
        if self.algebra not in _BACKENDS or not algebra_available(self.algebra):
            raise RuntimeError(f'Algebra {self.algebra} not available')
        self.ext = _backend(self.algebra)
        self._dtype = (np.float64, np.float32)[int(self.ext.OSQP_USE_FLOAT == 1)]      # the C ABI's OSQPFloat / OSQPInt
        self._itype = (np.int32, np.int64)[int(self.ext.OSQP_USE_LONG == 1)]
        self.m = self.n = None
        self.settings = None
        self._solver = None

    def __str__(self):
        state = 'Uninitialized OSQP' if self._solver is None else 'OSQP'
        detail = '' if self._solver is None else f' ({self.solver_type})'
        return f'{state} with algebra={self.algebra}{detail}'

    # ------------------------------------------------------------------ helpers
    @classmethod
    def raises_error(cls, fn, *args, **kwargs):
        """Call ``fn``; the extension reports a failed C call as ``ValueError(str(code))`` -- re-raise as OSQPException(code)."""
        try:
            return fn(*args, **kwargs)
        except ValueError as err:
            text = str(err.args[0]) if err.args else ''
            raise OSQPException(int(text) if text.lstrip('-').isdigit() else None)

    def constant(self, which):
        return constant(which, self.algebra)

    @property
    def capabilities(self):
        mask = self.ext.osqp_capabilities()
        return int(mask)

    def has_capability(self, capability: str):
        members = self.ext.osqp_capabilities_type.__members__
        if capability not in members:
            raise RuntimeError(f'Unrecognized capability {capability}')
        return bool(self.capabilities & int(members[capability]))

    def _enum_setting(self, keyword):
        field, names = _ENUM_SETTINGS[keyword]
        current = int(getattr(self.settings, field))
        for value, member in names.items():
            if int(getattr(self.ext, member)) == current:
                return value
        return None

    @property
    def solver_type(self):
        return self._enum_setting('solver_type')

    @property
    def cg_preconditioner(self):
        return self._enum_setting('cg_preconditioner')

    # ------------------------------------------------------------------ problem data
    def _infer_mnpqalu(self, P=None, q=None, A=None, l=None, u=None):
        """(m, n, P, q, A, l, u) with every missing piece filled in: dimensions from whichever of P / q / A is given, an empty
        P or A, infinite bounds on the missing side; P reduced to its upper triangle, both matrices CSC with sorted indices,
        bounds clipped to +-OSQP_INFTY."""
        shapes = [P.shape[0] if P is not None else None, len(q) if q is not None else None, A.shape[1] if A is not None else None]
        known = [s for s in shapes if s is not None]
        if not known:
            raise ValueError('The problem does not have any variables')
        n = known[0]
        if A is None:
            _require(l is None and u is None, 'If A is unspecified, leave l/u unspecified too.')
            m, A, l, u = 0, spa.csc_matrix((0, n), dtype=self._dtype), np.zeros(0), np.zeros(0)
        else:
            _require(l is not None or u is not None, 'If A is specified, specify at least one of l/u.')
            m = A.shape[0]
            l = np.full(m, -np.inf) if l is None else l
            u = np.full(m, np.inf) if u is None else u
        P = spa.csc_matrix((n, n), dtype=self._dtype) if P is None else P
        q = np.zeros(n) if q is None else q
        for name, vec, size in (('q', q, n), ('l', l, m), ('u', u, m)):
            _require(len(vec) == size, f'Incorrect dimension of {name}')
        P = _sparse_csc(P, 'P', self._dtype, upper_only=True)
        A = _sparse_csc(A, 'A', self._dtype)
        big = self.constant('OSQP_INFTY')
        return m, n, P, np.asarray(q, dtype=self._dtype), A, np.clip(l, -big, None).astype(self._dtype), np.clip(u, None, big).astype(self._dtype)

    # ------------------------------------------------------------------ settings
    def update_settings(self, **kwargs):
        _require(self.settings is not None, 'setup() has to be called first')
        for old, new in _RENAMED.items():
            if old in kwargs:
                _warnings.warn(f'"{old}" is deprecated. Please use "{new}" instead.', DeprecationWarning)
                kwargs[new] = kwargs.pop(old)
        if 'rho' in kwargs and self._solver is not None:          # rho of a live solver goes through its own entry point
            new_rho = kwargs.pop('rho')
            self._solver.update_rho(new_rho)
        assignments = {}
        for keyword in [k for k in kwargs if k in _ENUM_SETTINGS]:
            field, names = _ENUM_SETTINGS[keyword]
            value = kwargs.pop(keyword)
            _require(value in names, f'{keyword} must be one of {list(names)}')
            assignments[field] = getattr(self.ext, names[value])
        fields = {name for name, _ in self.ext.OSQPSettings._fields_}
        unknown = [k for k in kwargs if k not in fields]
        if unknown:
            raise ValueError(f'Unrecognized settings {unknown}')
        assignments.update(kwargs)
        for field, value in assignments.items():
            setattr(self.settings, field, value)
        if assignments and self._solver is not None:
            push = self._solver.update_settings
            self.raises_error(push, self.settings)

    # ------------------------------------------------------------------ life cycle
    def setup(self, P, q, A, l, u, **settings):
        self.m, self.n, P, q, A, l, u = self._infer_mnpqalu(P=P, q=q, A=A, l=l, u=u)
        self._solver = None
        self.settings = defaults = self.ext.OSQPSettings()
        self.ext.osqp_set_default_settings(defaults)
        self.update_settings(**settings)                           # validates names; nothing is sent yet (no solver)
        self._solver = self.raises_error(self.ext.OSQPSolver, self.ext.CSC(P), q, self.ext.CSC(A), l, u, self.m, self.n, self.settings)
        rho = settings.get('rho')
        if rho is not None:
            self._solver.update_rho(rho)

    def update(self, **kwargs):
        """New q / l / u and / or new values Px, Ax (optionally at the positions Px_idx, Ax_idx of the CSC data arrays)."""
        big = self.constant('OSQP_INFTY')
        vectors = {'q': kwargs.get('q'),
                   'l': None if kwargs.get('l') is None else np.maximum(kwargs['l'], -big),
                   'u': None if kwargs.get('u') is None else np.minimum(kwargs['u'], big)}
        if any(v is not None for v in vectors.values()):
            self._solver.update_data_vec(**vectors)
        if {'Px', 'Px_idx', 'Ax', 'Ax_idx'} & set(kwargs):
            self._solver.update_data_mat(P_x=kwargs.get('Px'), P_i=kwargs.get('Px_idx'), A_x=kwargs.get('Ax'), A_i=kwargs.get('Ax_idx'))

    def warm_start(self, x=None, y=None):
        return self._solver.warm_start(x, y)

    def solve(self, raise_error=None):
        if raise_error is None:
            _warnings.warn('The default value of raise_error will change to True in the future.', PendingDeprecationWarning)
        strict = bool(raise_error)                 # (None: today's default, report through info.status_val only)
        self._solver.solve()
        raw = self._solver.info
        if raw.status_val == SolverStatus.OSQP_NON_CVX:
            raw.obj_val = np.nan
        if strict and raw.status_val != SolverStatus.OSQP_SOLVED:
            raise OSQPException(raw.status_val)
        info = _Namespace(**{name: getattr(raw, name) for name in vars(type(raw)) if not name.startswith('_')})
        sol = self._solver.solution
        return _Namespace(x=sol.x, y=sol.y, prim_inf_cert=sol.prim_inf_cert, dual_inf_cert=sol.dual_inf_cert, info=info)

    # ------------------------------------------------------------------ not part of this engine
    def _out_of_scope(self, *_args, **_kwargs):
        raise NotImplementedError('code generation and adjoint derivatives are outside the MI355X engine (SURVEY.md section 2)')

    codegen = adjoint_derivative_compute = adjoint_derivative_get_mat = adjoint_derivative_get_vec = _out_of_scope
