"""osqp_amd -- MI355X-native OSQP ADMM engine behind the osqp.OSQP().setup/solve/update/warm_start API.

Same names as the reference's /root/reference/src/osqp/__init__.py:5-13.
"""
from osqp_amd.interface import (  # noqa: F401
    OSQPException,
    OSQP,
    constant,
    algebra_available,
    algebras_available,
    default_algebra,
    SolverStatus,
    SolverError,
)

__version__ = '1.0.0+hip.r1'
