"""Per-algebra subclass, like the reference's src/osqp/cuda.py:1-6."""
from osqp_amd.interface import OSQP as _OSQP


class OSQP(_OSQP):
    def __init__(self, *args, **kwargs):
        super(OSQP, self).__init__(*args, **kwargs, algebra='hip')
