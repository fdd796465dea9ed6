"""Shared helpers for the tests: fixture loading (tests/golden/*.npz) and problem generators."""
import json
import os

import numpy as np
import scipy.sparse as sp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


class Fixture:
    def __init__(self, name):
        d = np.load(os.path.join(GOLDEN, name + '.npz'))
        self.name = name
        self.raw = d
        self.n, self.m = int(d['n']), int(d['m'])
        self.P = sp.csc_matrix((d['P_data'], d['P_indices'], d['P_indptr']), shape=(self.n, self.n))   # upper triangle
        self.A = sp.csc_matrix((d['A_data'], d['A_indices'], d['A_indptr']), shape=(self.m, self.n))
        self.q, self.l, self.u = d['q'], d['l'], d['u']
        self.settings = json.loads(str(d['settings']))

    def has(self, k):
        return k in self.raw.files

    def __getitem__(self, k):
        return self.raw[k]

    def oracle_settings(self, **over):
        """purepy-style settings dict -> oracle settings"""
        s = dict(self.settings)
        s.pop('polish', None)
        s['check_termination'] = int(s['check_termination'])
        for k in ('adaptive_rho', 'warm_start', 'scaled_termination'):
            s[k] = int(s[k])
        s.update(over)
        return s

    def hip_settings(self, **over):
        """purepy-style settings dict -> osqp v1 front-end kwargs"""
        s = dict(self.settings)
        s['polishing'] = bool(s.pop('polish', False))
        s['warm_starting'] = bool(s.pop('warm_start', True))
        s['check_termination'] = int(s['check_termination'])
        s['verbose'] = False
        s.update(over)
        return s


def free_port():
    """An ephemeral TCP port for a torch.distributed rendezvous on 127.0.0.1 (fixed numbers collide when tests run in parallel)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def record_deviation(test, case, **vals):
    """Measured deviations of a parity test (|dx|, |dy| relative to the solution's scale, iteration counts ...) appended to
    gpurun_out/parity_deviations.json -- the GPU run leaves the numbers behind even when pytest runs with -q (a copy of the round's
    file is kept under profiles/).  Best effort: a read-only tree must not fail a test."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, 'gpurun_out', 'parity_deviations.json')
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = {}
        if os.path.exists(path):
            with open(path) as f:
                data = json.load(f)
        data.setdefault(test, {})[case] = {k: (float(v) if isinstance(v, (float, np.floating)) else int(v) if isinstance(v, (int, np.integer)) else v) for k, v in vals.items()}
        with open(path, 'w') as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except OSError:
        pass
