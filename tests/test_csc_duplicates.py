"""Valid CSC input with an entry stored twice: scipy does not sum duplicates on construction and neither does the
reference's binding (bindings.cpp.in:12-62 hands the arrays through), so the C ABI must cope: every stored (j, j) entry of P adds
into the one diagonal slot of B = [P + sigma I | A'] (engine_setup.cpp Engine::setup, k_asm_scatter)."""
import warnings

import numpy as np
import numpy.testing as npt
import pytest
import scipy.sparse as sp

import osqp_amd
from backend_param import BACKENDS, engine
from oracle import Oracle, SOLVED

warnings.simplefilter('ignore')


@pytest.mark.parametrize('backend', BACKENDS)
def test_repeated_diagonal_entries_of_P_sum_up(backend):
    n, m = 6, 8
    rng = np.random.default_rng(2)
    A = sp.random(m, n, density=0.5, random_state=rng, data_rvs=rng.standard_normal, format='csc')
    q = rng.standard_normal(n); l = -np.ones(m); u = np.ones(m)
    indptr, indices, data = [0], [], []
    for j in range(n):                                   # column j: (0, j) [j > 0], then the diagonal stored TWICE: 0.7 + 0.5
        if j > 0:
            indices.append(0); data.append(0.1)
        indices += [j, j]; data += [0.7, 0.5]
        indptr.append(len(indices))
    Pdup = sp.csc_matrix((np.array(data), np.array(indices), np.array(indptr)), shape=(n, n))
    assert Pdup.nnz == 3 * n - 1                         # the duplicates really are stored
    with engine(backend):
        ext = osqp_amd.interface._backend('hip')
        st = ext.OSQPSettings(); ext.osqp_set_default_settings(st)
        st.verbose = 0; st.eps_abs = st.eps_rel = 1e-7
        solver = ext.OSQPSolver(ext.CSC(Pdup), q, ext.CSC(A), l, u, m, n, st)      # straight through the C ABI
        solver.solve()
        x, status = np.array(solver.solution.x), solver.info.status_val
    Psum = Pdup.copy(); Psum.sum_duplicates()
    Pfull = (Psum + sp.triu(Psum, 1).T).tocsc()
    xo, yo, io = Oracle().setup(Pfull, q, A, l, u, eps_abs=1e-9, eps_rel=1e-9, adaptive_rho_interval=50, max_iter=50000).solve()
    assert status == 1 and io.status_val == SOLVED
    npt.assert_allclose(x, xo, rtol=0, atol=1e-5 * (1 + np.abs(xo).max()))
