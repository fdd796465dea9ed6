"""GPU tier: a batch whose elements have their OWN matrix values (osqp_hip_batch_solve_mat; the reference's forward with a P_val / A_val per batch
element builds one solver per element, /root/reference/src/osqp/nn/torch.py:128-157, 184-217).  One solve launch; every element is assembled and
equilibrated with its own data, so it must behave exactly as a solver set up with that element's matrices alone: iteration counts equal the
oracle's per element, x / y to 1e-7 -- and equal to this engine's own single-QP solve of the element."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import osqp_amd
import problems
from oracle import Oracle, SOLVED
from util import record_deviation

pytestmark = pytest.mark.gpu
warnings.simplefilter('ignore')
EPS = 1e-6
ST = dict(eps_abs=EPS, eps_rel=EPS, verbose=False, max_iter=4000)
OST = dict(eps_abs=EPS, eps_rel=EPS, max_iter=4000, adaptive_rho_interval=50, check_termination=25)


def _perturbed(B, seed=3):
    """B MPC problems with their own dynamics: the entries of A (other than the +-1 of the identity / box / rate rows) and the diagonal of P jittered per element"""
    P, q, A, L, U = problems.mpc_batch(B, seed=11)
    rng = np.random.default_rng(seed)
    Pt = sp.triu(P, format='csc')
    sel = np.abs(np.abs(A.data) - 1.0) > 1e-12
    Ax = np.tile(A.data, (B, 1)); Px = np.tile(Pt.data, (B, 1))
    Ax[:, sel] *= 1 + 0.1 * rng.standard_normal((B, int(sel.sum())))
    Px *= 1 + 0.2 * rng.random((B, Pt.nnz))
    Q = 0.1 * rng.standard_normal((B, P.shape[0]))
    return P, Pt, q, A, L, U, Px, Ax, Q


def _element(Pt, A, Px, Ax, i):
    Pi = sp.csc_matrix((Px[i], Pt.indices, Pt.indptr), shape=Pt.shape)
    Pi = (Pi + Pi.T - sp.diags(Pi.diagonal())).tocsc()
    Ai = sp.csc_matrix((Ax[i], A.indices, A.indptr), shape=A.shape)
    return Pi, Ai


@pytest.mark.parametrize('B', [5, 96])
def test_per_element_matrices_match_the_oracle_element_by_element(B):
    P, Pt, q, A, L, U, Px, Ax, Q = _perturbed(B)
    s = osqp_amd.OSQP(algebra='hip'); s.setup(P, q, A, L[0], U[0], **ST)
    launches = []
    x, y, rec = s._solver.hip_batch_solve(q=Q, l=L, u=U, Px=Px, Ax=Ax)
    assert (rec[:, 0] == 1).all(), rec[:, 0]
    worst = 0.0
    for i in sorted(set([0, 1, B // 2, B - 1])):
        Pi, Ai = _element(Pt, A, Px, Ax, i)
        xo, yo, io = Oracle().setup(Pi, Q[i], Ai, L[i], U[i], **OST).solve()
        assert io.status_val == SOLVED
        assert int(rec[i, 1]) == io.iter, (i, rec[i, 1], io.iter)                   # the same algorithm with the same scaling: the same course
        ex = np.abs(x[i] - xo).max() / (1 + np.abs(xo).max()); ey = np.abs(y[i] - yo).max() / (1 + np.abs(yo).max())
        worst = max(worst, ex, ey)
        assert ex <= 1e-7 and ey <= 1e-7, (i, ex, ey)
        assert abs(rec[i, 2] - io.obj_val) <= 1e-8 * (1 + abs(io.obj_val))
    record_deviation('test_per_element_matrices_match_the_oracle', 'B=%d' % B, worst_rel=worst)
    # ... and as this engine's own single-QP solve of one element (a handle set up with that element's data: its one-launch direct path)
    i = B - 1
    Pi, Ai = _element(Pt, A, Px, Ax, i)
    s1 = osqp_amd.OSQP(algebra='hip'); s1.setup(Pi, Q[i], Ai, L[i], U[i], **ST)
    r1 = s1.solve()
    assert r1.info.iter == int(rec[i, 1]) and np.abs(r1.x - x[i]).max() <= 1e-9 * (1 + np.abs(x[i]).max())


def test_only_A_or_only_P_per_element():
    B = 7
    P, Pt, q, A, L, U, Px, Ax, Q = _perturbed(B, seed=9)
    s = osqp_amd.OSQP(algebra='hip'); s.setup(P, q, A, L[0], U[0], **ST)
    xa, ya, ra = s._solver.hip_batch_solve(q=Q, l=L, u=U, Ax=Ax)                      # P shared (the handle's), A per element
    xp, yp, rp = s._solver.hip_batch_solve(q=Q, l=L, u=U, Px=Px)                      # A shared, P per element
    assert (ra[:, 0] == 1).all() and (rp[:, 0] == 1).all()
    for i in (0, 6):
        Pi, Ai = _element(Pt, A, np.tile(Pt.data, (B, 1)), Ax, i)
        xo, _, io = Oracle().setup(Pi, Q[i], Ai, L[i], U[i], **OST).solve()
        assert int(ra[i, 1]) == io.iter and np.abs(xa[i] - xo).max() <= 1e-7 * (1 + np.abs(xo).max())
        Pi, Ai = _element(Pt, A, Px, np.tile(A.data, (B, 1)), i)
        xo, _, io = Oracle().setup(Pi, Q[i], Ai, L[i], U[i], **OST).solve()
        assert int(rp[i, 1]) == io.iter and np.abs(xp[i] - xo).max() <= 1e-7 * (1 + np.abs(xo).max())
    # identical matrices for every element = the shared-matrix batch (same kernel variant, other scaling route): same solutions
    xs, ys, rs = s._solver.hip_batch_solve(q=Q, l=L, u=U)
    xe, ye, re_ = s._solver.hip_batch_solve(q=Q, l=L, u=U, Px=np.tile(Pt.data, (B, 1)), Ax=np.tile(A.data, (B, 1)))
    assert np.abs(xs - xe).max() <= 2e-5 * (1 + np.abs(xs).max())                    # (two eps = 1e-6 iterates: the per-element call scales with every element's own q)


def test_device_pointer_entry_point_equals_the_host_one():
    import torch
    B = 33
    P, Pt, q, A, L, U, Px, Ax, Q = _perturbed(B, seed=5)
    s = osqp_amd.OSQP(algebra='hip'); s.setup(P, q, A, L[0], U[0], **ST)
    xh, yh, rh = s._solver.hip_batch_solve(q=Q, l=L, u=U, Px=Px, Ax=Ax)
    dev = torch.device('cuda', 0)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    Qd, Ld, Ud, Pd, Ad = t(Q), t(L), t(U), t(Px), t(Ax)
    x = torch.empty((B, s.n), dtype=torch.float64, device=dev); y = torch.empty((B, s.m), dtype=torch.float64, device=dev); rec = torch.zeros((B, 12), dtype=torch.float64, device=dev)
    s._solver.hip_batch_solve_device(B, Qd.data_ptr(), Ld.data_ptr(), Ud.data_ptr(), x.data_ptr(), y.data_ptr(), rec.data_ptr(), warm=False,
                                     stream=torch.cuda.current_stream(dev).cuda_stream, Px_ptr=Pd.data_ptr(), Ax_ptr=Ad.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(x.cpu().numpy(), xh) and np.array_equal(rec.cpu().numpy()[:, 1], rh[:, 1])


def test_nn_layer_per_element_matrices_take_one_launch():
    """osqp_amd.nn.torch.OSQP forward with 2-D P_val / A_val (nn/torch.py:184-217): ONE batched launch, not an update() + solve() loop"""
    import torch
    from osqp_amd.nn.torch import OSQP as OSQPLayer
    B = 12
    P, Pt, q, A, L, U, Px, Ax, Q = _perturbed(B, seed=2)
    Pc, Ac = P.tocoo(), A.tocoo()
    layer = OSQPLayer((Pc.row, Pc.col), P.shape, (Ac.row, Ac.col), A.shape, eps_abs=EPS, eps_rel=EPS)
    # the layer takes P_val / A_val in the order of ITS index lists (COO of the full symmetric P / of A): build them per element from the same perturbation
    Pfull = []; Afull = []
    for i in range(B):
        Pi, Ai = _element(Pt, A, Px, Ax, i)
        Pfull.append(np.asarray(Pi[Pc.row, Pc.col]).ravel()); Afull.append(np.asarray(Ai[Ac.row, Ac.col]).ravel())
    x = layer(torch.tensor(np.array(Pfull)), torch.tensor(Q), torch.tensor(np.array(Afull)), torch.tensor(L), torch.tensor(U))
    assert layer.setup_count == 1 and getattr(layer, 'mat_batch_launches', 0) == 1
    for i in (0, 11):
        Pi, Ai = _element(Pt, A, Px, Ax, i)
        xo, _, io = Oracle().setup(Pi, Q[i], Ai, L[i], U[i], **OST).solve()
        assert np.abs(x[i].numpy() - xo).max() <= 1e-7 * (1 + np.abs(xo).max())
    x2 = layer(torch.tensor(np.array(Pfull)), torch.tensor(Q), torch.tensor(np.array(Afull)), torch.tensor(L), torch.tensor(U))      # second forward: same handle
    assert layer.setup_count == 1 and layer.mat_batch_launches == 2 and torch.equal(x, x2)
