"""GPU tier: the batched small-QP kernel (osqp_hip_batch_solve; BASELINE configs[4] = MPC QPs n=120, m=240) against the
oracle and against the large-problem engine on the same problems."""
import time
import warnings

import numpy as np
import pytest

import osqp_amd
import problems
from oracle import Oracle, SOLVED, PRIMAL_INFEASIBLE

pytestmark = pytest.mark.gpu
warnings.simplefilter('ignore')
EPS = 1e-6
ST = dict(eps_abs=EPS, eps_rel=EPS, verbose=False, max_iter=4000)


def base_solver(P, q, A, l, u, **kw):
    s = osqp_amd.OSQP()
    st = dict(ST); st.update(kw)
    s.setup(P, q, A, l, u, **st)
    return s


def test_batch_matches_oracle_and_single_engine():
    B = 64
    P, q, A, L, U = problems.mpc_batch(B)
    s = base_solver(P, q, A, L[0], U[0])
    x, y, rec = s._solver.hip_batch_solve(l=L, u=U)
    assert (rec[:, 0] == 1).all(), rec[:, 0]
    for i in (0, 5, 17, 63):
        xo, yo, io = Oracle().setup(P, q, A, L[i], U[i], eps_abs=1e-9, eps_rel=1e-9, adaptive_rho_interval=50, max_iter=100000).solve()
        assert io.status_val == SOLVED
        assert abs(rec[i, 2] - io.obj_val) <= 1e-5 * (1 + abs(io.obj_val))
        assert np.abs(x[i] - xo).max() <= 2e-5 * (1 + np.abs(xo).max())          # (eps = 1e-6 iterate against the 1e-9 solution)
        assert np.abs(y[i] - yo).max() <= 1e-4 * (1 + np.abs(yo).max())
        # ... and against the oracle run with the SAME settings: the direct variant is the oracle's algorithm -- equal iteration counts, 1e-7
        xs, ys, is_ = Oracle().setup(P, q, A, L[i], U[i], eps_abs=EPS, eps_rel=EPS, max_iter=4000, adaptive_rho_interval=50, check_termination=25).solve()
        assert is_.status_val == SOLVED and int(rec[i, 1]) == is_.iter
        assert np.abs(x[i] - xs).max() <= 1e-7 * (1 + np.abs(xs).max()) and np.abs(y[i] - ys).max() <= 1e-7 * (1 + np.abs(ys).max())
        k = problems.kkt_certificate(P, q, A, L[i], U[i], x[i], y[i])
        assert k['pri'] <= 2 * EPS * (1 + np.abs(A @ x[i]).max()) and k['dua'] <= 2 * EPS * (1 + np.abs(A.T @ y[i]).max() + np.abs(P @ x[i]).max())
    # same problems through the single-QP engine (update + cold-started solve per problem: nn/torch.py:136-157)
    s.update_settings(warm_starting=False)
    for i in range(0, B, 9):
        s.update(l=L[i], u=U[i])
        r = s.solve()
        assert r.info.status_val == 1
        assert abs(r.info.obj_val - rec[i, 2]) <= 1e-5 * (1 + abs(rec[i, 2]))
        assert np.abs(r.x - x[i]).max() <= 1e-4 * (1 + np.abs(x[i]).max())


def test_batch_with_q_updates_infeasible_member_and_warm_start():
    B = 16
    P, q, A, L, U = problems.mpc_batch(B, seed=11)
    rng = np.random.default_rng(3)
    Q = 0.05 * rng.standard_normal((B, P.shape[0]))
    L = L.copy(); U = U.copy()
    L[3, :8] += 500.0; U[3, :8] += 500.0          # x_1 = Ad x0 + Bd u0 forced far outside |x| <= 5 : primal infeasible
    s = base_solver(P, q, A, L[0], U[0])
    x, y, rec = s._solver.hip_batch_solve(q=Q, l=L, u=U)
    xo, yo, io = Oracle().setup(P, Q[3], A, L[3], U[3], eps_abs=EPS, eps_rel=EPS, adaptive_rho_interval=50, check_termination=25).solve()
    assert io.status_val == PRIMAL_INFEASIBLE and rec[3, 0] == PRIMAL_INFEASIBLE
    cert = y[3] / np.abs(y[3]).max()
    assert np.abs(A.T @ cert).max() < 1e-3 and U[3] @ np.maximum(cert, 0) + L[3] @ np.minimum(cert, 0) < 0
    ok = np.arange(B) != 3
    assert (rec[ok, 0] == 1).all()
    for i in (0, 8):
        xo, yo, io = Oracle().setup(P, Q[i], A, L[i], U[i], eps_abs=1e-9, eps_rel=1e-9, adaptive_rho_interval=50, max_iter=100000).solve()
        assert abs(rec[i, 2] - io.obj_val) <= 1e-5 * (1 + abs(io.obj_val))
    # warm start at the solution: converged at the first check
    x2, y2, rec2 = s._solver.hip_batch_solve(q=Q[ok], l=L[ok], u=U[ok], x0=x[ok], y0=y[ok])
    assert (rec2[:, 0] == 1).all() and (rec2[:, 1] <= 25).all()
    assert np.abs(x2 - x[ok]).max() < 5e-5          # (both are eps = 1e-6 solutions of the same QP)


def test_batch_sharded_table_and_throughput():
    from osqp_amd import sharded
    B = 4096
    P, q, A, L, U = problems.mpc_batch(B)
    s = base_solver(P, q, A, L[0], U[0])
    s._solver.hip_batch_solve(l=L[:64], u=U[:64])          # warm-up
    t0 = time.perf_counter()
    table, x, y, (lo, hi) = sharded.solve_batch_sharded(s, l=L, u=U)
    dt = time.perf_counter() - t0
    assert table.shape == (B, len(sharded.RECORD_FIELDS)) and (table[:, 1] == 1).all() and (lo, hi) == (0, B)
    print('batch of %d MPC QPs: %.1f ms (%.0f QPs/s, %.0f ADMM iter/s aggregate)' % (B, dt * 1e3, B / dt, table[:, 2].sum() / dt))
    # every one of the 4096: the KKT certificate recomputed on the host from the unscaled data (termination criterion _osqp.py:728-794)
    AX = x @ A.T.toarray(); PX = x @ P.toarray(); ATY = y @ A.toarray()
    pri = np.abs(AX - np.clip(AX, L, U)).max(axis=1)
    dua = np.abs(PX + q[None, :] + ATY).max(axis=1)
    sp_ = np.maximum(np.abs(AX).max(axis=1), np.abs(np.clip(AX, L, U)).max(axis=1))
    sd = np.maximum(np.maximum(np.abs(PX).max(axis=1), np.abs(ATY).max(axis=1)), np.abs(q).max())
    assert (pri <= 1.01 * (EPS + EPS * sp_)).all() and (dua <= 1.01 * (EPS + EPS * sd)).all(), (pri.max(), dua.max())
    # a random 64 of them against the oracle run with the SAME settings: the default variant solves the reduced KKT system directly (banded
    # LDL' in LDS) with the reference's rho rule, i.e. it is the oracle's algorithm -- equal iteration counts, x and y to 1e-7
    rng = np.random.default_rng(5)
    for i in rng.choice(B, 64, replace=False):
        xo, yo, io = Oracle().setup(P, q, A, L[i], U[i], eps_abs=EPS, eps_rel=EPS, max_iter=4000, adaptive_rho_interval=50, check_termination=25).solve()
        assert io.status_val == SOLVED
        assert int(table[i, 2]) == io.iter, (i, table[i, 2], io.iter)
        assert np.abs(x[i] - xo).max() <= 1e-7 * (1 + np.abs(xo).max()) and np.abs(y[i] - yo).max() <= 1e-7 * (1 + np.abs(yo).max())
        assert abs(table[i, 3] - io.obj_val) <= 1e-9 * (1 + abs(io.obj_val))


def test_nn_module_forward_shared_and_per_element_matrices():
    """osqp_amd.nn.torch.OSQP forward (reference: src/osqp/nn/torch.py:22-230; nn_test.py builds it the same way)."""
    import torch
    from osqp_amd.nn.torch import OSQP as OSQPLayer
    B = 6
    P, q, A, L, U = problems.mpc_batch(B, seed=21)
    Pc, Ac = P.tocoo(), A.tocoo()
    layer = OSQPLayer((Pc.row, Pc.col), P.shape, (Ac.row, Ac.col), A.shape, eps_abs=1e-6, eps_rel=1e-6)
    Pv, Av = torch.tensor(Pc.data), torch.tensor(Ac.data)
    qv = torch.zeros(B, P.shape[0], dtype=torch.float64)
    x1 = layer(Pv, qv, Av, torch.tensor(L), torch.tensor(U))                       # shared matrices -> batched kernel
    x2 = layer(Pv.repeat(B, 1) * 1.0, qv, Av.repeat(B, 1), torch.tensor(L), torch.tensor(U))   # per-element -> update() loop
    assert x1.shape == (B, P.shape[0]) and torch.allclose(x1, x2, atol=2e-4)
    xo, yo, io = Oracle().setup(P, q, A, L[2], U[2], eps_abs=1e-9, eps_rel=1e-9, adaptive_rho_interval=50, max_iter=100000).solve()
    assert np.abs(x1[2].numpy() - xo).max() <= 1e-4 * (1 + np.abs(xo).max())
    with pytest.raises(RuntimeError):                                              # unsolved element raises (nn/torch.py:158-162)
        Lb, Ub = L.copy(), U.copy(); Lb[1, :8] += 500; Ub[1, :8] += 500
        layer(Pv, qv, Av, torch.tensor(Lb), torch.tensor(Ub))


def test_nn_module_forward_zero_copy_device_tensors():
    """q, l, u as torch ROCm tensors: handed to the batch kernel by device pointer on torch's current stream
    (osqp_hip_batch_solve_device), result produced on the device; must equal the host-array path bit for bit (same kernel,
    same inputs), also when the inputs are produced by torch kernels queued just before on the same stream."""
    import torch
    from osqp_amd.nn.torch import OSQP as OSQPLayer
    B = 40
    P, q, A, L, U = problems.mpc_batch(B, seed=8)
    Pc, Ac = P.tocoo(), A.tocoo()
    layer = OSQPLayer((Pc.row, Pc.col), P.shape, (Ac.row, Ac.col), A.shape, eps_abs=1e-6, eps_rel=1e-6)
    Pv, Av = torch.tensor(Pc.data), torch.tensor(Ac.data)
    qh = 0.02 * torch.randn(B, P.shape[0], dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    xh = layer(Pv, qh, Av, torch.tensor(L), torch.tensor(U))
    dev = torch.device('cuda:0')
    # inputs computed ON the device right before the call (ordering on the stream matters)
    qd = (qh.to(dev) * 2.0) / 2.0
    Ld, Ud = torch.tensor(L, device=dev) + 0.0, torch.tensor(U, device=dev) + 0.0
    xd = layer(Pv, qd, Av, Ld, Ud)
    assert xd.is_cuda and xd.shape == xh.shape and xd.dtype == xh.dtype
    assert torch.equal(xd.cpu(), xh)
    assert layer.last_dual.is_cuda and layer.last_dual.shape == (B, A.shape[0])
    with pytest.raises(RuntimeError):
        Lb, Ub = Ld.clone(), Ud.clone(); Lb[1, :8] += 500; Ub[1, :8] += 500
        layer(Pv, qd, Av, Lb, Ub)


def test_nn_module_keeps_one_solver_across_forwards():
    """The reference keeps its solver objects across forward calls and only update()s them (nn/torch.py:113-140).  Here: one
    osqp_setup for any number of same-structure forwards (host and device tensors), a changed P_val / A_val goes through
    update_data_mat (device re-assembly), and every later forward equals a freshly set-up layer's result bit for bit."""
    import torch
    from osqp_amd.nn.torch import OSQP as OSQPLayer
    B = 16
    P, q, A, L, U = problems.mpc_batch(B, seed=5)
    Pc, Ac = P.tocoo(), A.tocoo()
    mk = lambda: OSQPLayer((Pc.row, Pc.col), P.shape, (Ac.row, Ac.col), A.shape, eps_abs=1e-6, eps_rel=1e-6)
    layer = mk()
    Pv, Av = torch.tensor(Pc.data), torch.tensor(Ac.data)
    qv = torch.zeros(B, P.shape[0], dtype=torch.float64)
    t = lambda a: torch.tensor(a)
    layer(Pv, qv, Av, t(L), t(U))
    x2 = layer(Pv, qv, Av, t(L + 0.05), t(U + 0.05))                              # new bounds only
    assert layer.setup_count == 1
    assert torch.equal(x2, mk()(Pv, qv, Av, t(L + 0.05), t(U + 0.05)))
    dev = torch.device('cuda:0')
    x3 = layer(Pv, qv.to(dev), Av, t(L + 0.05).to(dev), t(U + 0.05).to(dev))      # device tensors: same handle (device 0)
    assert layer.setup_count == 1 and torch.equal(x3.cpu(), x2)
    x4 = layer(Pv * 1.5, qv, Av * 0.9, t(L), t(U))                                # new matrix VALUES: update_data_mat, no setup
    assert layer.setup_count == 1
    x4f = mk()(Pv * 1.5, qv, Av * 0.9, t(L), t(U))
    assert torch.allclose(x4, x4f, atol=5e-5)                                     # (scaling D, E, c stays that of the first setup: _osqp.py:1443)
    x5 = layer(Pv, qv, Av, t(L), t(U))                                            # and back
    assert layer.setup_count == 1 and torch.allclose(x5, mk()(Pv, qv, Av, t(L), t(U)), atol=5e-5)


@pytest.mark.parametrize('variant', ['direct', 'direct256', 'w64'])
def test_batch_variants_agree_with_oracle(variant, monkeypatch):
    """Both linear-solve variants of the batch kernel (banded Cholesky in LDS / PCG) on the same MPC batch; the direct one
    follows the reference's direct algorithm (same rho rule, exact solves), so its ADMM iteration counts must match the
    oracle's (checked every 25 iterations on both sides)."""
    monkeypatch.setenv('OSQP_HIP_BATCH_VARIANT', variant)
    B = 24
    P, q, A, L, U = problems.mpc_batch(B, seed=5)
    s = base_solver(P, q, A, L[0], U[0])
    x, y, rec = s._solver.hip_batch_solve(l=L, u=U)
    assert (rec[:, 0] == 1).all(), rec[:, 0]
    assert (rec[:, 7] == 0).all() if variant.startswith('direct') else (rec[:, 7] > 0).all()      # PCG iterations
    for i in (0, 11, 23):
        o = Oracle().setup(P, q, A, L[i], U[i], eps_abs=EPS, eps_rel=EPS, adaptive_rho_interval=s.settings.adaptive_rho_interval or 50,
                           check_termination=25, max_iter=4000)
        xo, yo, io = o.solve()
        assert io.status_val == SOLVED
        assert abs(rec[i, 2] - io.obj_val) <= 1e-5 * (1 + abs(io.obj_val))
        assert np.abs(x[i] - xo).max() <= 2e-4 * (1 + np.abs(xo).max())
        if variant == 'direct':
            print('direct iterations', rec[i, 1], 'oracle', io.iter)


def test_batch_direct_longer_horizon_three_epochs(monkeypatch, variant='direct256'):
    """n = 180 (horizon 15): the substitutions run over three 64-pivot epochs and the factor needs 100 KB of LDS (the
    one-wave variant does not apply: more than 24 matrix entries per lane)."""
    monkeypatch.setenv('OSQP_HIP_BATCH_VARIANT', variant)
    B = 12
    P, q, A, L, U = problems.mpc_batch(B, N=15, seed=2)
    s = base_solver(P, q, A, L[0], U[0])
    x, y, rec = s._solver.hip_batch_solve(l=L, u=U)
    assert (rec[:, 0] == 1).all() and (rec[:, 7] == 0).all(), rec[:, [0, 7]]
    for i in (0, 6, 11):
        xo, yo, io = Oracle().setup(P, q, A, L[i], U[i], eps_abs=1e-9, eps_rel=1e-9, adaptive_rho_interval=50, max_iter=100000).solve()
        assert io.status_val == SOLVED
        assert abs(rec[i, 2] - io.obj_val) <= 1e-5 * (1 + abs(io.obj_val))
        assert np.abs(x[i] - xo).max() <= 1e-4 * (1 + np.abs(xo).max())


def test_launch_order_of_a_repeated_batch_does_not_change_results(monkeypatch):
    """The second call of a batch launches its problems longest-last-time first (Engine::batch_solve): scheduling only -- every
    record and solution is bitwise what index order gives."""
    B = 96
    P, q, A, L, U = problems.mpc_batch(B, seed=21)
    s = osqp_amd.OSQP(); s.setup(P, q, A, L[0], U[0], eps_abs=1e-6, eps_rel=1e-6, verbose=False)
    x1, y1, r1 = s._solver.hip_batch_solve(l=L, u=U)            # index order (no history yet)
    x2, y2, r2 = s._solver.hip_batch_solve(l=L, u=U)            # reordered by r1's iteration counts
    assert len(set(r1[:, 1])) > 3                                 # the iteration counts do differ between problems
    assert np.array_equal(x1, x2) and np.array_equal(y1, y2) and np.array_equal(r1[:, :9], r2[:, :9])
    monkeypatch.setenv('OSQP_HIP_BATCH_REORDER', '0')


def test_device_pointer_batches_reorder_on_the_device_and_keep_their_results():
    """osqp_hip_batch_solve_device: the launch order of a repeated batch comes from a rank kernel over the previous call's iteration
    counts (nothing of it reaches the host); results are bitwise those of the first (index-order) call."""
    import torch
    B = 64
    P, q, A, L, U = problems.mpc_batch(B, seed=31)
    s = osqp_amd.OSQP(); s.setup(P, q, A, L[0], U[0], eps_abs=1e-6, eps_rel=1e-6, verbose=False)
    dev = torch.device('cuda:0')
    Ld, Ud = torch.tensor(L, device=dev), torch.tensor(U, device=dev)
    outs = []
    for _ in range(3):
        x = torch.empty((B, P.shape[0]), dtype=torch.float64, device=dev); y = torch.empty((B, A.shape[0]), dtype=torch.float64, device=dev)
        rec = torch.empty((B, s._solver.BATCH_REC), dtype=torch.float64, device=dev)
        s._solver.hip_batch_solve_device(B, None, Ld.data_ptr(), Ud.data_ptr(), x.data_ptr(), y.data_ptr(), rec.data_ptr(), warm=False,
                                         stream=torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize()
        outs.append((x.cpu().numpy(), y.cpu().numpy(), rec.cpu().numpy()))
    assert (outs[0][2][:, 0] == 1).all()
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2][:, :9], outs[0][2][:, :9])
    xh, yh, rh = s._solver.hip_batch_solve(l=L, u=U)                 # host path afterwards: same results again
    assert np.array_equal(xh, outs[0][0]) and np.array_equal(rh[:, :9], outs[0][2][:, :9])
