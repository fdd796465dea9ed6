"""The per-handle engine policy (include/osqp_hip.h OSQPHipPolicy): defaults, get/set through the C ABI, validation, and -- on the
GPU -- that policy fields reach the engine (kernel forms, device-driven boundaries) without changing what is computed."""
import ctypes
import os
import warnings

import numpy as np
import pytest

import osqp_amd
import problems
from backend_param import BACKENDS, engine

warnings.simplefilter('ignore')
S = osqp_amd.SolverStatus


@pytest.mark.parametrize('backend', BACKENDS)
def test_policy_defaults_roundtrip_and_validation(backend):
    with engine(backend) as h:
        from osqp_amd import _lib
        p = _lib.PolicyStruct()
        h.osqp_hip_default_policy(ctypes.byref(p))
        assert (p.graph, p.slots, p.pcg_fused, p.f1, p.window, p.woodbury, p.woodbury_direct, p.woodbury_large, p.device_driven, p.small_direct, p.batch_reorder, p.batch_variant) == (1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0)
        assert (p.extrap, p.rho_eq_factor, p.rho_window, p.rho_window_tol, p.rho_persist, p.rho_tol_exp) == (0.9, 0.0, 10, 0.1, 1, 0.5)
        assert (p.budget_sigma, p.cg_escalate, p.stall, p.finish_pairs) == (3.0, 1, 1, 12)
        P, q, A, l, u = problems.banded_qp(400, window=30, seed=4)
        m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, eps_abs=1e-6, eps_rel=1e-6)
        pol = m._solver.get_policy()
        assert pol['rho_window'] == 10 and pol['device_driven'] == 1
        m._solver.set_policy(rho_window=0, rho_persist=0, rho_tol_exp=1.0, small_direct=0)       # the reference's literal rho rule
        pol = m._solver.get_policy()
        assert (pol['rho_window'], pol['rho_persist'], pol['rho_tol_exp'], pol['small_direct']) == (0, 0, 1.0, 0)
        r = m.solve()
        assert r.info.status_val == S.OSQP_SOLVED
        for bad in (dict(extrap=-1.0), dict(rho_tol_exp=0.0), dict(batch_variant=9), dict(rho_eq_factor=0.5), dict(finish_pairs=0)):
            with pytest.raises(ValueError):
                m._solver.set_policy(**bad)
        with pytest.raises(ValueError):
            m._solver.set_policy(no_such_field=1)


@pytest.mark.parametrize('backend', BACKENDS)
def test_environment_is_read_per_handle_not_per_process(backend, monkeypatch):
    """A handle created after the environment changed sees the change (the policy is not a process-wide static)."""
    with engine(backend):
        P, q, A, l, u = problems.banded_qp(400, window=30, seed=4)
        monkeypatch.setenv('OSQP_HIP_RHO_WINDOW', '0')
        m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False)
        assert m._solver.get_policy()['rho_window'] == 0
        monkeypatch.setenv('OSQP_HIP_RHO_WINDOW', '7')
        m2 = osqp_amd.OSQP(); m2.setup(P, q, A, l, u, verbose=False)
        assert m2._solver.get_policy()['rho_window'] == 7 and m._solver.get_policy()['rho_window'] == 0


@pytest.mark.gpu
def test_device_driven_boundaries_compute_exactly_what_the_host_driven_loop_computes():
    """Chunk boundaries decided on the device (k_decide: policy.h) vs on the host (the same policy.h), graph replay vs eager
    launches: identical iteration counts, PCG counts, rho updates and bitwise-identical x, y -- over repeated solves of a handle
    (rho carried over) -- although the host sizes and times its launch strings differently every run."""
    from osqp_amd import _lib
    assert _lib.handle().osqp_hip_backend() == b'hip-gfx950'
    P, q, A, l, u = problems.banded_qp(20000, window=40)
    ref = None
    for dd, graph in ((0, 1), (1, 1), (1, 0), (1, 1)):
        os.environ['OSQP_HIP_DEVICE_DRIVEN'] = str(dd); os.environ['OSQP_HIP_GRAPH'] = str(graph)
        try:
            m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, check_termination=25, adaptive_rho_interval=50,
                                         verbose=False, warm_starting=False)
            assert m._solver.get_policy()['device_driven'] == dd
            got = []
            for rep in range(3):
                r = m.solve(); s = m._solver.hip_stats()
                assert r.info.status_val == S.OSQP_SOLVED
                got.append((r.info.iter, int(s['pcg_iters_total']), r.info.rho_updates, r.x.copy(), r.y.copy()))
        finally:
            os.environ.pop('OSQP_HIP_DEVICE_DRIVEN'); os.environ.pop('OSQP_HIP_GRAPH')
        if ref is None:
            ref = got
        for a, b in zip(ref, got):
            assert a[:3] == b[:3], (dd, graph, a[:3], b[:3])
            assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])


@pytest.mark.gpu
def test_check_dualgap_on_a_small_problem_takes_the_host_driven_loop():
    """The one-launch kernel has no duality-gap test: with check_dualgap a small QP goes through the multi-kernel loop, which has."""
    P, q, A, l, u = problems.random_qp()
    m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False, eps_abs=1e-6, eps_rel=1e-6, check_dualgap=True)
    r = m.solve()
    assert r.info.status_val == S.OSQP_SOLVED
    assert m._solver.hip_stats()['kernel_launches'] > 1
    assert abs(r.info.duality_gap) < 1e-6 + 1e-6 * max(abs(r.info.obj_val), abs(r.info.dual_obj_val))
    m2 = osqp_amd.OSQP(); m2.setup(P, q, A, l, u, verbose=False, eps_abs=1e-6, eps_rel=1e-6)
    r2 = m2.solve()
    assert m2._solver.hip_stats()['kernel_launches'] == 1
    assert np.abs(r.x - r2.x).max() < 1e-4


@pytest.mark.gpu
def test_scalars_frozen_into_captured_launches_follow_the_handle():
    """Captured launches take the device struct by value: the equality-weight rule (k_set_rho inside the boundary group of a
    device-driven solve) and the extrapolation weight are frozen into every graph.  Changing them on a handle that has already solved
    must drop the graphs (Engine::sync_graph_scalars): the next solve equals, bit for bit, that of a fresh handle built with the value."""
    P, q, A, l, u = problems.banded_qp(20000, window=40)
    st = dict(eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, check_termination=25, adaptive_rho_interval=50, verbose=False, warm_starting=False)
    for field, value in (('rho_eq_factor', 50.0), ('extrap', 0.5)):
        m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, **st)
        r0 = m.solve()                                       # captures the slot strings and the boundary group with the defaults
        assert r0.info.status_val == S.OSQP_SOLVED and r0.info.rho_updates >= 1
        m._solver.set_policy(**{field: value})
        m.update_settings(rho=0.1)
        r1 = m.solve()
        f = osqp_amd.OSQP(); f.setup(P, q, A, l, u, **st)
        f._solver.set_policy(**{field: value})
        r2 = f.solve()
        assert r1.info.status_val == r2.info.status_val == S.OSQP_SOLVED
        assert (r1.info.iter, r1.info.rho_updates) == (r2.info.iter, r2.info.rho_updates), (field, r1.info.iter, r2.info.iter)
        assert np.array_equal(r1.x, r2.x) and np.array_equal(r1.y, r2.y), field
        assert r1.info.iter != r0.info.iter or not np.array_equal(r1.x, r0.x)      # (the field does change the trajectory)
