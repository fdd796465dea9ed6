"""CPU tier: the rules of osqp-python_amd/csrc/policy.h, one at a time (chunk schedule, PCG budget, cap escalation, adaptive-rho rule,
inner tolerance) -- the same text the host driver and the device's k_decide compile, here behind tests/hostsim/policy_probe.cpp.
Reference rules they build on: /root/reference/src/osqppurepy/_osqp.py:880-930 (rho estimate / adapt_rho), :998-1077 (termination);
what is added on the indirect path is described in DESIGN.md sections 2.1 and 4.6."""
import ctypes as C
import math
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'hostsim', 'policy_probe.cpp')
OUT = os.path.join(ROOT, 'tests', '_build', 'libpolicy_probe.so')
DEPS = [SRC, os.path.join(ROOT, 'osqp-python_amd', 'csrc', 'policy.h'), os.path.join(ROOT, 'osqp-python_amd', 'csrc', 'backend.h')]


@pytest.fixture(scope='module')
def lib():
    if not (os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(f) for f in DEPS)):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-I', os.path.join(ROOT, 'include'), '-o', OUT, SRC])
    L = C.CDLL(OUT)
    L.pp_new.restype = C.c_void_p
    L.pp_free.argtypes = [C.c_void_p]
    L.pp_set.argtypes = [C.c_void_p, C.c_char_p, C.c_double]; L.pp_set.restype = C.c_int
    L.pp_get.argtypes = [C.c_void_p, C.c_char_p]; L.pp_get.restype = C.c_double
    L.pp_next_chunk.argtypes = [C.c_void_p]
    L.pp_chunk_tol_abs.argtypes = [C.c_void_p]; L.pp_chunk_tol_abs.restype = C.c_double
    L.pp_next_budget.argtypes = [C.c_void_p] + [C.c_int] * 7; L.pp_next_budget.restype = C.c_int
    L.pp_budget_rule.argtypes = [C.c_void_p] + [C.c_int] * 6
    L.pp_rho_rule.argtypes = [C.c_void_p] + [C.c_double] * 4; L.pp_rho_rule.restype = C.c_int
    L.pp_tol_rule.argtypes = [C.c_void_p, C.c_double]
    L.pp_init_tol.argtypes = [C.c_void_p, C.c_double]
    return L


class Ctl:
    def __init__(self, lib, **kw):
        self.lib, self.p = lib, lib.pp_new()
        self.set(**kw)

    def set(self, **kw):
        for k, v in kw.items():
            assert self.lib.pp_set(self.p, k.encode(), float(v)) == 0, k

    def __getitem__(self, k):
        v = self.lib.pp_get(self.p, k.encode())
        assert not math.isnan(v), k
        return v

    def __del__(self):
        self.lib.pp_free(self.p)


def _defaults(lib, **kw):
    st = dict(ct=25, ari=50, max_iter=4000, tightW=10, tightF=0.1, has_quad=1, persist=1, esc_on=1, stall_on=1, cap=50, cap_max=1024, budget0=50, budget1=50,
              budget_min=2, budget_sigma=3.0, budget_tolerate=0.0, budget_slack=0, tol_exp=0.5, cg_tol_fraction=0.15, cg_tol_reduction=10, rho_tolerance=5.0, rho_bar=0.1)
    st.update(kw)
    return Ctl(lib, **st)


def test_chunk_schedule_checks_adaptation_points_and_tight_windows(lib):
    """check_termination 25, adaptation every 50, a tight PCG window over the 10 iterations before each adaptation point:
    chunks 0..25 (kind 0), 25..40 (kind 1), 40..50 (tight, kind 2, ends at a check), then the same again."""
    c = _defaults(lib, budget0=7)
    seen = []
    for _ in range(6):
        lib.pp_next_chunk(c.p)
        seen.append((int(c['iter']), int(c['ch_next']), int(c['ch_kind']), int(c['ch_tight']), int(c['ch_at_check'])))
        c.set(iter=c['ch_next'])
    assert seen == [(0, 25, 0, 0, 1), (25, 40, 1, 0, 0), (40, 50, 2, 1, 1), (50, 75, 0, 0, 1), (75, 90, 1, 0, 0), (90, 100, 2, 1, 1)]
    assert c['tight_seen'] == 1 and c['budget1'] == min(50, 3 * 7 + 2)        # the first tight window starts from 3 b + 2
    # the tight window runs its PCG tightF times tighter (never below the absolute floor)
    c.set(tol_abs=1e-4, ch_tight=1); assert lib.pp_chunk_tol_abs(c.p) == pytest.approx(1e-5)
    c.set(tol_abs=1e-13); assert lib.pp_chunk_tol_abs(c.p) == pytest.approx(1e-13)
    c.set(ch_tight=0, tol_abs=1e-4); assert lib.pp_chunk_tol_abs(c.p) == pytest.approx(1e-4)


def test_chunks_stop_at_max_iter_and_without_a_window(lib):
    c = _defaults(lib, max_iter=60, tightW=0)
    ends = []
    for _ in range(3):
        lib.pp_next_chunk(c.p); ends.append((int(c['ch_next']), int(c['ch_at_check']))); c.set(iter=c['ch_next'])
    assert ends == [(25, 1), (50, 1), (60, 1)]


def test_budget_is_mean_plus_three_sigma_of_the_last_chunk(lib):
    c = _defaults(lib)
    # 25 solves: 20 took 4 iterations, 5 took 6  -> mean 4.4, sigma 0.8 -> ceil(4.4 + 2.4) = 7, but never above max = 6
    n, s, ss = 25, 20 * 4 + 5 * 6, 20 * 16 + 5 * 36
    assert lib.pp_next_budget(c.p, 10, s, ss, n, 6, 0, 0) == 6
    assert lib.pp_next_budget(c.p, 10, s, ss, n, 9, 0, 0) == 7
    assert lib.pp_next_budget(c.p, 10, 25, 25, 25, 1, 0, 0) == 2                     # never below budget_min
    c.set(budget_min=1); assert lib.pp_next_budget(c.p, 10, 25, 25, 25, 1, 0, 0) == 1   # (the Woodbury forms: one iteration per solve)
    c.set(budget_slack=2); assert lib.pp_next_budget(c.p, 10, s, ss, n, 9, 0, 0) == 9


def test_budget_grows_when_solves_run_out(lib):
    c = _defaults(lib, cap=50)
    assert lib.pp_next_budget(c.p, 5, 125, 625, 25, 5, 2, 0) == 6          # a few unconverged solves: one more iteration
    assert lib.pp_next_budget(c.p, 5, 125, 625, 25, 5, 10, 0) == 10        # more than a quarter: doubled
    assert lib.pp_next_budget(c.p, 1, 25, 25, 25, 1, 10, 0) == 3           # (at least + 2)
    assert lib.pp_next_budget(c.p, 40, 1000, 40000, 25, 40, 10, 0) == 50   # never above the cap
    c.set(budget_tolerate=0.2)
    assert lib.pp_next_budget(c.p, 5, 125, 625, 25, 5, 4, 0) == 5          # tolerated share of unconverged solves: the statistics rule applies


def test_cap_escalates_only_for_stagnating_solves_at_the_cap(lib):
    c = _defaults(lib, cap=50, budget0=50, ch_tight=0)
    lib.pp_budget_rule(c.p, 1250, 62500, 25, 50, 25, 5)            # all ran out, few of them stagnating: no escalation
    assert c['cap'] == 50 and c['escalations'] == 0
    c.set(budget0=50)
    lib.pp_budget_rule(c.p, 1250, 62500, 25, 50, 25, 13)           # most reduced their residual by less than 10x: the cap doubles
    assert c['cap'] == 100 and c['escalations'] == 1 and c['budget0'] == 100
    c.set(budget0=20)
    lib.pp_budget_rule(c.p, 500, 10000, 25, 20, 25, 25)            # below the cap: the budget grows first
    assert c['cap'] == 100 and c['budget0'] == 40
    c.set(cap=1024, budget0=1024)
    lib.pp_budget_rule(c.p, 25600, 26214400, 25, 1024, 25, 25)     # cap_max is the end
    assert c['cap'] == 1024
    c.set(esc_on=0, cap=50, budget0=50)
    lib.pp_budget_rule(c.p, 1250, 62500, 25, 50, 25, 25)
    assert c['cap'] == 50


def _estimate(rho, factor):
    """residuals that make the reference's estimate rho * sqrt(pri / dua) = factor * rho (normalisations 1)"""
    return (factor * factor * 1e-3, 1.0, 1e-3, 1.0)


def test_rho_rule_spends_the_tolerance_on_a_square_root_scale(lib):
    c = _defaults(lib, rho_bar=0.1, rho_tolerance=5.0)
    assert lib.pp_rho_rule(c.p, *_estimate(0.1, 2.0)) == 0 and c['rho_bar'] == pytest.approx(0.1)        # 2.0 < sqrt(5) = 2.24
    assert c['rho_estimate'] == pytest.approx(0.2, rel=1e-6)
    c.set(last_side=0)
    assert lib.pp_rho_rule(c.p, *_estimate(0.1, 2.5)) == 1 and c['rho_bar'] == pytest.approx(0.25, rel=1e-6) and c['rho_updates'] == 1
    assert lib.pp_rho_rule(c.p, *_estimate(0.25, 1 / 2.5)) == 1 and c['rho_bar'] == pytest.approx(0.1, rel=1e-6)
    lp = _defaults(lib, rho_bar=0.1, has_quad=0)                  # LPs keep the reference's literal factor
    assert lib.pp_rho_rule(lp.p, *_estimate(0.1, 4.0)) == 0
    lp.set(last_side=0)
    assert lib.pp_rho_rule(lp.p, *_estimate(0.1, 5.5)) == 1


def test_rho_rule_applies_a_persistent_one_sided_estimate(lib):
    c = _defaults(lib, rho_bar=0.1)
    f = 1.8                                                         # between sqrt(2.24) = 1.5 and 2.24: not big, but on one side
    assert lib.pp_rho_rule(c.p, *_estimate(0.1, f)) == 0 and c['last_side'] == 1
    assert lib.pp_rho_rule(c.p, *_estimate(0.1, f)) == 1 and c['rho_bar'] == pytest.approx(0.18, rel=1e-6) and c['last_side'] == 0
    c = _defaults(lib, rho_bar=0.1)
    assert lib.pp_rho_rule(c.p, *_estimate(0.1, f)) == 0
    assert lib.pp_rho_rule(c.p, *_estimate(0.1, 1 / f)) == 0 and c['last_side'] == -1      # the side changed: no update
    assert lib.pp_rho_rule(c.p, *_estimate(0.1, 1.2)) == 0 and c['last_side'] == 0         # inside the band: the evidence is dropped
    c = _defaults(lib, rho_bar=0.1, persist=0)
    assert lib.pp_rho_rule(c.p, *_estimate(0.1, f)) == 0 and lib.pp_rho_rule(c.p, *_estimate(0.1, f)) == 0
    hi = _defaults(lib, rho_bar=1e5)                                # the estimate is clamped to [1e-6, 1e6] (_osqp.py:25-26)
    assert lib.pp_rho_rule(hi.p, *_estimate(1e5, 100.0)) == 1 and hi['rho_bar'] == pytest.approx(1e6)


def test_inner_tolerance_follows_the_dual_residual_and_never_loosens(lib):
    c = _defaults(lib)
    lib.pp_init_tol(c.p, 2.0)
    assert c['tol_abs'] == pytest.approx(0.3) and c['tol_rel'] == pytest.approx(1e-14)
    lib.pp_tol_rule(c.p, 1.0); assert c['tol_abs'] == pytest.approx(0.15)
    lib.pp_tol_rule(c.p, 4.0); assert c['tol_abs'] == pytest.approx(0.15)           # a worse residual does not loosen it
    lib.pp_tol_rule(c.p, 1e-3); assert c['tol_abs'] == pytest.approx(1.5e-4)
    lib.pp_tol_rule(c.p, 1e-20); assert c['tol_abs'] == pytest.approx(1e-13)        # absolute floor
    z = _defaults(lib)
    lib.pp_init_tol(z.p, 0.0)                                                        # dual-feasible start (q = 0): relative tolerance
    assert z['tol_rel'] == pytest.approx(0.1) and z['tol_abs'] == pytest.approx(1e-13)


def test_inner_tolerance_drops_while_an_unbounded_problem_runs_away(lib):
    c = _defaults(lib)
    lib.pp_init_tol(c.p, 1.0)
    c.set(obj_val=-10.0); lib.pp_tol_rule(c.p, 1.0)
    assert c['stall'] == 1.0
    c.set(obj_val=-20.0); lib.pp_tol_rule(c.p, 1.0)                 # no progress of the dual residual, |objective| growing: first strike
    assert c['stall'] == 1.0 and c['stalled_checks'] == 1
    c.set(obj_val=-40.0); lib.pp_tol_rule(c.p, 1.0)                 # second: 10x tighter
    assert c['stall'] == pytest.approx(0.1) and c['tol_abs'] == pytest.approx(0.015)
    c.set(obj_val=-80.0); lib.pp_tol_rule(c.p, 1.0)
    assert c['stall'] == pytest.approx(0.01)
    c.set(obj_val=-80.0); lib.pp_tol_rule(c.p, 0.5)                 # the dual residual improves: back up by 10x per check
    assert c['stall'] == pytest.approx(0.1) and c['stalled_checks'] == 0
    off = _defaults(lib, stall_on=0)
    lib.pp_init_tol(off.p, 1.0)
    for o in (-10.0, -20.0, -40.0, -80.0):
        off.set(obj_val=o); lib.pp_tol_rule(off.p, 1.0)
    assert off['stall'] == 1.0
