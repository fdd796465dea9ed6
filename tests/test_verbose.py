"""The default-settings path: `verbose` is ON by default (osqp_set_default_settings, as /root/reference/src/bindings.cpp.in:411 exposes it) and must change
nothing about how a solve runs -- same driver (device-driven chunks on the GPU, the one-launch path for small QPs), same iterates bit for bit -- while the
text goes where the reference's goes: through the handle's print function to sys.stdout (the reference: c_print = PySys_WriteStdout under the GIL,
/root/reference/cmake/printing.h:2-7), in the reference's layout (/root/reference/src/osqppurepy/_osqp.py:564-613 header, :960-978 table lines every 200
iterations + the last, :1079-1096 footer)."""
import ctypes as C
import re
import threading
import warnings

import numpy as np
import pytest

import osqp_amd
import problems
from backend_param import BACKENDS, engine

warnings.simplefilter('ignore')

LINE = re.compile(r'^\s*(\d+)\s+(-?\d\.\d{4}e[+-]\d\d)\s+(\d\.\d\de[+-]\d\d)\s+(\d\.\d\de[+-]\d\d)\s+(\d\.\d\de[+-]\d\d)\s+(\d\.\d\de[+-]\d\d)s$')


def _check_text(out, r, n, m):
    lines = out.splitlines()
    assert lines[0].startswith('-----') and 'Operator Splitting QP Solver' in lines[1] and lines[3].startswith('-----')
    assert 'problem:  variables n = %d, constraints m = %d' % (n, m) in out
    assert re.search(r'^          nnz\(P\) \+ nnz\(A\) = \d+$', out, re.M)
    assert re.search(r'^settings: linear system solver = indirect', out, re.M)
    assert re.search(r'^          eps_abs = \d\.\d\de-\d\d, eps_rel = \d\.\d\de-\d\d,$', out, re.M)
    assert re.search(r'^          rho = \d\.\d\de[+-]\d\d \(adaptive\)$', out, re.M)
    assert re.search(r'^          sigma = 1\.00e-06, alpha = 1\.60, max_iter = \d+$', out, re.M)
    assert re.search(r'^          scaling: on, scaled_termination: off$', out, re.M)
    assert re.search(r'^          warm_start: on, polish: off$', out, re.M)
    assert 'iter   objective    pri res    dua res    rho       time' in lines
    table = [LINE.match(ln) for ln in lines[lines.index('iter   objective    pri res    dua res    rho       time') + 1:]]
    table = [t for t in table if t]
    its = [int(t.group(1)) for t in table]
    assert its and its[-1] == r.info.iter                                  # the last iteration is always printed (_osqp.py:1259-1261)
    assert all(i % 200 == 0 for i in its[:-1]) and its == sorted(set(its))  # ... the others every PRINT_INTERVAL = 200 (:32, :1230)
    assert abs(float(table[-1].group(2)) - r.info.obj_val) <= 1e-4 * (1 + abs(r.info.obj_val))
    tail = out[out.rindex('status:'):]
    assert re.match(r'status:               solved\nnumber of iterations: %d\noptimal objective:    -?\d+\.\d{4}\nrun time:             \d\.\d\de[+-]\d\ds\noptimal rho estimate: \d\.\d\de[+-]\d\d\n\n$' % r.info.iter, tail)
    return its


@pytest.mark.parametrize('backend', BACKENDS)
def test_default_settings_print_and_change_nothing(backend, capsys):
    P, q, A, l, u = problems.random_qp()
    with engine(backend):
        quiet = osqp_amd.OSQP(); quiet.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False)
        rq = quiet.solve()
        capsys.readouterr()
        loud = osqp_amd.OSQP(); loud.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6)          # NO verbose argument: the default
        assert loud.settings.verbose
        rl = loud.solve()
        out = capsys.readouterr().out
        sq, sl = quiet._solver.hip_stats(), loud._solver.hip_stats()
    assert rq.info.status_val == rl.info.status_val == 1 and rq.info.iter == rl.info.iter
    assert np.array_equal(rq.x, rl.x) and np.array_equal(rq.y, rl.y)       # bit for bit
    assert sq['pcg_iters_total'] == sl['pcg_iters_total'] and sq['pcg_fused'] == sl['pcg_fused']      # the same path (launch COUNTS depend on how the host timed its strings)
    _check_text(out, rl, len(q), len(l))
    assert capsys.readouterr().out == ''


@pytest.mark.gpu
@pytest.mark.parametrize('gen,kw', [(problems.banded_qp, dict(n=20000, window=200)), (problems.portfolio_qp, dict(na=2000, k=20))])
def test_verbose_keeps_the_device_driven_driver(gen, kw, capsys):
    """A solve long enough for table lines at 200, 400, ...: they come from the state block's log (policy.h Ctl::log), whether the host reads
    them while the chunks run or at the end; launch counts and iterates equal the quiet solve's."""
    P, q, A, l, u = gen(**kw)
    st = dict(eps_abs=1e-7, eps_rel=1e-7, max_iter=20000, adaptive_rho_interval=50, check_termination=25)
    quiet = osqp_amd.OSQP(); quiet.setup(P, q, A, l, u, verbose=False, **st)
    rq = quiet.solve(); sq = quiet._solver.hip_stats()
    capsys.readouterr()
    loud = osqp_amd.OSQP(); loud.setup(P, q, A, l, u, **st)
    rl = loud.solve(); sl = loud._solver.hip_stats()
    out = capsys.readouterr().out
    assert rq.info.status_val == rl.info.status_val == 1 and rq.info.iter == rl.info.iter and rl.info.iter > 200
    assert np.array_equal(rq.x, rl.x) and np.array_equal(rq.y, rl.y)
    assert sl['pcg_iters_total'] == sq['pcg_iters_total']
    its = _check_text(out, rl, len(q), len(l))
    assert its[:-1] == list(range(200, rl.info.iter, 200))[:len(its) - 1] and len(its) >= 2
    # a second solve of the loud handle prints the table and the footer again, not the header
    loud.update_settings(rho=0.1); loud.warm_start(x=np.zeros(len(q)), y=np.zeros(len(l)))
    r2 = loud.solve()
    out2 = capsys.readouterr().out
    assert 'Operator Splitting' not in out2 and out2.count('status:') == 1 and r2.info.status_val == 1


@pytest.mark.parametrize('backend', BACKENDS)
def test_per_handle_print_function_and_threads(backend, capsys):
    """osqp_hip_set_print: one handle's text goes to ITS function (here: collected per handle while two handles solve on two threads, each call
    running with the GIL released); the other handles keep the default (sys.stdout)."""
    from osqp_amd import _lib
    P, q, A, l, u = problems.random_qp()
    with engine(backend) as h:
        got = {0: [], 1: []}
        cbs = [_lib.PRINT_FN(lambda text, _u, k=k: got[k].append(text.decode())) for k in (0, 1)]
        solvers = []
        for k in (0, 1):
            s = osqp_amd.OSQP(); s.setup(P, q, A, l, u, eps_abs=1e-5, eps_rel=1e-5)
            assert h.osqp_hip_set_print(s._solver._p, C.cast(cbs[k], C.c_void_p), None) == 0
            solvers.append(s)
        capsys.readouterr()                                                  # (the two setup headers went to the default)
        res = [None, None]
        th = [threading.Thread(target=lambda k=k: res.__setitem__(k, solvers[k].solve())) for k in (0, 1)]
        [t.start() for t in th]; [t.join() for t in th]
        assert capsys.readouterr().out == ''                                 # nothing of the two solves reached sys.stdout
        for k in (0, 1):
            text = ''.join(got[k])
            assert res[k].info.status_val == 1 and text.count('status:               solved') == 1 and 'number of iterations: %d' % res[k].info.iter in text
        assert np.array_equal(res[0].x, res[1].x)
        assert h.osqp_hip_set_print(solvers[0]._solver._p, None, None) == 0   # back to stdout (fputs): nothing to assert beyond the return code
