"""Inner-tolerance / policy sweep at config 2 with the round-3 kernels: cold-solve time and iteration counts."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT]
warnings.simplefilter('ignore')
import numpy as np
import osqp_amd, problems
seeds = [12345, 1, 2]
probs = [problems.banded_qp(100000, seed=s) for s in seeds]
def run(tag, pol=None, **kw):
    out = []
    for P, q, A, l, u in probs:
        st = dict(eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, check_termination=25, adaptive_rho_interval=50, verbose=False, warm_starting=False)
        st.update(kw)
        m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, **st)
        if pol: m._solver.set_policy(**pol)
        m.solve()                       # graph capture etc.
        ts = []
        for rep in range(2):
            m.update_settings(rho=0.1)
            t = time.perf_counter(); r = m.solve(); ts.append(time.perf_counter() - t)
        s = m._solver.hip_stats()
        out.append((1e3 * min(ts), r.info.iter, s['pcg_iters_total'] / r.info.iter, r.info.status_val))
    print('%-34s' % tag, ' | '.join('%.1f ms %d it %.2f pcg%s' % (o[0], o[1], o[2], '' if o[3] == 1 else ' !!') for o in out), ' mean %.1f ms' % np.mean([o[0] for o in out]), flush=True)
run('default')
for f in (0.1, 0.2, 0.25, 0.3, 0.4):
    run('cg_tol_fraction=%g' % f, cg_tol_fraction=f)
for w in (0, 5):
    run('rho_window=%d' % w, pol=dict(rho_window=w))
for ct, ari in ((25, 100), (50, 50), (50, 100)):
    run('check=%d ari=%d' % (ct, ari), check_termination=ct, adaptive_rho_interval=ari)
