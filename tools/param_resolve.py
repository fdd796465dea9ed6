"""Parametric re-solve latency on one MI355X (the MPC-style use of update() + warm start):
setup once, then repeatedly perturb q / the bounds / the matrix values and re-solve from the previous solution.

    python tools/param_resolve.py [n]"""
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT]
warnings.simplefilter('ignore')
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
import osqp_amd  # noqa: E402
import problems  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
P, q, A, l, u = problems.banded_qp(n)
rng = np.random.default_rng(0)
m = osqp_amd.OSQP()
t = time.perf_counter(); m.setup(P, q, A, l, u, verbose=False, eps_abs=1e-6, eps_rel=1e-6); t_setup = time.perf_counter() - t
t = time.perf_counter(); r = m.solve(); t_cold = time.perf_counter() - t
print('setup %.1f ms; cold solve %.1f ms, %d iterations, %s' % (1e3 * t_setup, 1e3 * t_cold, r.info.iter, r.info.status))
Pt = sp.triu(P, format='csc')
import torch  # noqa: E402
dev = torch.device('cuda:0')
stream = torch.cuda.current_stream(dev).cuda_stream
for what in ('q', 'bounds', 'q (device pointer)', 'bounds (device pointer)', 'matrices'):
    tu, ts, its = [], [], []
    for rep in range(7):
        # the new data is prepared OUTSIDE the timed region (the clock measures the update call, not numpy's random numbers)
        q2 = q * (1 + 0.01 * rng.standard_normal(len(q)))
        d = 0.01 * rng.random(len(l)); l2, u2 = l - d, u + d
        Px2, Ax2 = Pt.data * (1 + 0.01 * rng.random(Pt.nnz)), A.data * (1 + 0.01 * rng.standard_normal(A.nnz))
        tq, tl, tu_ = torch.tensor(q2, device=dev), torch.tensor(l2, device=dev), torch.tensor(u2, device=dev)
        torch.cuda.synchronize()
        t = time.perf_counter()
        if what == 'q':
            m.update(q=q2)
        elif what == 'bounds':
            m.update(l=l2, u=u2)
        elif what == 'q (device pointer)':
            assert m._solver.hip_update_data_vec_device(tq.data_ptr(), None, None, stream) == 0
        elif what == 'bounds (device pointer)':
            assert m._solver.hip_update_data_vec_device(None, tl.data_ptr(), tu_.data_ptr(), stream) == 0
        else:
            m.update(Px=Px2, Ax=Ax2)
        tu.append(time.perf_counter() - t)
        t = time.perf_counter(); r = m.solve(); ts.append(time.perf_counter() - t); its.append(r.info.iter)
        assert r.info.status_val == 1, r.info.status
    print('1%% change of %-24s update call %.3f ms, warm re-solve %.1f ms (%d iterations)' % (what + ':', 1e3 * np.median(tu), 1e3 * np.median(ts), int(np.median(its))))
