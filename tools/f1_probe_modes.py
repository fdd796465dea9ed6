"""F1 launch duration by probe mode: 14 = without the scalar fold, 15 = fold paid, fixed scalars, 16 = F launches of the slot kernel itself."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT]
warnings.simplefilter('ignore')
import osqp_amd, problems
P, q, A, l, u = problems.banded_qp(100000)
m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, max_iter=20000, verbose=False)
m.solve()
s = m._solver
for rep in range(2):
    print(os.environ.get('OSQP_HIP_LIBRARY', 'default'), {w: round(0.5e3 * s.hip_time_kernel(w, 200), 3) for w in (14, 15, 16)})
