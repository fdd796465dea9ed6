"""Where a step of sharded.solve_batch_sharded_device goes (one rank): the batch kernel vs the torch glue around it."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'osqp-python_amd'), os.path.join(ROOT, 'oracle')]
warnings.simplefilter('ignore')
import numpy as np, torch
import osqp_amd, problems
from osqp_amd import sharded
B = 4096
P, q, A, L, U = problems.mpc_batch(B)
s = osqp_amd.OSQP(); s.setup(P, q, A, L[0], U[0], eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=4000)
dev = torch.device('cuda', 0)
Ld, Ud = torch.tensor(L, device=dev), torch.tensor(U, device=dev)
def sync(): torch.cuda.synchronize()
for _ in range(2): sharded.solve_batch_sharded_device(s, l=Ld, u=Ud); sync()
t0 = time.perf_counter()
for _ in range(5): table, x, y, span = sharded.solve_batch_sharded_device(s, l=Ld, u=Ud); n = int((table[:, 1] == 1).sum().item())
sync(); print('sharded step %.2f ms' % (1e3 * (time.perf_counter() - t0) / 5))
x = torch.empty((B, 120), dtype=torch.float64, device=dev); y = torch.empty((B, 240), dtype=torch.float64, device=dev); rec = torch.zeros((B, 12), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
for _ in range(2): s._solver.hip_batch_solve_device(B, None, Ld.data_ptr(), Ud.data_ptr(), x.data_ptr(), y.data_ptr(), rec.data_ptr(), warm=False, stream=st); sync()
t0 = time.perf_counter()
for _ in range(5):
    s._solver.hip_batch_solve_device(B, None, Ld.data_ptr(), Ud.data_ptr(), x.data_ptr(), y.data_ptr(), rec.data_ptr(), warm=False, stream=st); sync()
print('entry point alone %.2f ms' % (1e3 * (time.perf_counter() - t0) / 5))
t0 = time.perf_counter()
for _ in range(5):
    s._solver.hip_batch_solve_device(B, None, Ld.data_ptr(), Ud.data_ptr(), x.data_ptr(), y.data_ptr(), rec.data_ptr(), warm=False, stream=st)
    t1 = time.perf_counter(); sync(); 
print('  of which host-side call %.2f ms (last)' % (1e3 * (t1 - t0) / 1) )
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record(); s._solver.hip_batch_solve_device(B, None, Ld.data_ptr(), Ud.data_ptr(), x.data_ptr(), y.data_ptr(), rec.data_ptr(), warm=False, stream=st); ev1.record(); sync()
print('events around the call %.2f ms' % ev0.elapsed_time(ev1))
