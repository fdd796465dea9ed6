"""Run under torch.distributed.run with ONE rank (a test box has one GPU): the device-resident sharded batch path
(osqp_amd.sharded.solve_batch_sharded_device + gather_rows_device) with the process group initialised on RCCL, so that both
all_gathers run on device tensors; then the two shares a 2-rank job would solve, one after the other, against the full batch.
Prints one JSON line."""
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), ROOT]
warnings.simplefilter('ignore')
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import osqp_amd  # noqa: E402
from osqp_amd import sharded  # noqa: E402
import problems  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
B = 37
P, q, A, L, U = problems.mpc_batch(B, seed=11)
s = osqp_amd.OSQP(); s.setup(P, q, A, L[0], U[0], verbose=False, eps_abs=1e-6, eps_rel=1e-6, max_iter=20000)
dev = torch.device('cuda', 0)
Ld, Ud = torch.tensor(L, device=dev), torch.tensor(U, device=dev)
cpu_calls = []
orig_cpu = torch.Tensor.cpu
torch.Tensor.cpu = lambda self, *a, **k: (cpu_calls.append(tuple(self.shape)), orig_cpu(self, *a, **k))[1]      # count host copies
table, xl, yl, (lo, hi) = sharded.solve_batch_sharded_device(s, l=Ld, u=Ud, rank=0, world=1)
xg = sharded.gather_rows_device(xl, B)
torch.cuda.synchronize()
big_host_copies = [sh for sh in cpu_calls if int(np.prod(sh)) >= B * 100]
torch.Tensor.cpu = orig_cpu
xh, yh, rech = s._solver.hip_batch_solve(l=L, u=U)
# what ranks 0 and 1 of a two-rank job would each do (no collective: the shares are compared with the full batch)
parts = []
for r in range(2):
    dist_init = dist.is_initialized
    dist.is_initialized = lambda: False
    try:
        recs, x2, y2, (a, b) = sharded.solve_batch_sharded_device(s, l=Ld, u=Ud, rank=r, world=2)
    finally:
        dist.is_initialized = dist_init
    parts.append((a, b, x2.cpu().numpy(), recs.cpu().numpy()))
x2 = np.concatenate([p[2] for p in parts])
out = {'solved': int((table[:, 1] == 1).sum().item()), 'B': B, 'table_is_cuda': bool(table.is_cuda), 'x_is_cuda': bool(xg.is_cuda),
       'max_dx_vs_host_path': float(np.abs(xg.cpu().numpy() - xh).max()), 'max_dx_two_shares': float(np.abs(x2 - xh).max()),
       'shares': [(p[0], p[1]) for p in parts], 'indices_ok': bool(np.array_equal(np.concatenate([p[3][:, 0] for p in parts]), np.arange(B))),
       'big_host_copies': big_host_copies}
print(json.dumps(out))
dist.barrier()
dist.destroy_process_group()
