"""Batch path, device-resident arrays: ms per batch of the workgroup-per-problem kernel (batch_wave = -1) and the wave-per-problem kernel (batch_wave = 1) over
batch sizes; for each: the first call of a warm handle without launch-order history, and the steady call (launch order from the previous call).
usage: python tools/batch_size_sweep.py [sizes ...]"""
import sys, time
sys.path.insert(0, 'osqp-python_amd'); sys.path.insert(0, '.')
import numpy as np, torch
import osqp_amd, problems

sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 4096, 8192]
dev = torch.device('cuda', 0)
for B in sizes:
    P, q, A, L, U = problems.mpc_batch(B)
    row = []
    for w in (-1, 1):
        s = osqp_amd.OSQP(); s.setup(P, q, A, L[0], U[0], eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=4000)
        s._solver.set_policy(batch_wave=w)
        Ld, Ud = torch.tensor(L, device=dev), torch.tensor(U, device=dev)
        xd = torch.empty((B, 120), dtype=torch.float64, device=dev); yd = torch.empty((B, 240), dtype=torch.float64, device=dev)
        rd = torch.empty((B, s._solver.BATCH_REC), dtype=torch.float64, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        def step():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            s._solver.hip_batch_solve_device(B, None, Ld.data_ptr(), Ud.data_ptr(), xd.data_ptr(), yd.data_ptr(), rd.data_ptr(), warm=False, stream=st)
            torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0)
        step()                                   # the handle's first call: preparation
        s._solver.set_policy(batch_reorder=0); nohist = min(step() for _ in range(3)); s._solver.set_policy(batch_reorder=1)
        step(); steady = sorted(step() for _ in range(5))[2]
        row.append((nohist, steady, int((rd[:, 0] == 1).sum().item())))
    print('B %5d | workgroup: no history %.3f ms, steady %.3f ms (%d solved) | wave: no history %.3f ms, steady %.3f ms (%d solved)' % (B, *row[0], *row[1]), flush=True)
