#!/usr/bin/env python3
"""bench_batch.py -- BASELINE configs[4]: a batch of 4096 MPC QPs (n=120, m=240) sharded across N MI355X, one batched kernel
launch per GPU (one workgroup per problem), final RCCL all_gather of the per-problem status/objective records.
Same launch contract as bench.py (python bench_batch.py --gpus N --steps K --warmup W, torch.distributed.run for N > 1);
not the headline bench -- the parity cases of this config live in tests/test_gpu_batch.py.  scaling: strong (fixed batch)."""
import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'osqp-python_amd'), os.path.join(ROOT, 'oracle')):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1); ap.add_argument('--steps', type=int, default=5); ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=4096); ap.add_argument('--cpu-sample', type=int, default=48)
    args = ap.parse_args()
    warnings.simplefilter('ignore')
    import numpy as np
    import torch
    import torch.distributed as dist
    import osqp_amd
    import problems
    from osqp_amd import sharded
    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1)); local = int(os.environ.get('LOCAL_RANK', 0))
    assert world == args.gpus
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)
    B = args.batch
    P, q, A, L, U = problems.mpc_batch(B)
    s = osqp_amd.OSQP(algebra='hip')
    s.setup(P, q, A, L[0], U[0], eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=4000, device=local)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        table, x, y, rng = sharded.solve_batch_sharded(s, l=L, u=U, rank=rank, world=world, device=dev if world > 1 else None)
    barrier(); t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        table, x, y, rng = sharded.solve_batch_sharded(s, l=L, u=U, rank=rank, world=world, device=dev if world > 1 else None)
        step_ms.append(1e3 * (time.perf_counter() - ts))
    barrier(); el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device='cuda')
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = float(el.item())
    if rank == 0:
        out = {'metric': 'QPs/sec, batch of %d MPC QPs (n=120, m=240), eps 1e-6' % B, 'value': B * args.steps / el, 'unit': 'QP/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * el / args.steps, 'higher_is_better': True, 'scaling': 'strong',
               'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
               'config': {'workload': 'BASELINE configs[4]: %d MPC QPs, horizon 10, nx=8, nu=4 (problems.mpc_batch); host scaling + H2D/D2H included' % B,
                          'ms_per_step_median': float(sorted(step_ms)[len(step_ms) // 2]), 'kernel_ms_last_step': s._solver.hip_stats()['gpu_solve_ms'],
                          'solved': int((table[:, 1] == 1).sum()), 'admm_iters_total': float(table[:, 2].sum()),
                          'admm_iters_per_s': float(table[:, 2].sum()) * args.steps / el}}
        if args.cpu_sample > 0:
            from oracle import Oracle
            t0 = time.perf_counter(); its = 0
            for i in range(args.cpu_sample):
                xo, yo, io = Oracle().setup(P, q, A, L[i], U[i], eps_abs=1e-6, eps_rel=1e-6, adaptive_rho_interval=50, check_termination=25).solve()
                its += io.iter
            dt = time.perf_counter() - t0
            out['cpu_baseline'] = {'value': args.cpu_sample / dt, 'unit': 'QP/s', 'cores': 1, 'kind': 'port',
                                   'sample': '%d of the %d QPs, oracle direct LDL\' (setup+solve per problem), %d ADMM iterations in %.2f s' % (args.cpu_sample, B, its, dt)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == '__main__':
    main()
