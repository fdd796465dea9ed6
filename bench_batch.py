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


_CPU = {}


def _cpu_init(P, q, A, L, U):
    import oracle
    oracle.use_native()                                # -O3 -march=native, compiled on this host
    oracle.lib()                                       # load the library in the worker before the clock starts
    _CPU['data'] = (P, q, A, L, U)


def _cpu_solve(span):
    """One worker's contiguous share of the sample with the reference's batch semantics (/root/reference/src/osqp/nn/torch.py:136-140: persistent solver
    objects, update(l, u) + solve() per element): ONE setup (scaling + LDL' factorisation) per worker, then per problem update of the bounds, rho back to
    the setting, cold start, solve.  (Round 5 ran setup + solve per problem: a fresh factorisation each -- not what the GPU side, which shares one setup
    outside the clock, is compared with.)"""
    import time as _t
    from oracle import Oracle
    P, q, A, L, U = _CPU['data']
    t0 = _t.perf_counter()
    o = _CPU.get('solver')
    if o is None:
        o = _CPU['solver'] = Oracle().setup(P, q, A, L[span[0]], U[span[0]], eps_abs=1e-6, eps_rel=1e-6, adaptive_rho_interval=50, check_termination=25, warm_start=0)
    ts = _t.perf_counter() - t0
    t0 = _t.perf_counter(); its = 0
    for i in range(*span):
        o.update(l=L[i], u=U[i]); o.update_rho(0.1)
        its += o.solve()[2].iter
    return its, _t.perf_counter() - t0, ts


def cpu_batch_baseline(P, q, A, L, U, cores=None):
    """The oracle over ALL host cores (SURVEY 8(d) / BASELINE.md section 3: 'an OpenMP-over-problems mode using all host cores'): the batch split
    into one contiguous share per worker process, setup + solve per problem like the reference's per-element solver objects
    (/root/reference/src/osqp/nn/torch.py:200-217).  Workers load the -march=native build made on this host."""
    import multiprocessing as mp
    B = len(L)
    ncpu = os.cpu_count() or 1
    try:
        ncpu = len(os.sched_getaffinity(0))              # the cores this process may actually use
    except (AttributeError, OSError):
        pass
    cores = max(1, min(int(cores or ncpu), B))
    spans = [((B * w) // cores, (B * (w + 1)) // cores) for w in range(cores)]
    import oracle
    oracle.build_native()                                # compile once, in the parent
    with mp.get_context('spawn').Pool(cores, initializer=_cpu_init, initargs=(P, q, A, L, U)) as pool:      # (spawn: the parent holds a HIP context)
        pool.map(_cpu_solve, [(0, 1)] * cores)                                            # every worker has started, loaded the library and set its solver up (untimed, like the GPU side's setup)
        t0 = time.perf_counter()
        parts = pool.map(_cpu_solve, spans, chunksize=1)
        dt = time.perf_counter() - t0
    its = sum(p[0] for p in parts); busy = max(p[1] for p in parts)
    return {'value': B / dt, 'unit': 'QP/s', 'cores': cores, 'kind': 'port', 'build': 'gcc -O3 -march=native -fno-fast-math, compiled on this host',
            'sample': 'all %d QPs split over %d worker processes (every core this process may use; os.cpu_count() = %d), oracle direct LDL\': ONE setup per worker (outside the '
                      'clock, like the GPU side\'s), then update(l, u) + rho reset + cold solve per problem (the reference\'s batch semantics, nn/torch.py:136-140); '
                      '%d ADMM iterations in %.2f s wall (slowest worker busy %.2f s)' % (B, cores, os.cpu_count() or 1, its, dt, busy)}


def sharded_kernel_only(s, Ld, Ud, rank, world):
    """This rank's share (Ld, Ud hold exactly its rows) as ONE batched launch on torch's current stream, nothing else (for the event-timed kernel figure)."""
    import torch
    lo, hi = 0, Ld.shape[0]
    nb = hi - lo
    x = torch.empty((nb, s.n), dtype=torch.float64, device=Ld.device); y = torch.empty((nb, s.m), dtype=torch.float64, device=Ld.device)
    rec = torch.zeros((max(nb, 1), 12), dtype=torch.float64, device=Ld.device)
    s._solver.hip_batch_solve_device(nb, None, Ld[lo:hi].data_ptr(), Ud[lo:hi].data_ptr(), x.data_ptr(), y.data_ptr(), rec.data_ptr(), warm=False,
                                     stream=torch.cuda.current_stream(Ld.device).cuda_stream)
    return x, y, rec


def measure_sharded_host(B, steps, rank, world, use_dist):
    """The same split through the HOST-array path (osqp_amd.sharded.solve_batch_sharded, all_gather on CPU tensors): what bench.py's test mode on
    the host simulator runs (tests/test_bench_launch.py) -- the sharding arithmetic, the collective and the table are the product's, the
    solves are the simulator's.  Returns the same dict shape as measure_sharded_device."""
    import torch
    import torch.distributed as dist
    import osqp_amd
    import problems
    from osqp_amd import sharded
    P, q, A, L, U = problems.mpc_batch(B, nx=3, nu=2, N=4)
    s = osqp_amd.OSQP(algebra='hip')
    s.setup(P, q, A, L[0], U[0], eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=4000)
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter(); ms_each = []
    for _ in range(steps):
        ts = time.perf_counter()
        table, x, y, span = sharded.solve_batch_sharded(s, l=L, u=U, rank=rank, world=world)
        ms_each.append(round(1e3 * (time.perf_counter() - ts), 3))
    own = torch.tensor([sum(ms_each) / max(steps, 1)], dtype=torch.float64)
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    per_rank_ms = [float(own.item())]
    if use_dist:
        dist.barrier()
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        owns = [torch.empty_like(own) for _ in range(world)]
        dist.all_gather(owns, own)
        per_rank_ms = [float(o.item()) for o in owns]
    el = float(el.item())
    owners = {int(i * world // B) for i in table[:, 0].astype(int)} if world > 1 else {0}
    return {'workload': '%d small MPC QPs, contiguous blocks over %d rank(s), host-array path on the HOST SIMULATOR (test mode)' % (B, world),
            'QP_per_s': B * steps / el, 'ms_per_batch': 1e3 * el / steps, 'ms_each': ms_each, 'steps': steps, 'solved': int((table[:, 1] == 1).sum()), 'records': int(table.shape[0]),
            'n_ranks_seen': len(owners), 'per_rank_ms': [round(v, 3) for v in per_rank_ms], 'scaling': 'strong', 'admm_iters_total': float(table[:, 2].sum()),
            'collective': 'all_gather (%s)' % ('gloo' if use_dist else 'single process: none needed'), '_data': (P, q, A, L, U)}


def measure_sharded_device(B, steps, warmup, rank, world, local, use_dist):
    """BASELINE configs[4] through the path a multi-GPU job runs (osqp_amd.sharded.solve_batch_sharded_device: this rank's contiguous
    share in one batched launch by device pointer, then the job's ONE collective -- an all_gather of the 7-field records over RCCL -- INSIDE
    the timed region).  Called by bench.py on every rank (any --gpus N, N = 1 included) so that the driver's scaling run carries the batch's
    strong-scaling curve next to the replica metric.  Returns the dict bench.py prints as config.batch (rank 0) -- all ranks must call."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import osqp_amd
    import problems
    from osqp_amd import sharded
    dev = torch.device('cuda', local)
    P, q, A, L, U = problems.mpc_batch(B)
    s = osqp_amd.OSQP(algebra='hip')
    s.setup(P, q, A, L[0], U[0], eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=4000, device=local)
    lo, hi = sharded.shard_range(B, rank, world)
    Ld, Ud = torch.tensor(L[lo:hi], device=dev), torch.tensor(U[lo:hi], device=dev)        # every rank uploads ITS OWN row block only

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def step(Lt, Ut):
        tb, x, y, span = sharded.solve_batch_sharded_device(s, l=Lt, u=Ut, rank=rank, world=world, total=B)
        ok = int((tb[:, 1] == 1).sum().item())                                   # (the host reads the gathered table: synchronises the step)
        return tb, ok
    table = None
    # the very first call of the handle: host-side preparation of the spectral form (dense eigen-decomposition, once per handle and matrix epoch), no launch-order history
    barrier(); tf = time.perf_counter(); table, _ = step(Ld, Ud); first_call_ms = 1e3 * (time.perf_counter() - tf)
    # a batch WITHOUT launch-order history (OSQPHipPolicy::batch_reorder = 0: index order), everything else warm: what a batch of new problems costs
    s._solver.set_policy(batch_reorder=0)
    nohist = []
    for _ in range(3):
        barrier(); tf = time.perf_counter(); table, _ = step(Ld, Ud); nohist.append(1e3 * (time.perf_counter() - tf))
    s._solver.set_policy(batch_reorder=1)
    for _ in range(max(warmup, 1)):            # (the SAME body as a timed step: the first use of a torch op loads its kernels -- ~100 ms once)
        table, ncheck = step(Ld, Ud)
    barrier(); t0 = time.perf_counter()
    ms_each = []
    for _ in range(steps):
        ts = time.perf_counter()
        table, ncheck = step(Ld, Ud)
        ms_each.append(round(1e3 * (time.perf_counter() - ts), 3))
    own = torch.tensor([1e3 * sum(ms_each) / max(steps, 1)], dtype=torch.float64, device=dev)      # this rank's own mean step (solve of its share + the gather)
    barrier(); el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    per_rank_ms = [float(own.item())]
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        owns = [torch.empty_like(own) for _ in range(world)]
        dist.all_gather(owns, own)
        per_rank_ms = [float(o.item()) for o in owns]
    el = float(el.item())
    # PERTURBED batches: every step solves NEW problems (x0 redrawn: other dynamics bounds) -- the launch order comes from the previous, different batch
    pert = []; pert_solved = []; pert_other = []
    for k in range(steps):
        _, _, _, Lk, Uk = problems.mpc_batch(B, seed=1000 + k)
        Lk_d, Uk_d = torch.tensor(Lk[lo:hi], device=dev), torch.tensor(Uk[lo:hi], device=dev)
        barrier(); tf = time.perf_counter(); tbk, okk = step(Lk_d, Uk_d); pert.append(1e3 * (time.perf_counter() - tf)); pert_solved.append(okk)
        if okk < B:                                                              # (random x0 can make an MPC QP infeasible: e.g. seed 1004, problem 3251 -- the oracle says primal infeasible too)
            st = tbk[:, 1].cpu().numpy().astype(int)
            pert_other.append({'step': k, 'status_counts': {str(v): int((st == v).sum()) for v in sorted(set(st.tolist())) if v != 1}})
    # the launch alone, this rank's share: HIP events around the batch kernel on its stream (no gather, no host read of the table)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); sharded_kernel_only(s, Ld, Ud, rank, world); e1.record(); e1.synchronize()
    kernel_ms = e0.elapsed_time(e1)
    shares = {}
    if world == 1 and B >= 8:
        # what ONE rank of an N-GPU job would do: a contiguous N-th of the batch in one launch + the read of its table, on this GPU -- the numbers that
        # predict the strong-scaling curve while no multi-GPU node is at hand (the slowest QP's chain bounds a share, not throughput)
        for parts in (2, 4, 8):
            loN, hiN = sharded.shard_range(B, 0, parts)
            tN = []
            for rep in range(steps + 1):
                torch.cuda.synchronize(); ts = time.perf_counter()
                tbN, _, _, _ = sharded.solve_batch_sharded_device(s, l=Ld[loN:hiN], u=Ud[loN:hiN], rank=0, world=1)
                _ = int((tbN[:, 1] == 1).sum().item())
                if rep:
                    tN.append(1e3 * (time.perf_counter() - ts))
            shares[parts] = sorted(tN)[len(tN) // 2]
    share8 = shares.get(8)
    tab = table.cpu().numpy()
    owners = {int(i * world // B) for i in tab[:, 0].astype(int)} if world > 1 else {0}       # ranks whose records arrived (problem i lives on rank i * world // B)
    return {'workload': 'BASELINE configs[4]: %d MPC QPs (n=120, m=240, problems.mpc_batch), eps 1e-6, contiguous blocks over %d rank(s), one batched launch per rank, '
                        'one all_gather of the 7-field records (inside the timed region); bounds resident in HBM, x / y left in HBM' % (B, world),
            'QP_per_s': B * steps / el, 'ms_per_batch': 1e3 * el / steps, 'ms_each': ms_each, 'steps': steps, 'solved': int((tab[:, 1] == 1).sum()), 'records': int(tab.shape[0]),
            'n_ranks_seen': len(owners), 'per_rank_ms': [round(v, 3) for v in per_rank_ms], 'kernel_ms_rank0_share': kernel_ms, 'share_of_8_ms': share8, 'share_of_4_ms': shares.get(4), 'share_of_2_ms': shares.get(2),
            'first_call_ms': first_call_ms, 'first_batch_ms': sorted(nohist)[len(nohist) // 2], 'first_batch_ms_each': [round(v, 3) for v in nohist],
            'first_batch_what': 'warm handle, NO launch-order history (batch_reorder = 0: index order); first_call_ms = the handle\'s very first call, host-side preparation of the spectral form included',
            'perturbed_ms': sorted(pert)[len(pert) // 2], 'perturbed_ms_each': [round(v, 3) for v in pert], 'perturbed_solved': pert_solved, 'perturbed_other_statuses': pert_other,
            'perturbed_what': 'every step a batch of NEW problems (x0 redrawn, seeds 1000 + k); the launch order is the previous, different batch\'s', 'scaling': 'strong', 'admm_iters_total': float(tab[:, 2].sum()),
            'collective': 'all_gather (%s)' % ('RCCL' if use_dist else 'single process: none needed'), '_data': (P, q, A, L, U)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1); ap.add_argument('--steps', type=int, default=5); ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=4096); ap.add_argument('--cpu-sample', type=int, default=48)
    ap.add_argument('--cpu-cores', type=int, default=0, help='worker processes of the CPU baseline (0 = every core this process may use)')
    args = ap.parse_args()
    warnings.simplefilter('ignore')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:      # no launcher around us: start the ranks ourselves (as bench.py does)
        import bench
        sys.exit(bench.self_launch(args.gpus, os.path.abspath(__file__)))
    import numpy as np
    import torch
    import torch.distributed as dist
    import osqp_amd
    import problems
    from osqp_amd import sharded
    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1)); local = int(os.environ.get('LOCAL_RANK', 0))
    assert world == args.gpus, 'WORLD_SIZE = %d but --gpus %d' % (world, args.gpus)
    use_dist = world > 1 or bool(os.environ.get('OSQP_BENCH_FORCE_DIST'))     # (one-rank RCCL run: exercises init / barrier / all_reduce on a 1-GPU box)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if use_dist:
        dist.init_process_group('nccl', device_id=dev)
    B = args.batch
    P, q, A, L, U = problems.mpc_batch(B)
    s = osqp_amd.OSQP(algebra='hip')
    s.setup(P, q, A, L[0], U[0], eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=4000, device=local)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        table, x, y, rng = sharded.solve_batch_sharded(s, l=L, u=U, rank=rank, world=world, device=dev if use_dist else None)
    barrier(); t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        table, x, y, rng = sharded.solve_batch_sharded(s, l=L, u=U, rank=rank, world=world, device=dev if use_dist else None)
        step_ms.append(1e3 * (time.perf_counter() - ts))
    barrier(); el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device='cuda')
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el_host = float(el.item())
    kernel_ms = s._solver.hip_stats()['gpu_solve_ms']               # the batch kernel's launch, HIP events on its stream (last host-array step)
    # ---- the same batch with every array RESIDENT in HBM (osqp_hip_batch_solve_device: bounds in, x / y / records out by device
    #      pointer, the zero-copy path of osqp_amd.nn.torch): this is `value`.  A step = this rank's share in one launch, the
    #      records (12 doubles per QP) to the host, one all_gather of the packed status table; x and y stay on the device. ----
    lo, hi = sharded.shard_range(B, rank, world)
    nb = hi - lo
    Ld, Ud = torch.tensor(L[lo:hi], device=dev), torch.tensor(U[lo:hi], device=dev)
    xd = torch.empty((nb, P.shape[0]), dtype=torch.float64, device=dev); yd = torch.empty((nb, A.shape[0]), dtype=torch.float64, device=dev)
    recd = torch.empty((nb, s._solver.BATCH_REC), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def device_step():
        s._solver.hip_batch_solve_device(nb, None, Ld.data_ptr(), Ud.data_ptr(), xd.data_ptr(), yd.data_ptr(), recd.data_ptr(), warm=False, stream=stream)
        rec = recd.cpu().numpy()                                     # (synchronises with the solve)
        recs = np.zeros((nb, len(sharded.RECORD_FIELDS))); recs[:, 0] = np.arange(lo, hi); recs[:, 1:6] = rec[:, 0:5]
        return sharded.gather_records(recs, B, device=dev if use_dist else None)
    for _ in range(max(args.warmup, 1)):
        table_d = device_step()
    barrier(); t0 = time.perf_counter()
    for _ in range(args.steps):
        table_d = device_step()
    barrier(); el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device='cuda')
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = float(el.item())
    assert np.array_equal(table_d[:, 1:3], table[:, 1:3]), 'device-pointer and host-array paths disagree on status / iterations'
    assert np.array_equal(xd.cpu().numpy(), x) and np.array_equal(yd.cpu().numpy(), y), 'device-pointer and host-array paths disagree on x / y'
    if rank == 0:
        out = {'metric': 'QPs/sec, batch of %d MPC QPs (n=120, m=240), eps 1e-6' % B, 'value': B * args.steps / el, 'unit': 'QP/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * el / args.steps, 'higher_is_better': True, 'scaling': 'strong',
               'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
               'config': {'workload': 'BASELINE configs[4]: %d MPC QPs, horizon 10, nx=8, nu=4 (problems.mpc_batch); bounds resident in HBM, x / y left in HBM, status table gathered to the host' % B,
                          'host_array_path': {'QP_per_s': B * args.steps / el_host, 'ms_per_step': 1e3 * el_host / args.steps,
                                              'note': 'osqp_hip_batch_solve with numpy arrays: H2D of the bounds and D2H of x, y, records (28 MB over PCIe per step) included'},
                          'ms_per_step_median': float(sorted(step_ms)[len(step_ms) // 2]), 'kernel_ms_last_step': kernel_ms,
                          'solved': int((table[:, 1] == 1).sum()), 'admm_iters_total': float(table[:, 2].sum()),
                          'admm_iters_per_s': float(table[:, 2].sum()) * args.steps / el}}
        # Roofline of the batch kernel (k_batch_admm<256, 6, 6, DIRECT, SPEC>: one workgroup per QP, iterates in LDS, the matrices and K^-1 in registers): neither
        # HBM (7.7 KB of vectors per QP in and out) nor MFMA applies -- the kernel is a chain of short dependent phases.  Since round 5 the linear solve
        # is the spectral form: one dense 128 x 128 fp64 product from registers (64 FMAs per thread: 256 issue cycles) instead
        # of the 240-pivot substitution chain.  Floor of one ADMM iteration = what one workgroup cannot overlap with itself: 9 barrier-separated phases
        # (t, B product, row sums, right-hand side copy, dense product, A product, row sums + z / y update, x update), each at least one LDS round trip
        # + barrier (~170 cycles: 128 + 40), + the product's FMA issue: 9 x 170 + 256 = 1 786 cycles at 2.4 GHz = 0.74 us.  achieved = ADMM iterations per
        # second per resident QP.  (The banded form's floor was the pivot chain: 2n x 43.5 cycles = 4.35 us; measured 13.9.)
        kernel_s = 1e-3 * kernel_ms
        iters_per_qp = float(table[:, 2].sum()) / B
        share = int(B // world)
        per_cu = 2 if share > 3 * 256 else 1      # batch_hip.hip: one workgroup per CU (everything in registers) up to three rounds, two per CU (scratch) beyond
        resident = per_cu * 256
        waves_of_qps = -(-share // resident)
        t_iter = kernel_s / (waves_of_qps * iters_per_qp)              # wall time of one ADMM iteration of a resident QP
        floor_iter = (9 * 170 + 256) / 2.4e9
        split = int(s._solver.hip_stats().get('batch_wave_split', -1))
        if split >= 0:
            # the wave-per-problem kernel (batch_hip.hip k_batch_wave: V and the matrices in LDS once per CU, a problem = one wave, eight in flight per CU).  Its
            # bound is VALU issue: a problem's solve is two dense n x n products by ONE wave -- 2 n^2 FMAs, each with its v_readlane broadcast and LDS read --
            # and the ELL products; floor = the FMAs of the two products alone at the CU's fp64 rate (4 SIMDs x 16 lanes): 2 x 120^2 / 64 cycles per QP-iteration.
            cus = 256
            t_cu = kernel_s / (float(table[:, 2].sum()) / cus)            # CU time per (problem, ADMM iteration)
            floor_cu = 2 * 120 * 120 / 64.0 / 2.4e9
            out['roofline'] = {'bound': 'valu', 'kernel': 'k_batch_wave<120> (one wave per QP, eight QPs in flight per CU; the %d longest-expected QPs on k_batch_admm beside it)' % split,
                               'unit': 'ADMM iter/s per CU', 'achieved': 1.0 / t_cu, 'peak': 1.0 / floor_cu, 'frac': floor_cu / t_cu, 'traffic': None,
                               'model': 'fp64 FMAs of the two dense products of a solve (2 x 120^2) at 64 FMAs per cycle per CU = %.2f us per QP-iteration; measured %.2f us '
                                        '(%.1f ADMM iterations per QP on average, kernel %.2f ms; ~3 500 wave instructions per QP-iteration: v_readlane broadcasts, LDS reads, ELL products, updates)'
                                        % (1e6 * floor_cu, 1e6 * t_cu, iters_per_qp, 1e3 * kernel_s)}
        else:
          out['roofline'] = {'bound': 'latency', 'kernel': 'k_batch_admm<256,6,6,true,false,false,true,%d> (spectral direct solve)' % per_cu, 'unit': 'ADMM iter/s per resident QP',
                           'achieved': 1.0 / t_iter, 'peak': 1.0 / floor_iter, 'frac': floor_iter / t_iter, 'traffic': None,
                           'model': 'chain of 9 barrier-separated phases (LDS round trip + barrier ~170 cycles each) + 64 fp64 FMAs per thread of the dense product (256 issue cycles) '
                                    '= %.2f us per ADMM iteration at 2.4 GHz; measured %.2f us (%.1f ADMM iterations per QP on average, %d QPs resident at a time in %d round(s), kernel %.2f ms)'
                                    % (1e6 * floor_iter, 1e6 * t_iter, iters_per_qp, resident, waves_of_qps, 1e3 * kernel_s)}
        if args.cpu_sample > 0 and world == 1:
            out['cpu_baseline'] = cpu_batch_baseline(P, q, A, L, U, cores=args.cpu_cores or None)
            out['config']['gpu_over_cpu_all_cores'] = out['value'] / out['cpu_baseline']['value']
        print(json.dumps(out))
    if use_dist:
        dist.barrier(); dist.destroy_process_group()


if __name__ == '__main__':
    main()
