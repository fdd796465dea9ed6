"""Multi-GPU execution of a batch of independent QPs (BASELINE north_star: "a batch of independent QPs shards
one-problem-per-GPU ... with only a final RCCL status/objective gather over xGMI").

There is no reference analogue with more than one process; the closest semantics are the reference's thread pools over
independent solver objects (/root/reference/src/osqp/nn/torch.py:200-224, src/osqp/tests/multithread_test.py:44-53): every
problem is solved on its own, results are collected per problem.  Here: problem i of B goes to rank  i*world // B
(contiguous blocks), each rank drives its own GPU (OSQPSettings.device = LOCAL_RANK) with no data-path collective, and ONE
all_gather of a packed record per problem ends the job (torch.distributed backend "nccl" = RCCL on ROCm, "gloo" on CPU).
"""
import numpy as np

RECORD_FIELDS = ('index', 'status_val', 'iter', 'obj_val', 'prim_res', 'dual_res', 'solve_time')


def shard_range(nproblems, rank, world):
    """Contiguous block [lo, hi) of problem indices owned by `rank`."""
    return (nproblems * rank) // world, (nproblems * (rank + 1)) // world


def solve_local(problems_iter, make_solver, rank=0, world=1, nproblems=None):
    """Solve this rank's share.  problems_iter(i) -> (P, q, A, l, u); make_solver() -> an un-setup osqp_amd.OSQP.
    Same-structure consecutive problems re-use one solver through update(), like nn/torch.py:136-140."""
    lo, hi = shard_range(nproblems, rank, world)
    recs = np.zeros((hi - lo, len(RECORD_FIELDS)))
    xs = []
    for k, i in enumerate(range(lo, hi)):
        P, q, A, l, u = problems_iter(i)
        solver = make_solver()
        solver.setup(P, q, A, l, u)
        r = solver.solve()
        recs[k] = (i, r.info.status_val, r.info.iter, r.info.obj_val, r.info.prim_res, r.info.dual_res, r.info.solve_time)
        xs.append(r.x)
    return recs, xs


def gather_records(recs, nproblems, device=None):
    """all_gather of the per-problem records; every rank returns the full (nproblems x fields) table ordered by index.
    Ranks may own different counts (B not divisible by world): records are padded to the largest share."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):        # (a one-rank group still goes through the collective: smoke-tests RCCL)
        return recs[np.argsort(recs[:, 0])]
    world = dist.get_world_size()
    share = max(shard_range(nproblems, r, world)[1] - shard_range(nproblems, r, world)[0] for r in range(world))
    pad = np.full((share, recs.shape[1]), -1.0)
    pad[:len(recs)] = recs
    t = torch.as_tensor(pad, dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    table = torch.cat(out).cpu().numpy()
    table = table[table[:, 0] >= 0]
    return table[np.argsort(table[:, 0])]


def solve_batch_sharded(solver, q=None, l=None, u=None, rank=0, world=1, device=None):
    """BASELINE configs[4] path: a batch of B same-structure QPs (rows of q / l / u) is block-partitioned over the ranks
    (SURVEY.md §8e), each rank solves its share with ONE batched kernel launch on its GPU (osqp_hip_batch_solve: one
    workgroup per problem), then ONE all_gather of the packed per-problem records.  `solver` is a set-up osqp_amd.OSQP
    holding the shared P, A and settings.  Returns (table[B, fields], x_local, y_local, (lo, hi))."""
    B = next(np.asarray(a).shape[0] for a in (q, l, u) if a is not None)
    lo, hi = shard_range(B, rank, world)
    sl = lambda a: None if a is None else np.asarray(a)[lo:hi]
    try:
        x, y, rec = solver._solver.hip_batch_solve(q=sl(q), l=sl(l), u=sl(u), nbatch=hi - lo)
    except ValueError:                         # the problem does not fit the one-workgroup batch kernel: this rank's share, one
        x = np.zeros((hi - lo, solver.n)); y = np.zeros((hi - lo, solver.m)); rec = np.zeros((hi - lo, 12))   # update()+solve() at a time
        for k in range(hi - lo):
            solver.update(**{name: a[k] for name, a in (('q', sl(q)), ('l', sl(l)), ('u', sl(u))) if a is not None})
            r = solver.solve()
            x[k], y[k] = r.x, r.y
            rec[k, 0:5] = (r.info.status_val, r.info.iter, r.info.obj_val, r.info.prim_res, r.info.dual_res)
    recs = np.zeros((hi - lo, len(RECORD_FIELDS)))
    recs[:, 0] = np.arange(lo, hi)
    recs[:, 1:6] = rec[:, 0:5]                 # status_val, iter, obj_val, prim_res, dual_res
    return gather_records(recs, B, device=device), x, y, (lo, hi)


def gather_rows(rows_local, nproblems, device=None):
    """all_gather of per-problem rows (e.g. the solutions x of every rank's share, block-partitioned like shard_range):
    every rank returns the full (nproblems x k) array in problem order.  One collective; shares may differ by one row."""
    import torch
    import torch.distributed as dist
    rows_local = np.asarray(rows_local, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()):
        return rows_local
    world = dist.get_world_size()
    spans = [shard_range(nproblems, r, world) for r in range(world)]
    share = max(hi - lo for lo, hi in spans)
    pad = np.zeros((share, rows_local.shape[1]))
    pad[:len(rows_local)] = rows_local
    t = torch.as_tensor(pad, dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.concatenate([o.cpu().numpy()[:hi - lo] for o, (lo, hi) in zip(out, spans)])


# ---------------------------------------------------------------------------------------------------------------- device-resident
def solve_batch_sharded_device(solver, q=None, l=None, u=None, rank=0, world=1, total=None):
    """The same split with every array resident on the GPU (torch ROCm tensors, float64, shape (B, n) / (B, m); None = the solver's
    own vector): this rank's contiguous row block goes to osqp_hip_batch_solve_device by device pointer on torch's current stream --
    no host copy of q / l / u / x / y -- the 7-field records are assembled on the device, and the ONE all_gather (RCCL) runs on device
    tensors.  Returns (table[B, fields] device tensor, x_local, y_local device tensors, (lo, hi)).  Raises ValueError(str(code)) like
    hip_batch_solve when the problem does not fit the batch kernel.
    total = B: the tensors hold ONLY this rank's row block [lo, hi) of a batch of B problems (a rank need not upload the rows of the others)."""
    import torch
    import torch.distributed as dist
    ref = next(a for a in (q, l, u) if a is not None)
    assert ref.is_cuda and all(a is None or (a.is_cuda and a.dtype == torch.float64 and a.dim() == 2) for a in (q, l, u))
    B, dev = (ref.shape[0] if total is None else int(total)), ref.device
    lo, hi = shard_range(B, rank, world)
    nb = hi - lo
    if total is not None:
        assert ref.shape[0] == nb, 'total=%d: rank %d of %d holds rows [%d, %d), got %d rows' % (B, rank, world, lo, hi, ref.shape[0])
    sl = lambda a: None if a is None else (a if total is not None else a[lo:hi]).contiguous()          # (a row block of a contiguous tensor: a view, no copy)
    ql, ll, ul = sl(q), sl(l), sl(u)
    x = torch.empty((nb, solver.n), dtype=torch.float64, device=dev)
    y = torch.empty((nb, solver.m), dtype=torch.float64, device=dev)
    rec = torch.zeros((max(nb, 1), 12), dtype=torch.float64, device=dev)        # OSQP_HIP_BATCH_REC
    # (a rank with an EMPTY share calls too -- nbatch = 0 launches nothing and raises exactly when a non-empty share would: every rank takes
    #  the same branch before the job's first collective)
    ptr = lambda t: None if t is None else t.data_ptr()
    solver._solver.hip_batch_solve_device(nb, ptr(ql), ptr(ll), ptr(ul), x.data_ptr(), y.data_ptr(), rec.data_ptr(), warm=False,
                                          stream=torch.cuda.current_stream(dev).cuda_stream)
    recs = torch.zeros((nb, len(RECORD_FIELDS)), dtype=torch.float64, device=dev)
    recs[:, 0] = torch.arange(lo, hi, dtype=torch.float64, device=dev)
    recs[:, 1:6] = rec[:nb, 0:5]                     # status_val, iter, obj_val, prim_res, dual_res
    if not (dist.is_available() and dist.is_initialized()):
        return recs, x, y, (lo, hi)
    wsz = dist.get_world_size()
    share = max(shard_range(B, r, wsz)[1] - shard_range(B, r, wsz)[0] for r in range(wsz))
    pad = torch.full((share, recs.shape[1]), -1.0, dtype=torch.float64, device=dev)
    pad[:nb] = recs
    out = [torch.empty_like(pad) for _ in range(wsz)]
    dist.all_gather(out, pad)                        # the job's one collective: 56 bytes per problem over xGMI
    table = torch.cat(out)
    table = table[table[:, 0] >= 0]
    return table[torch.argsort(table[:, 0])], x, y, (lo, hi)


def gather_rows_device(rows_local, nproblems):
    """all_gather of per-problem rows held in a device tensor (block-partitioned like shard_range); returns the full (nproblems, k)
    device tensor in problem order.  One collective on device tensors; shares may differ by one row."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return rows_local
    wsz = dist.get_world_size()
    spans = [shard_range(nproblems, r, wsz) for r in range(wsz)]
    share = max(hi - lo for lo, hi in spans)
    pad = torch.zeros((share, rows_local.shape[1]), dtype=rows_local.dtype, device=rows_local.device)
    pad[:rows_local.shape[0]] = rows_local
    out = [torch.empty_like(pad) for _ in range(wsz)]
    dist.all_gather(out, pad)
    return torch.cat([o[:hi - lo] for o, (lo, hi) in zip(out, spans)])
