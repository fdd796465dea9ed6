"""N > 1 path on CPU: 2 processes, torch.distributed 'gloo', each rank solves its share of a batch of independent QPs
on the host-simulator build and the result records are all_gathered (the GPU job does the same over RCCL)."""
import os
import subprocess
import sys

import numpy as np

from util import free_port

WORKER = r'''
import os, sys, warnings
warnings.simplefilter('ignore')
root = sys.argv[1]
for p in (root, os.path.join(root, 'osqp-python_amd'), os.path.join(root, 'tests'), os.path.join(root, 'oracle')):
    sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
import osqp_amd, problems
from osqp_amd import sharded
from hostsim_util import hostsim
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%s' % os.environ['MASTER_PORT'], rank=rank, world_size=world)
B = 5                                   # not divisible by 2: ragged shares
gen = lambda i: problems.random_qp(n=20, m=30, seed=100 + i)
with hostsim():
    mk = lambda: type('S', (), {})()
    def make():
        m = osqp_amd.OSQP()
        orig = m.setup
        m.setup = lambda P, q, A, l, u: orig(P, q, A, l, u, eps_abs=1e-6, eps_rel=1e-6, verbose=False)
        return m
    recs, xs = sharded.solve_local(gen, make, rank, world, B)
table = sharded.gather_records(recs, B)
np.save(os.path.join(sys.argv[2], 'table_%d.npy' % rank), table)
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_gloo_shard_and_gather(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script), root, str(tmp_path)], env=dict(env, RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    t0, t1 = np.load(tmp_path / 'table_0.npy'), np.load(tmp_path / 'table_1.npy')
    assert np.array_equal(t0, t1)                       # every rank holds the full table
    assert t0.shape[0] == 5 and list(t0[:, 0]) == [0, 1, 2, 3, 4]
    assert (t0[:, 1] == 1).all()                        # all solved
    # objective of problem 3 equals a single-process solve of the same problem (oracle as the checker)
    sys.path.insert(0, os.path.join(root, 'oracle')); sys.path.insert(0, root)
    import problems
    from oracle import Oracle
    P, q, A, l, u = problems.random_qp(n=20, m=30, seed=103)
    xo, yo, io = Oracle().setup(P, q, A, l, u, eps_abs=1e-8, eps_rel=1e-8, adaptive_rho_interval=50).solve()
    assert abs(t0[3, 3] - io.obj_val) < 1e-5 * (1 + abs(io.obj_val))


def test_shard_range_partition():
    from osqp_amd.sharded import shard_range
    for B in (1, 5, 8, 4096):
        for w in (1, 2, 3, 8):
            r = [shard_range(B, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == B and all(r[k][1] == r[k + 1][0] for k in range(w - 1))


LAYER_WORKER = r'''
import os, sys, warnings
warnings.simplefilter('ignore')
root = sys.argv[1]
for p in (root, os.path.join(root, 'osqp-python_amd'), os.path.join(root, 'tests'), os.path.join(root, 'oracle')):
    sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
import problems
from hostsim_util import hostsim
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%s' % os.environ['MASTER_PORT'], rank=rank, world_size=world)
P, q, A, L, U = problems.mpc_batch(5, nx=3, nu=2, N=4)          # 5 problems: ragged shares over 2 ranks
P = P.tocoo(); A = A.tocoo()
with hostsim():
    from osqp_amd.nn.torch import OSQP as Layer
    layer = Layer((P.row, P.col), P.shape, (A.row, A.col), A.shape, eps_abs=1e-7, eps_rel=1e-7)
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    x1 = layer(t(P.data), t(q), t(A.data), t(L), t(U))
    x2 = layer(t(P.data), t(q), t(A.data), t(L + 0.01), t(U + 0.01))       # second forward: no new setup
    np.save(os.path.join(sys.argv[2], 'layer_%d.npy' % rank), np.stack([x1.numpy(), x2.numpy()]))
    assert layer.setup_count == 1, layer.setup_count
dist.barrier(); dist.destroy_process_group()
'''


def test_torch_layer_two_rank_gloo(tmp_path):
    """The torch layer under torch.distributed: the batch is block-partitioned over the ranks, rows are all-gathered, every rank
    returns the full solution; the handle is set up once for both forwards (reference: nn/torch.py:136-140)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'layer_worker.py'
    script.write_text(LAYER_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script), root, str(tmp_path)], env=dict(env, RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    a, b = np.load(tmp_path / 'layer_0.npy'), np.load(tmp_path / 'layer_1.npy')
    assert np.array_equal(a, b) and a.shape[:2] == (2, 5)
    sys.path.insert(0, os.path.join(root, 'oracle')); sys.path.insert(0, root)
    import problems
    from oracle import Oracle
    P, q, A, L, U = problems.mpc_batch(5, nx=3, nu=2, N=4)
    for i in (0, 4):
        xo, yo, io = Oracle().setup(P, q, A, L[i], U[i], eps_abs=1e-9, eps_rel=1e-9, adaptive_rho_interval=50).solve()
        assert np.abs(a[0, i] - xo).max() < 1e-5 * (1 + np.abs(xo).max())
