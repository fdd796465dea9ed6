"""GPU tier: the multi-GPU launch contract of bench.py / bench_batch.py on the one GPU a test box has -- a ONE-rank
torch.distributed.run job with OSQP_BENCH_FORCE_DIST=1, so that process-group initialisation (backend "nccl" = RCCL), the
barriers around the timed region, the max-over-ranks reduction and the all_gather of the per-problem records all execute on
device tensors.  (N > 1 ranks need N GPUs; the sharding arithmetic itself is covered by the 2-rank gloo tests.)"""
import json
import os
import subprocess
import sys

import pytest

from util import free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, extra):
    env = dict(os.environ, OSQP_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, script), '--gpus', '1'] + extra
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1]
    return json.loads(line)


def test_batch_bench_through_rccl_with_one_rank():
    d = _run('bench_batch.py', ['--steps', '1', '--warmup', '1', '--batch', '96', '--cpu-sample', '0'])
    assert d['n_gpus'] == 1 and d['config']['solved'] == 96 and d['value'] > 0 and d['scaling'] == 'strong'


def test_headline_bench_through_rccl_with_one_rank():
    d = _run('bench.py', ['--steps', '1', '--warmup', '0', '--vars', '4000', '--cpu-seconds', '0', '--probe-reps', '5', '--batch', '96', '--batch-steps', '1', '--hbm-n', '0'])
    assert d['n_gpus'] == 1 and d['config']['status'] == 'solved' and d['value'] > 0
    b = d['config']['batch']                                # BASELINE configs[4] through the sharded device path, all_gather over RCCL in the timed region
    assert b['solved'] == b['records'] == 96 and b['n_ranks_seen'] == 1 and b['QP_per_s'] > 0 and 'RCCL' in b['collective']


def test_device_resident_sharded_batch_through_rccl_with_one_rank():
    """sharded.solve_batch_sharded_device: this rank's share by device pointer, records assembled on the device, both all_gathers on
    device tensors over RCCL -- and no host copy of q / l / u / x (torch.Tensor.cpu is instrumented in the script); the two shares of a
    2-rank split reproduce the full batch bit for bit."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'tests', 'scripts', 'sharded_device_rccl.py')]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['solved'] == d['B'] and d['table_is_cuda'] and d['x_is_cuda']
    assert d['max_dx_vs_host_path'] == 0.0 and d['max_dx_two_shares'] == 0.0
    assert d['shares'] == [[0, 18], [18, 37]] and d['indices_ok']
    assert d['big_host_copies'] == []
