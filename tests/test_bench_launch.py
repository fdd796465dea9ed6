"""The driver's launch contract: `python bench.py --gpus N ...` with NO launcher around it must start its N ranks itself (torch.distributed.run,
one process per GPU) -- here on the CPU tier: N = 2, the engine's host driver on the host simulator (OSQP_BENCH_HOSTSIM=1, a test-only switch
of bench.py), gloo.  Checks what a scaling run needs from the line: both ranks' records arrived, the sharded batch was solved by both ranks,
per-rank times are reported.  (The GPU-tier counterpart with RCCL is tests/test_gpu_rccl_one_rank.py.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, extra):
    env = dict(os.environ, OSQP_BENCH_HOSTSIM='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)                                   # exactly the driver's situation: no launcher environment
    out = subprocess.run([sys.executable, os.path.join(ROOT, script)] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]             # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks():
    d = _run('bench.py', ['--gpus', '2', '--steps', '1', '--warmup', '0', '--vars', '300', '--batch', '6', '--batch-steps', '1', '--single-device'])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['value'] > 0
    assert len(d['config']['per_rank']) == 2 and all(r['status'] == 1 for r in d['config']['per_rank'])
    assert d['config']['per_rank'][0]['iter'] == d['config']['per_rank'][1]['iter']          # replicas of one QP
    b = d['config']['batch']
    assert b['n_ranks_seen'] == 2 and b['solved'] == b['records'] == 6 and len(b['per_rank_ms']) == 2
    assert 'HOST SIMULATOR' in d['data'] and d['roofline'] is None                            # never mistaken for a measurement
    assert d['config']['solve_ms_hipevent_median'] is not None and d['config']['ms_per_step_median'] is not None


def test_single_rank_needs_no_launcher():
    d = _run('bench.py', ['--steps', '1', '--warmup', '0', '--vars', '300', '--batch', '4', '--batch-steps', '1'])
    assert d['n_gpus'] == 1 and d['config']['batch']['n_ranks_seen'] == 1 and d['config']['batch']['solved'] == 4
