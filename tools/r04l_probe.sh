#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_reorder.py tests/test_gpu_f1.py tests/test_gpu_parity.py -q -m gpu -x < /dev/null 2>&1 | tail -4
for n in 30000 50000 70000; do
  for f1 in 1 0; do
    OSQP_HIP_F1=$f1 timeout 300 python bench.py --n $n --steps 5 --warmup 1 --cpu-seconds 0 --batch 0 2>/dev/null < /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('n=$n f1=$f1', round(d['ms_per_step'],2), c['admm_iters_per_step'][0], c['pcg_kernels_per_iteration'], round(d['roofline']['pcg_iteration']['ms']*1e3,2), 'us/PCG-iteration', c['setup_s'])"
  done
done
OSQP_HIP_SETUP_TIMING=1 timeout 600 python bench.py --config lasso --steps 1 --warmup 0 --cpu-seconds 0 --batch 0 2>&1 < /dev/null | grep "osqp_hip setup" | head -20
