#!/bin/bash
# sharded batch path: where a step goes; portfolio bench with the Jacobi-only leg
cd "$(dirname "$0")/.."
python tools/sharded_overhead.py 2>&1 | tail -6
python - <<'PY' 2>&1 | tail -4
import sys; sys.path[:0]=[".","osqp-python_amd"]
import warnings; warnings.simplefilter("ignore")
import bench_batch, torch
torch.cuda.set_device(0)
for i in range(2):
    d = bench_batch.measure_sharded_device(4096, 5, 1, 0, 1, 0, False); d.pop("_data"); print(d["QP_per_s"], d["ms_per_batch"])
PY
python bench.py --config portfolio --steps 3 --warmup 1 --cpu-seconds 0 --batch 0 > gpurun_out/r04d_bench_portfolio.json 2>gpurun_out/r04d_portfolio.err
python - <<'PY'
import json; d=json.load(open("gpurun_out/r04d_bench_portfolio.json")); c=d["config"]; print(d["ms_per_step"], c["preconditioner"], c["woodbury_factorisations_last_solve"], c["woodbury_factor_ms_last_solve"], c.get("jacobi_only"))
PY
tail -3 gpurun_out/r04d_portfolio.err
