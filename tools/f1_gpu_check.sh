#!/bin/bash
# quick GPU loop for the F1 kernel: parity tests, in-kernel trace, bench (no CPU baseline)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_f1.py -x -q -s 2>&1 | grep -E "^n=|tight|passed|failed|Error|assert" | head -20
timeout 200 python tools/ktrace_f1.py 2>&1 | tail -14
timeout 120 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 > gpurun_out/f1_bench.json 2> gpurun_out/f1_bench.err
python - <<PY
import json
for f in ("gpurun_out/f1_bench.json",):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f, d["ms_per_step"], d["value"], d["config"]["admm_iters_per_solve"], d["config"]["pcg_iters_per_admm_iter"], d["config"]["kernel_launches_per_solve"], r["frac"], r["ms_per_launch"], r.get("frac_streamed"), r.get("replicas"))
        for k,v in r["kernels"].items(): print("   ", k, v["ms_same_kernel_repeat"], v["ms"])
    except Exception as e: print(f, "ERR", e)
PY
