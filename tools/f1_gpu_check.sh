#!/bin/bash
# quick GPU loop: parity tests of the F1 form, in-kernel trace, bench (no CPU baseline) with device-driven boundaries on and off
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_f1.py -x -q -s 2>&1 | grep -E "^n=|tight|passed|failed|Error|assert" | head -20
[ -z "$NOTRACE" ] && timeout 200 python tools/ktrace_f1.py 2>&1 | tail -14
for dd in 1 0; do
OSQP_HIP_DEVICE_DRIVEN=$dd timeout 120 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 > gpurun_out/f1_bench_dd$dd.json 2> gpurun_out/f1_bench.err
python - <<PY
import json
for f in ("gpurun_out/f1_bench_dd$dd.json",):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f, "ms/step %.2f it/s %.0f iters %d pcg/it %.2f launches %d frac %.3f ms/launch %.5f frac_streamed %s D %s" % (d["ms_per_step"], d["value"], d["config"]["admm_iters_per_solve"], d["config"]["pcg_iters_per_admm_iter"], d["config"]["kernel_launches_per_solve"], r["frac"], r["ms_per_launch"], r.get("frac_streamed"), r.get("replicas")))
    except Exception as e: print(f, "ERR", e); print(open("gpurun_out/f1_bench.err").read()[-2000:])
PY
done
