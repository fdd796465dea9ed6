#!/bin/bash
# A/B of two engine builds on the same box: alternating runs
for rep in 1 2; do
  for v in $VARIANTS; do
    OSQP_HIP_LIBRARY=$PWD/ab/libosqp_hip_$v.so python bench.py --steps 3 --warmup 1 --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/ab_$v.json
    python - <<PY
import json; d=json.load(open("gpurun_out/ab_$v.json")); r=d["roofline"]
print("$v", round(d["value"],1), round(d["ms_per_step"],2), "pcg_pair_us", round(r["pcg_iteration"]["ms"]*1e3,2), {k.split()[0]: round(v["ms_same_kernel_repeat"]*1e3,2) for k,v in r["kernels"].items()})
PY
  done
done
