#!/bin/bash
# A/B of several engine builds on the SAME GPU box (box-to-box noise is larger than most kernel changes; on one box the
# bench repeats to +-0.3 %).  Put the builds in ab/libosqp_hip_<name>.so, then:  VARIANTS="base new" bash tools/ab_bench.sh
# Extra environment for one variant:  ENV_<name>="VAR=value"
for rep in 1 2; do
  for v in $VARIANTS; do
    lib=$PWD/ab/libosqp_hip_$v.so; [ -f $lib ] || lib=$PWD/osqp-python_amd/osqp_amd/libosqp_hip.so
    envvar=ENV_$v
    env ${!envvar} OSQP_HIP_LIBRARY=$lib python bench.py --steps 3 --warmup 1 --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/ab_$v.json
    python - <<PY
import json; d=json.load(open("gpurun_out/ab_$v.json")); r=d["roofline"]
print("$v", round(d["value"],1), round(d["ms_per_step"],2), d["config"]["admm_iters_per_solve"], round(d["config"]["pcg_iters_per_admm_iter"],2), int(d["config"]["kernel_launches_per_solve"]), "unconv", d["config"].get("pcg_budget_limited_iters"), "pcg_pair_us", round(r["pcg_iteration"]["ms"]*1e3,2), {k.split()[0]: round(v["ms_same_kernel_repeat"]*1e3,2) for k,v in r["kernels"].items()})
PY
  done
done
