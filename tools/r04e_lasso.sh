#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_woodbury.py "tests/test_gpu_baseline_configs.py::test_config3_lasso_full_size" "tests/test_gpu_baseline_configs.py::test_config3_lasso_tight_tolerance" -x -q < /dev/null 2>&1 | tail -3
for tol in 1e-6 1e-9; do
  OSQP_HIP_WOODBURY_DIRECT_TOL=$tol OSQP_HIP_WB_LOG=1 timeout 900 python bench.py --config lasso --steps 2 --warmup 1 --cpu-seconds 0 --batch 0 > gpurun_out/r04e_bench_lasso_$tol.json 2> gpurun_out/r04e_lasso_$tol.err < /dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r04e_bench_lasso_$tol.json")); c=d["config"]
print("$tol", d["ms_per_step"], c["admm_iters_per_step"], c["pcg_iters_per_admm_iter"], c["preconditioner"], c["woodbury_factorisations_last_solve"], c["woodbury_factor_ms_last_solve"], c["first_cold_solve_ms"], c.get("jacobi_only",{}).get("first_cold_solve_ms"))
PY
  grep -i "exact\|direct" gpurun_out/r04e_lasso_$tol.err | head -4
done
