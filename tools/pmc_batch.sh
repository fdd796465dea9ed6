#!/bin/bash
# Instruction / wait counters of the batch kernel (one rocprofv3 pass per counter group; --pmc with --kernel-trace only):
#   bash tools/pmc_batch.sh <tag> [nbatch]
tag=$1; nb=${2:-4096}
cd "$(dirname "$0")/.." && repo=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cat > /tmp/pmc_batch_child.py <<PY
import sys, warnings; sys.path[:0] = ["$repo/osqp-python_amd", "$repo"]; warnings.simplefilter("ignore")
import osqp_amd, problems
P, q, A, L, U = problems.mpc_batch($nb)
s = osqp_amd.OSQP(); s.setup(P, q, A, L[0], U[0], eps_abs=1e-6, eps_rel=1e-6, verbose=False, max_iter=4000)
for rep in range(2):
    x, y, rec = s._solver.hip_batch_solve(l=L, u=U)
print("solved", int((rec[:, 0] == 1).sum()), "iterations", float(rec[:, 1].sum()))
PY
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  out=/tmp/pmcb_${tag}; rm -rf $out; mkdir -p $out
  (cd /tmp && timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -o pmc -- python /tmp/pmc_batch_child.py > $out/stdout.log 2>&1)
  f=$(find $out -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python $repo/profiles/summarize_pmc.py "$f" /tmp/pmcb.csv | grep -E "k_batch_admm|^kernel" | cut -c1-300; else tail -3 $out/stdout.log; fi
done
