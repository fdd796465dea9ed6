#!/bin/bash
# Instruction / wait counters of the PCG kernels under a time_kernel probe (one rocprofv3 pass per counter group):
#   bash tools/pmc_sq.sh <tag> <probe id>
tag=$1; p=${2:-12}
cd "$(dirname "$0")/.." && repo=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cat > /tmp/pmc_probe_child.py <<PY
import sys, warnings; sys.path[:0] = ["$repo/osqp-python_amd", "$repo"]; warnings.simplefilter("ignore")
import osqp_amd, problems
P, q, A, l, u = problems.banded_qp(100000)
m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False); m.update_settings(max_iter=25); m.solve()
print(m._solver.hip_time_kernel(int(sys.argv[1]), 100) * 1e3, "us")
PY
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  out=/tmp/pmcsq_${tag}; rm -rf $out; mkdir -p $out
  (cd /tmp && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -o pmc -- python /tmp/pmc_probe_child.py $p > $out/stdout.log 2>&1)
  f=$(find $out -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python $repo/profiles/summarize_pmc.py "$f" /tmp/pmcsq.csv | grep -E "^k_k2f|^k_k1f"; else tail -3 $out/stdout.log; fi
done
