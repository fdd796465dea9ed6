#!/bin/bash
# HBM-side traffic (rocprofv3 PMC, one pass per counter) of the PCG kernels under the time_kernel probes:
#   bash tools/pmc_probe.sh <tag> "<probe ids>"      e.g.  bash tools/pmc_probe.sh r02a "12 11 10"
# probe 12 = k_k2f repeated alone, 11 = k_k1f repeated alone, 10 = the pair as a solve alternates them.
tag=$1; probes=${2:-"12 11 10"}
cd "$(dirname "$0")/.." && repo=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cat > /tmp/pmc_probe_child.py <<PY
import sys, warnings; sys.path[:0] = ["$repo/osqp-python_amd", "$repo"]; warnings.simplefilter("ignore")
import osqp_amd, problems
P, q, A, l, u = problems.banded_qp(100000)
m = osqp_amd.OSQP(); m.setup(P, q, A, l, u, verbose=False); m.update_settings(max_iter=25); m.solve()
print(m._solver.hip_time_kernel(int(sys.argv[1]), 200) * 1e3, "us")
PY
for p in $probes; do
  for ctr in FETCH_SIZE TCC_HIT_sum TCC_MISS_sum; do
    out=/tmp/pmc_${tag}_${p}_$ctr; rm -rf $out; mkdir -p $out
    (cd /tmp && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o pmc -- python /tmp/pmc_probe_child.py $p > $out/stdout.log 2>&1)
    f=$(find $out -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then python $repo/profiles/summarize_pmc.py "$f" gpurun_out/pmc_${tag}_probe${p}_$ctr.csv | grep -E "k_k2f|k_k1f" ; else tail -5 $out/stdout.log; fi
  done
done
