#!/bin/bash
# What binds the two-kernel PCG iteration on UNSTRUCTURED columns: L2 / fabric request counters of k_k2f and k_k1f on the
# unstructured variant of config 2 against the same kernels on the banded one (OSQP_HIP_F1=0 keeps the banded QP on the two-kernel form).
# One rocprofv3 pass per counter group (counters only with --kernel-trace).   bash tools/pmc_unstructured.sh <tag>     (on the GPU box)
tag=${1:-r04}
cd "$(dirname "$0")/.." && repo=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in unstructured banded; do
  for ctr in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum" "FETCH_SIZE"; do
    name=$(echo $ctr | tr ' ' '+')
    out=/tmp/pmcu_${cfg}_$name; rm -rf $out; mkdir -p $out
    (cd /tmp && OSQP_HIP_F1=0 timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o pmc -- python $repo/bench.py --config $cfg --cpu-seconds 0 --batch 0 --steps 1 --warmup 0 --probe-reps 5 > $out/stdout.log 2>&1 < /dev/null)
    f=$(find $out -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then python profiles/summarize_pmc.py "$f" gpurun_out/${tag}_pmcu_${cfg}_$name.csv | grep -E "^k_k2f|^k_k1f|^k_slot_a|^k_slot_b"; else echo "no counter file for $cfg $ctr"; tail -3 $out/stdout.log; fi
  done
done
