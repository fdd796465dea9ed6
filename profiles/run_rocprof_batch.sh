#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + stats of the batch benchmark, keeps only the summary.
#   bash profiles/run_rocprof_batch.sh <tag> [bench_batch args...]
tag=${1:-r01}; shift
cd "$(dirname "$0")/.." && repo=$(pwd)
export TMPDIR=/tmp
out=/tmp/profb_$tag; rm -rf $out; mkdir -p $out gpurun_out
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o batch -- python $repo/bench_batch.py --cpu-sample 0 "$@" > $out/stdout.log 2>&1)
tail -1 $out/stdout.log > gpurun_out/rocprof_batch_${tag}_bench.json
f=$(find $out -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/rocprof_batch_${tag}_kernel_stats.csv; cat "$f" | head -8; else tail -5 $out/stdout.log; fi
