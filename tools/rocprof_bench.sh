#!/bin/bash
# the rocprofv3 part of tools/collect_profiles.sh alone: kernel statistics + gap analysis of the headline bench
#   bash tools/rocprof_bench.sh <tag>
tag=${1:-r03}
cd "$(dirname "$0")/.." && repo=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
out=/tmp/prof_$tag; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench -- python $repo/bench.py --cpu-seconds 0 --steps 5 --warmup 1 > $out/bench_stdout.log 2>&1)
grep '^{' $out/bench_stdout.log | tail -1 > gpurun_out/${tag}_bench_under_rocprof.json
f=$(find $out -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${tag}_rocprofv3_kernel_stats.csv
t=$(find $out -name '*kernel_trace.csv' | head -1)
if [ -n "$t" ]; then python profiles/active_stats.py "$t" gpurun_out/${tag}_rocprofv3_active_stats.csv > /dev/null; python tools/gap_analysis.py "$t" > gpurun_out/${tag}_gap_analysis.txt; cat gpurun_out/${tag}_gap_analysis.txt; fi
