#!/bin/bash
# Runs ON THE GPU BOX (via gpurun):  rocprofv3 kernel trace + stats of the bench command, keeps only the summaries.
#   bash profiles/run_rocprof.sh <tag> [bench args...]
tag=${1:-r01}; shift
cd "$(dirname "$0")/.." && repo=$(pwd)
export TMPDIR=/tmp
out=/tmp/prof_$tag; rm -rf $out; mkdir -p $out gpurun_out
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench -- python $repo/bench.py --cpu-seconds 0 "$@" > $out/bench_stdout.log 2>&1)
grep '^{' $out/bench_stdout.log | tail -1 > gpurun_out/rocprof_${tag}_bench.json
f=$(find $out -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/rocprof_${tag}_kernel_stats.csv; fi
t=$(find $out -name '*kernel_trace.csv' | head -1)
if [ -n "$t" ]; then python $repo/profiles/active_stats.py "$t" gpurun_out/rocprof_${tag}_active_stats.csv; fi
find $out -name '*.csv' -size -2M -not -name '*kernel_trace*' -exec cp {} gpurun_out/ \; 2>/dev/null
ls -la $out $(dirname "$f") 2>/dev/null | head -20
cat gpurun_out/rocprof_${tag}_kernel_stats.csv 2>/dev/null | head -30
