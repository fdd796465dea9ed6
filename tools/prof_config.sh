#!/bin/bash
# rocprofv3 kernel statistics of one bench configuration:  bash tools/prof_config.sh <tag> <bench args...>   (runs on the GPU box)
tag=$1; shift
cd "$(dirname "$0")/.." && repo=$(pwd)
export TMPDIR=/tmp
out=/tmp/prof_$tag; rm -rf $out; mkdir -p $out gpurun_out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o pf -- python $repo/bench.py --cpu-seconds 0 --batch 0 --jacobi-leg 0 "$@" > $out/out.log 2>&1 < /dev/null)
f=$(find $out -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/${tag}_kernel_stats.csv; head -14 "$f" | cut -c1-170; else echo "no stats file"; tail -5 $out/out.log; fi
grep '^{' $out/out.log | tail -1 > gpurun_out/${tag}_bench_under_rocprof.json
