#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): HBM traffic counters of the bench command, one rocprofv3 pass per counter
# (TCC has 4 slots: FETCH_SIZE needs 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots"; counters only with --kernel-trace).
#   bash profiles/run_pmc.sh <tag> <workload tag as bench.py names it, e.g. banded_n100000> [bench args...]
# writes gpurun_out/<tag>_pmc_<workload>_<COUNTER>.csv -- copy them to profiles/: bench.py reports roofline.traffic only from a
# summary of the SAME workload.
tag=${1:-r03}; wl=${2:-banded_n100000}; shift; shift
cd "$(dirname "$0")/.." && repo=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=/tmp/pmc_${tag}_$ctr; rm -rf $out; mkdir -p $out
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o pmc -- python $repo/bench.py --cpu-seconds 0 --probe-reps 5 "$@" > $out/stdout.log 2>&1 < /dev/null)
  f=$(find $out -name '*counter_collection.csv' | head -1)
  echo "counter file: $f"
  if [ -n "$f" ]; then python $repo/profiles/summarize_pmc.py "$f" gpurun_out/${tag}_pmc_${wl}_$ctr.csv; else tail -5 $out/stdout.log; fi
done
