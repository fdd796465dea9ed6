#!/bin/bash
# A/B of two builds: per-launch durations of k_slot1 in a solve.  bash tools/slot_sequence.sh [library]
export TMPDIR=/tmp
repo=$(pwd)
for lib in "$@"; do
  out=/tmp/prof_seq_$(basename $lib .so); rm -rf $out; mkdir -p $out
  (cd /tmp && OSQP_HIP_LIBRARY=$repo/osqp-python_amd/osqp_amd/$lib OSQP_HIP_DEVICE_DRIVEN=0 rocprofv3 --kernel-trace --output-format csv -d $out -o bench -- python $repo/bench.py --cpu-seconds 0 --steps 3 --warmup 1 > $out/stdout.log 2>&1)
  echo "=== $lib: $(grep '^{' $out/stdout.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],2), d["config"]["kernel_launches_per_solve"])')"
  python tools/slot_sequence.py $(find $out -name '*kernel_trace.csv' | head -1) 100
done
