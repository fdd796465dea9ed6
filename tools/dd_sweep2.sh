#!/bin/bash
# second sweep of the device-driven feeding knobs (results never depend on them): bench.py ms per cold solve and launches per solve
for cfg in "0.8 12" "0.5 12" "0.3 12" "0.5 8" "0.3 8" "0.3 6"; do
  set -- $cfg
  OSQP_HIP_POLL_FIRST=$1 OSQP_HIP_FINISH_PAIRS=$2 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('poll_first $1 finish_pairs $2:', round(d['ms_per_step'],2), 'ms,', int(d['config']['kernel_launches_per_solve']), 'launches')"
done
