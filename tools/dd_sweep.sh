#!/bin/bash
# sweep of the device-driven scheduling knobs (results never depend on them): wall time of three solves of config 2
for cfg in "6 12 30" "12 16 30" "16 24 20" "24 32 20" "32 48 20" "48 64 20"; do
  set -- $cfg
  echo "== poll_low $1 finish_pairs $2 sleep_us $3"
  OSQP_HIP_POLL_LOW=$1 OSQP_HIP_FINISH_PAIRS=$2 OSQP_HIP_POLL_SLEEP_US=$3 timeout 300 python tools/dd_debug.py 2>&1 | grep "dd=1"
done
