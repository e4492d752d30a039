#!/bin/bash
# like ab.sh but prints both the timed kernels' HIP-event time and the whole step: tools/ab_step.sh A.so B.so rounds [bench args]
A=$1; B=$2; R=${3:-3}; shift 3
for i in $(seq 1 $R); do
  for L in "$A" "$B"; do
    IQD_LIB=$PWD/$L python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-path "$@" 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.read()); print('round $i $L kernel_ms', o['roofline']['kernel_ms'], 'ms_per_step', o['ms_per_step'])"
  done
done
