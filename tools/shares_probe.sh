#!/bin/bash
# Several families in one launch: shares by time (plan_fused_by_time, the default) against shares in proportion to cost
# (IQD_SHARES=cost, the rule of round 3).  Run on the GPU box.
for cfg in "4096 16" "4096 14" "16384 14" "8192 16" "1400 16" "8192 13" "2048 14" "4096 15" "12000 14" "6000 16" "32768 13"; do
  set -- $cfg
  line="mixed ch $1 log2 $2"
  for m in time cost time cost; do
    if [ $m = time ]; then unset IQD_SHARES; else export IQD_SHARES=cost; fi
    out=$(python3 bench.py --mode mixed --channels $1 --log2-samples $2 --steps 20 --warmup 3 --prewarm-ms 50 --no-cpu-baseline --no-host-path --no-live-pmc 2>/dev/null)
    s=$(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    line="$line | $m $s"
  done
  echo "$line"
done
