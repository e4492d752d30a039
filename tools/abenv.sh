#!/bin/bash
# Interleaved timing of ONE library under several environments on one GPU box:
#   tools/abenv.sh <rounds> "<bench.py arguments>" "<ENV=1 ...>" "<ENV=0 ...>" [...]      ("-" = no extra environment)
R=$1; ARGS=$2; shift 2
declare -A ALL
for i in $(seq 1 $R); do
  for E in "$@"; do
    if [ "$E" = "-" ]; then EV=""; else EV="$E"; fi
    out=$(env $EV python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-host-path --no-live-pmc --no-from-idle $ARGS 2>/dev/null)
    k=$(echo "$out" | grep -o '"kernel_ms": [0-9.]*' | cut -d' ' -f2)
    s=$(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    ALL[$E]="${ALL[$E]} $k:$s"
  done
done
for E in "$@"; do
  python3 - "$E" ${ALL[$E]} <<'PY'
import sys, statistics
ks = [float(x.split(":")[0]) for x in sys.argv[2:] if x.split(":")[0]]
ss = [float(x.split(":")[1]) for x in sys.argv[2:] if x.split(":")[1]]
if ss: print("median [%-24s] kernel_ms %.4f  ms_per_step %.5f  (n=%d)" % (sys.argv[1], statistics.median(ks) if ks else float("nan"), statistics.median(ss), len(ss)))
PY
done
