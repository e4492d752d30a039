#!/bin/bash
# Interleaved timing of SEVERAL builds of libiqdemod.so on ONE GPU box (run there, from the repo root):
#   tools/abn.sh <rounds> "<bench.py arguments>" <libA.so> <libB.so> [<libC.so> ...]
# Prints the timed kernels' HIP-event time and the step time per round and build, then the per-build medians
# (cdna_hip_programming.md 5.4 rule 24: never compare timings taken on different devices).
R=$1; ARGS=$2; shift 2
declare -A ALL
for i in $(seq 1 $R); do
  for L in "$@"; do
    out=$(IQD_LIB=$PWD/$L python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-path --no-live-pmc $ARGS 2>/dev/null)
    k=$(echo "$out" | grep -o '"kernel_ms": [0-9.]*' | cut -d' ' -f2)
    s=$(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    echo "round $i $L kernel_ms $k ms_per_step $s"
    ALL[$L]="${ALL[$L]} $k:$s"
  done
done
for L in "$@"; do
  python3 - "$L" ${ALL[$L]} <<'PY'
import sys, statistics
ks = [float(x.split(":")[0]) for x in sys.argv[2:] if x.split(":")[0]]
ss = [float(x.split(":")[1]) for x in sys.argv[2:] if x.split(":")[1]]
if ks: print("median %-40s kernel_ms %.4f  ms_per_step %.4f  (n=%d)" % (sys.argv[1], statistics.median(ks), statistics.median(ss), len(ks)))
PY
done
