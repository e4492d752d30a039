#!/bin/bash
# On the GPU box: the one-launch mixed call's shares by time with other per-family costs (IQD_FAMILY_NS=am,fm,wbfm,ssb, ns per sample of a
# segment) - interleaved rounds of bench.py, medians.   tools/famns_probe.sh "<bench args>" <rounds> <ns list> [<ns list> ...]   ("default" = none)
ARGS=$1; R=$2; shift 2
declare -A ALL
for i in $(seq 1 $R); do
  for NS in "$@"; do
    if [ "$NS" = default ]; then out=$(python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-path --no-live-pmc $ARGS 2>/dev/null)
    else out=$(IQD_FAMILY_NS=$NS python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-path --no-live-pmc $ARGS 2>/dev/null); fi
    s=$(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    ALL[$NS]="${ALL[$NS]} $s"
  done
done
for NS in "$@"; do python3 -c "import sys,statistics; v=[float(x) for x in sys.argv[2:]]; print('%-28s %-34s median ms_per_step %.4f (n=%d)' % (sys.argv[1], '$ARGS', statistics.median(v), len(v)))" "$NS" ${ALL[$NS]}; done
