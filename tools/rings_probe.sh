#!/bin/bash
# Rings per workgroup (IQD_RINGS=1|2|3, StreamArgs::rings) against the launch size: ms per step of single-family calls at the
# reference's operating point (one 64 ms block per channel and call) and around it.  Run on the GPU box from the repo root.
B="python3 bench.py --no-host-path --no-live-pmc --no-cpu-baseline --no-from-idle --steps 40 --warmup 5"
for shape in ${SHAPES:-"4096 14" "1024 14" "16384 14" "1024 16" "256 16"}; do
  set -- $shape
  for m in ${MODES:-fm am wbfm}; do
    line="$m ${1}x2^${2}:"
    for r in 3 2 1; do
      out=$(IQD_RINGS=$r IQD_WBFM_PATH=stream $B --mode $m --channels $1 --log2-samples $2 2>/dev/null | grep '"metric"')
      line="$line  R=$r $(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2) ($(echo "$out" | grep -o '"kernel_ms": [0-9.]*' | cut -d' ' -f2))"
    done
    echo "$line"
  done
done
