#!/bin/bash
# Interleaved A/B timing of two builds of libiqdemod.so on ONE GPU box (run there, from the repo root):
#   tools/ab.sh <libA.so> <libB.so> [rounds] [bench.py arguments ...]
# Prints the chain kernel's HIP-event time per round and build (cdna_hip_programming.md 5.4 rule 24:
# never compare timings taken on different devices).
A=$1; B=$2; R=${3:-4}; shift 3
for i in $(seq 1 $R); do
  for L in "$A" "$B"; do
    ms=$(IQD_LIB=$PWD/$L IQD_WBFM_PATH=stream python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-path "$@" 2>/dev/null | grep -o '"kernel_ms": [0-9.]*' | cut -d' ' -f2)
    echo "round $i $L kernel_ms $ms"
  done
done
