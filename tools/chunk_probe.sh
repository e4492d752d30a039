#!/bin/bash
# configs[4] (squelch-gated, 8192 channels x 2^16) as channel chunks: does the pipelines' second read of a chunk come out of
# the 256 MB Infinity Cache when the chunk's pre-pass has just read it?  (run on the GPU box)
for extra in "" "--inline-prepass"; do
  for c in 1 2 4 8 16 32; do
    out=$(python3 bench.py --config 4 --channel-chunks $c $extra --steps 20 --warmup 3 --no-cpu-baseline --no-host-path --no-live-pmc 2>/dev/null)
    s=$(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    echo "chunks $c ($((1024 / c)) MiB each) $extra ms_per_step $s"
  done
done
