#!/bin/bash
# round 5, first box: the whole -m gpu tier, then what the squelch magnitudes cost the AM / USB / FM pipelines
mkdir -p gpurun_out
( time python3 -m pytest tests -m gpu -x -q ) > gpurun_out/r5_gputests.log 2>&1
tail -5 gpurun_out/r5_gputests.log
B="python3 bench.py --no-host-path --no-live-pmc --no-cpu-baseline --steps 40 --warmup 5"
for i in 1 2; do
for m in am usb fm; do
  for mag in "" "--no-magnitude"; do
    out=$($B --mode $m --channels 4096 --log2-samples 16 $mag 2>/dev/null | grep '"metric"')
    echo "r$i $m $mag $(echo "$out" | grep -o '"ms_per_step": [0-9.]*') $(echo "$out" | grep -o '"kernel_ms": [0-9.]*')"
  done
done
done 2>&1 | tee gpurun_out/r5_nomag.log
$B 2>/dev/null | grep '"metric"' | tee gpurun_out/r5_default_line.json | cut -c1-400
