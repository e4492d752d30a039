#!/bin/bash
# round 5, sixth box: the whole -m gpu tier on the round's sources so far (rings rule, DC pass in the closing launch, SSB thresholds), bench lines
mkdir -p gpurun_out
( time python3 -m pytest tests -m gpu -x -q ) > gpurun_out/r5_gputests3.log 2>&1
tail -4 gpurun_out/r5_gputests3.log
B="python3 bench.py --no-host-path --no-live-pmc --no-cpu-baseline --steps 40 --warmup 5"
for args in "" "--config 2" "--config 3" "--config 4" "--mode am --channels 4096 --log2-samples 16" "--mode usb --channels 4096 --log2-samples 16" \
            "--config 2 --log2-samples 14" "--config 3 --log2-samples 14" "--mode am --channels 4096 --log2-samples 14" "--mode usb --channels 4096 --log2-samples 14" \
            "--mode am --channels 1024 --log2-samples 14" "--mode usb --channels 1024 --log2-samples 14" "--config 0"; do
  out=$($B $args 2>/dev/null | grep '"metric"')
  echo "[$args] $(echo "$out" | grep -o '"ms_per_step": [0-9.]*') $(echo "$out" | grep -o '"from_idle_ms_per_step": [0-9.]*') $(echo "$out" | grep -o '"kernel_ms": [0-9.]*') $(echo "$out" | grep -o '"us_per_block[a-z_0-9]*": [0-9.]*' | head -2 | tr '\n' ' ')"
done 2>&1 | tee gpurun_out/r5_lines3.log
