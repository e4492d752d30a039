#!/bin/bash
# round 6: the whole -m gpu tier on the warm / cold segment geometry, then short lead-ins against full ones, interleaved, every configuration
mkdir -p gpurun_out/r6
( time python3 -m pytest tests -m gpu -x -q ) > gpurun_out/r6/eighth_gputests.log 2>&1
tail -6 gpurun_out/r6/eighth_gputests.log
for args in "--mode am --channels 4096 --log2-samples 16" "--mode usb --channels 4096 --log2-samples 16" "--config 2" "--config 3" "--config 4" \
            "--config 2 --log2-samples 14" "--config 3 --log2-samples 14" "--mode am --channels 4096 --log2-samples 14" "--mode usb --channels 4096 --log2-samples 14" \
            "--mode am --channels 1024 --log2-samples 14" "--mode usb --channels 1024 --log2-samples 14" "--mode fm --channels 1024 --log2-samples 16"; do
  echo "## $args"
  tools/abenv.sh 3 "$args" IQD_D4_LEADFREE=1 IQD_D4_LEADFREE=0
done 2>&1 | tee gpurun_out/r6/leadfree_ab.txt
