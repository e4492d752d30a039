#!/bin/bash
# round 6: where short lead-ins lose what the piece count says they gain - channels per launch (segments per channel), replay and head stores off
mkdir -p gpurun_out/r6
for args in "--mode usb --channels 8192 --log2-samples 16" "--mode usb --channels 2048 --log2-samples 16" "--mode usb --channels 4096 --log2-samples 16 --no-magnitude" "--mode fm --channels 8192 --log2-samples 16" "--mode am --channels 8192 --log2-samples 16"; do
  echo "## $args"
  tools/abenv.sh 3 "$args" IQD_D4_LEADFREE=1 IQD_D4_PROBE=1 IQD_D4_PROBE=3 IQD_D4_LEADFREE=0
done 2>&1 | tee gpurun_out/r6/leadfree_where.txt
