#!/bin/bash
# round 6: short lead-ins by the default rule (channels of >= 8 segments, kernels of their own) against never / always, 7 interleaved rounds
mkdir -p gpurun_out/r6
for args in "--mode am --channels 4096 --log2-samples 16" "--mode usb --channels 4096 --log2-samples 16" "--config 2" "--config 3" "--config 4" \
            "--mode usb --channels 8192 --log2-samples 16" "--mode usb --channels 4096 --log2-samples 14" "--config 2 --log2-samples 14" "--mode am --channels 1024 --log2-samples 14"; do
  echo "## $args"
  tools/abenv.sh 7 "$args" - IQD_D4_LEADFREE=0 IQD_D4_LEADFREE=1
done 2>&1 | tee gpurun_out/r6/leadfree_rule_ab.txt
