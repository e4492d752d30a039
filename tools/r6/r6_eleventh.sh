#!/bin/bash
# round 6: does the segment spacing matter?  USB / FM 8192 x 2^16 with other segment-length granules (other spacings of a channel's segments)
mkdir -p gpurun_out/r6
for args in "--mode usb --channels 8192 --log2-samples 16" "--mode fm --channels 8192 --log2-samples 16"; do
  echo "## $args"
  tools/abenv.sh 3 "$args" IQD_D4_LEADFREE=1 "IQD_D4_LEADFREE=1 IQD_D4_GRAN=256" "IQD_D4_LEADFREE=1 IQD_D4_GRAN=512" IQD_D4_LEADFREE=0 "IQD_D4_LEADFREE=0 IQD_D4_GRAN=256" "IQD_D4_LEADFREE=0 IQD_D4_GRAN=512"
done 2>&1 | tee gpurun_out/r6/leadfree_gran.txt
