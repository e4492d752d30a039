#!/bin/bash
# round 6: the one launch's shares with short lead-ins - ns per sample of a segment (IQD_FAMILY_NS=am,fm,wbfm,ssb), interleaved on one box
mkdir -p gpurun_out/r6
for args in "--config 3" "--config 3 --log2-samples 14" "--mode mixed --channels 8192 --log2-samples 16"; do
  echo "## $args"
  tools/abenv.sh 3 "$args" - IQD_FAMILY_NS=22.2,39.1,58.3,22.4 IQD_FAMILY_NS=22.6,40.0,58.3,23.0 IQD_FAMILY_NS=22.2,41.0,58.3,23.5 IQD_D4_LEADFREE=0
done 2>&1 | tee gpurun_out/r6/famns_leadfree.txt
