#!/bin/bash
# round 6: per-family tile lead-ins (fir_halo) and kept tails (tail_keep): the whole GPU tier, then old / new builds interleaved
mkdir -p gpurun_out/r6
( time python3 -m pytest tests -m gpu -x -q ) > gpurun_out/r6/keep_tests.log 2>&1
tail -5 gpurun_out/r6/keep_tests.log
A="IQD_LIB=$PWD/tmp_variants/lib_keep.so"; B="IQD_LIB=$PWD/tmp_variants/lib_r6old.so"
{
for M in "--mode am --channels 4096 --log2-samples 16" "--mode usb --channels 4096 --log2-samples 16" "--config 2" "--config 3" \
         "--config 2 --log2-samples 14" "--mode am --channels 4096 --log2-samples 14" "--config 3 --log2-samples 14" \
         "--mode fm --channels 512 --log2-samples 16" "--mode am --channels 512 --log2-samples 16" "--config 0"; do
  echo "## $M"
  bash tools/abenv.sh 5 "$M" "$A" "$B"
done
} > gpurun_out/r6/keep_ab.txt 2>&1
cat gpurun_out/r6/keep_ab.txt
