#!/bin/bash
# round 6: write-through stores (nothing dirty in the L2 at the kernel boundary): the closing launch's PCM + tails (lib_wtclose), the
# consumer lanes' 8-byte stores (lib_wtd4) against the shipped build, interleaved on one box; parity of both first
mkdir -p gpurun_out/r6
for V in wtclose wtd4; do
  IQD_LIB=$PWD/tmp_variants/lib_$V.so python3 -m pytest tests/test_gpu_bench_paths.py tests/test_gpu_modes.py -m gpu -x -q 2>&1 | tail -1
done
A="IQD_LIB=$PWD/tmp_variants/lib_base.so"; B="IQD_LIB=$PWD/tmp_variants/lib_wtclose.so"; C="IQD_LIB=$PWD/tmp_variants/lib_wtd4.so"
{
for M in "--mode am --channels 4096 --log2-samples 16" "--mode usb --channels 4096 --log2-samples 16" "--config 2" "--config 3" \
         "--config 2 --log2-samples 14" "--mode am --channels 4096 --log2-samples 14" "--config 4"; do
  echo "## $M"
  bash tools/abenv.sh 5 "$M" "$A" "$B" "$C"
done
} > gpurun_out/r6/wt_ab.txt 2>&1
cat gpurun_out/r6/wt_ab.txt
