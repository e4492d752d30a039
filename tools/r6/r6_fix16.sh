#!/bin/bash
# round 6: the WBFM boundary fix-up with 16 threads per segment (16 segments per workgroup) instead of 32 (8): parity, A/B
mkdir -p gpurun_out/r6
python3 -m pytest tests/test_gpu_stream.py tests/test_gpu_wbfm.py tests/test_gpu_scale.py tests/test_gpu_gain_epochs.py tests/test_gpu_bench.py tests/test_gpu_fuzz_pins.py -m gpu -x -q 2>&1 | tail -1
A="IQD_LIB=$PWD/tmp_variants/lib_fix32.so"; B="IQD_LIB=$PWD/tmp_variants/lib_fix16.so"
{
for M in "--config 1" "--config 3" "--mode wbfm --channels 512 --log2-samples 16"; do
  echo "## $M"; bash tools/abenv.sh 7 "$M" "$A" "$B"
done
bash tools/r6/r6_kt.sh fix16 - --config 1
} > gpurun_out/r6/fix16_ab.txt 2>&1
cat gpurun_out/r6/fix16_ab.txt | cut -c1-200
