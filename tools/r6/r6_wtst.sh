#!/bin/bash
# round 6: the WBFM IIR lanes' 16-byte stores (PCM sectors, boundary records) as write-through stores (-DIQD_ST_WT_STORES=1): parity, A/B
mkdir -p gpurun_out/r6
IQD_LIB=$PWD/tmp_variants/lib_wtst.so python3 -m pytest tests/test_gpu_stream.py tests/test_gpu_wbfm.py -m gpu -x -q 2>&1 | tail -1
A="IQD_LIB=$PWD/tmp_variants/lib_base.so"; B="IQD_LIB=$PWD/tmp_variants/lib_wtst.so"
{
for M in "--config 1" "--config 1 --signal white" "--mode wbfm --channels 512 --log2-samples 16"; do
  echo "## $M"
  bash tools/abenv.sh 7 "$M" "$A" "$B"
done
} > gpurun_out/r6/wtst_ab.txt 2>&1
cat gpurun_out/r6/wtst_ab.txt
