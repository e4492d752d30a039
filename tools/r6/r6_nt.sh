#!/bin/bash
# round 6: the pipelines' output stores as NON-TEMPORAL stores (the compiler's own, nt bit): WBFM (-DIQD_ST_NT_STORES=1), FM / AM / SSB
# consumer lanes (-DIQD_D4_NT_STORES=1); parity, then interleaved against the shipped build
mkdir -p gpurun_out/r6
IQD_LIB=$PWD/tmp_variants/lib_ntst.so python3 -m pytest tests/test_gpu_stream.py tests/test_gpu_wbfm.py -m gpu -x -q 2>&1 | tail -1
IQD_LIB=$PWD/tmp_variants/lib_ntd4.so python3 -m pytest tests/test_gpu_stream2.py tests/test_gpu_bench_paths.py -m gpu -x -q 2>&1 | tail -1
A="IQD_LIB=$PWD/tmp_variants/lib_base.so"; B="IQD_LIB=$PWD/tmp_variants/lib_ntst.so"; C="IQD_LIB=$PWD/tmp_variants/lib_ntd4.so"
{
echo "## --config 1"; bash tools/abenv.sh 7 "--config 1" "$A" "$B"
for M in "--mode am --channels 4096 --log2-samples 16" "--mode usb --channels 4096 --log2-samples 16" "--config 2" "--config 2 --log2-samples 14"; do
  echo "## $M"; bash tools/abenv.sh 5 "$M" "$A" "$C"
done
} > gpurun_out/r6/nt_ab.txt 2>&1
cat gpurun_out/r6/nt_ab.txt
