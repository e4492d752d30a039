#!/bin/bash
# round 6: warm / cold segments (no records, no closing-launch fix-up) - parity, kernel times with and without, then the
# issue-arbitration probes (dynamic priority by lag; consumers on SIMD 3), interleaved on this box
mkdir -p gpurun_out/r6
( time python3 -m pytest tests/test_gpu_stream2.py tests/test_gpu_modes.py tests/test_gpu_bench_paths.py tests/test_gpu_boundary.py -x -q ) > gpurun_out/r6/seventh_tests.log 2>&1
tail -8 gpurun_out/r6/seventh_tests.log
for m in usb am; do
  for lf in 1 0; do bash tools/r6/r6_kt.sh ${m}_lf$lf IQD_D4_LEADFREE=$lf --mode $m --channels 4096 --log2-samples 16; done
done
for lf in 1 0; do bash tools/r6/r6_kt.sh fm_lf$lf IQD_D4_LEADFREE=$lf --config 2; done
for lf in 1 0; do bash tools/r6/r6_kt.sh usb14_lf$lf IQD_D4_LEADFREE=$lf --mode usb --channels 4096 --log2-samples 14; done
for lf in 1 0; do bash tools/r6/r6_kt.sh fm14_lf$lf IQD_D4_LEADFREE=$lf --config 2 --log2-samples 14; done
for m in "--mode am --channels 4096 --log2-samples 16" "--mode usb --channels 4096 --log2-samples 16" "--config 2"; do
  echo "## $m"
  tools/abn.sh 3 "$m --no-from-idle" tmp_variants/lib_base.so tmp_variants/lib_dyn1.so tmp_variants/lib_dyn2.so tmp_variants/lib_dyn3.so tmp_variants/lib_csimd3.so | grep median
done 2>&1 | tee gpurun_out/r6/probe_dynprio.txt
