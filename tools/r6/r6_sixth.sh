#!/bin/bash
# round 6: the boundary replay inside the consumer waves (lane below = predecessor) - parity, then kernel times with and without
mkdir -p gpurun_out/r6
( time python3 -m pytest tests/test_gpu_stream2.py tests/test_gpu_modes.py tests/test_gpu_bench_paths.py tests/test_gpu_boundary.py -x -q ) > gpurun_out/r6/sixth_tests.log 2>&1
tail -12 gpurun_out/r6/sixth_tests.log
for m in usb am; do
  for lf in 1 0; do bash tools/r6/r6_kt.sh ${m}_lf$lf IQD_D4_LEADFREE=$lf --mode $m --channels 4096 --log2-samples 16; done
done
for lf in 1 0; do bash tools/r6/r6_kt.sh fm_lf$lf IQD_D4_LEADFREE=$lf --config 2; done
for lf in 1 0; do bash tools/r6/r6_kt.sh usb14_lf$lf IQD_D4_LEADFREE=$lf --mode usb --channels 4096 --log2-samples 14; done
for lf in 1 0; do bash tools/r6/r6_kt.sh fm14_lf$lf IQD_D4_LEADFREE=$lf --config 2 --log2-samples 14; done
for lf in 1 0; do bash tools/r6/r6_kt.sh mixed_lf$lf IQD_D4_LEADFREE=$lf --config 3; done
for lf in 1 0; do bash tools/r6/r6_kt.sh c4_lf$lf IQD_D4_LEADFREE=$lf --config 4; done
