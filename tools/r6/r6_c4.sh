#!/bin/bash
# round 6: configs[4] - tail update and DC pass in one launch (against one behind the other), the pre-pass stream at low priority
mkdir -p gpurun_out/r6
( python3 -m pytest tests/test_gpu_stream2.py tests/test_gpu_modes.py tests/test_gpu_bench_paths.py tests/test_gpu_scale.py tests/test_gpu_agc.py tests/test_gpu_boundary.py -x -q 2>&1 | tail -2 )
for args in "--config 4" "--config 4 --inline-prepass" "--mode usb --channels 8192 --log2-samples 16 --squelch -60"; do
  echo "## $args"
  tools/abenv.sh 5 "$args" - IQD_SPLIT_TAIL_DC=1 IQD_PRE_PRIO=low
done 2>&1 | tee gpurun_out/r6/c4_tail_dc_ab.txt
bash tools/r6/r6_kt.sh c4_taildc - --config 4
