#!/bin/bash
# round 6: the closing launch's roles in another order - DC passes first - for AM / USB (ungated) and configs[4] (merged / split)
mkdir -p gpurun_out/r6
( python3 -m pytest tests/test_gpu_stream2.py tests/test_gpu_modes.py tests/test_gpu_bench_paths.py -x -q 2>&1 | tail -2 )
for args in "--mode am --channels 4096 --log2-samples 16" "--mode usb --channels 4096 --log2-samples 16" "--config 4" "--mode am --channels 4096 --log2-samples 14"; do
  echo "## $args"
  tools/abenv.sh 5 "$args" - IQD_DC_FIRST=0 IQD_SPLIT_TAIL_DC=1 "IQD_SPLIT_TAIL_DC=1 IQD_DC_FIRST=0"
done 2>&1 | tee gpurun_out/r6/dc_first_ab.txt
bash tools/r6/r6_kt.sh am_dcfirst - --mode am --channels 4096 --log2-samples 16
