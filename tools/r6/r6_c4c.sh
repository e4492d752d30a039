#!/bin/bash
# round 6: four DC rows per workgroup (one per wave) in the closing launch - parity, then AM / USB / configs[3] / configs[4]
mkdir -p gpurun_out/r6
( python3 -m pytest tests/test_gpu_stream2.py tests/test_gpu_modes.py tests/test_gpu_bench_paths.py tests/test_gpu_scale.py tests/test_gpu_agc.py tests/test_gpu_boundary.py -x -q 2>&1 | tail -2 )
for args in "--mode am --channels 4096 --log2-samples 16" "--mode usb --channels 4096 --log2-samples 16" "--config 3" "--config 4" "--mode am --channels 4096 --log2-samples 14"; do
  echo "## $args"
  tools/abenv.sh 5 "$args" - IQD_SPLIT_TAIL_DC=1
done 2>&1 | tee gpurun_out/r6/dc_four_rows_ab.txt
bash tools/r6/r6_kt.sh am_4rows - --mode am --channels 4096 --log2-samples 16
bash tools/r6/r6_kt.sh c4_4rows - --config 4
bash tools/r6/r6_kt.sh c3_4rows - --config 3
