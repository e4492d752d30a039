#!/bin/bash
# round 6: the DC pass in 5 KB of LDS, over int16 detector values in the PCM rows (in place) - parity, then against the int32 stream
mkdir -p gpurun_out/r6
( time python3 -m pytest tests/test_gpu_stream2.py tests/test_gpu_modes.py tests/test_gpu_bench_paths.py tests/test_gpu_boundary.py tests/test_gpu_scale.py tests/test_gpu_agc.py -x -q ) > gpurun_out/r6/thirteenth_tests.log 2>&1
tail -6 gpurun_out/r6/thirteenth_tests.log
for args in "--mode am --channels 4096 --log2-samples 16" "--mode usb --channels 4096 --log2-samples 16" "--config 3" "--config 4" "--mode usb --channels 4096 --log2-samples 14"; do
  echo "## $args"
  tools/abenv.sh 5 "$args" - IQD_NO_DET16=1
done 2>&1 | tee gpurun_out/r6/det16_ab.txt
bash tools/r6/r6_kt.sh am_det16 - --mode am --channels 4096 --log2-samples 16
bash tools/r6/r6_kt.sh c4_det16 - --config 4
