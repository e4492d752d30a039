#!/bin/bash
# round 6, third box: FM / AM / SSB pipelines with 128-sample lead-ins + boundary records (iqd_d4_fix.h): parity first, then timing
mkdir -p gpurun_out/r6
( time python3 -m pytest tests/test_gpu_stream2.py tests/test_gpu_modes.py tests/test_gpu_bench_paths.py tests/test_gpu_boundary.py -x -q ) > gpurun_out/r6/third_tests.log 2>&1
tail -15 gpurun_out/r6/third_tests.log
B="python3 bench.py --no-host-path --no-live-pmc --no-cpu-baseline --no-from-idle --steps 40 --warmup 5"
for i in 1 2; do
for args in "--mode am --channels 4096 --log2-samples 16" "--mode usb --channels 4096 --log2-samples 16" "--config 2" "--config 3" "--config 4" \
            "--config 2 --log2-samples 14" "--config 3 --log2-samples 14" "--mode am --channels 4096 --log2-samples 14" "--mode usb --channels 4096 --log2-samples 14"; do
  for lf in 1 0; do
    out=$(IQD_D4_LEADFREE=$lf $B $args 2>/dev/null | grep '"metric"')
    echo "r$i leadfree=$lf [$args] $(echo "$out" | grep -o '"ms_per_step": [0-9.]*') $(echo "$out" | grep -o '"kernel_ms": [0-9.]*')"
  done
done
done 2>&1 | tee gpurun_out/r6/third_leadfree_ab.txt
