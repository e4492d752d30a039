#!/bin/bash
# round 5, seventh box: the AM / SSB pipelines as compact 10-wave workgroups, two to a CU (IQD_COMPACT=1 always, 0 never): correctness, then timing
mkdir -p gpurun_out
( IQD_COMPACT=1 timeout 900 python3 -m pytest tests/test_gpu_stream2.py tests/test_gpu_bench_paths.py tests/test_gpu_modes.py -x -q ) > gpurun_out/r5_compact_tests.log 2>&1
echo "IQD_COMPACT=1 tests: $(tail -1 gpurun_out/r5_compact_tests.log)"
( IQD_COMPACT=1 IQD_WBFM_PATH=stream FUZZ_WIDE=1 timeout 100 python3 tools/gpu_fuzz.py 60 921 ) 2>&1 | tail -1
( IQD_COMPACT=1 IQD_WBFM_PATH=stream timeout 100 python3 tools/gpu_fuzz.py 45 922 ) 2>&1 | tail -1
B="python3 bench.py --no-host-path --no-live-pmc --no-cpu-baseline --no-from-idle --steps 40 --warmup 5"
for rep in 1 2; do
for args in "--mode am --channels 4096 --log2-samples 16" "--mode usb --channels 4096 --log2-samples 16" "--config 4" "--mode am --channels 8192 --log2-samples 16" \
            "--mode am --channels 4096 --log2-samples 14" "--mode usb --channels 4096 --log2-samples 14" "--mode am --channels 16384 --log2-samples 14" "--mode am --channels 2048 --log2-samples 16"; do
  line="[$args]"
  for c in 0 1; do
    out=$(IQD_COMPACT=$c $B $args 2>/dev/null | grep '"metric"')
    line="$line  compact=$c $(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2) ($(echo "$out" | grep -o '"kernel_ms": [0-9.]*' | cut -d' ' -f2))"
  done
  echo "$line"
done
done 2>&1 | tee gpurun_out/r5_compact_probe.log
