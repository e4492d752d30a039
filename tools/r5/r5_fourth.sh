#!/bin/bash
# round 5, fourth box: the software-pipelined P wave (IQD_ST_SWP=1) - correctness, then A/B; tile kernels against the streaming
# pipelines with the rings rule around the thresholds
mkdir -p gpurun_out
( IQD_LIB=$PWD/tmp_variants/lib_swp1.so timeout 900 python3 -m pytest tests/test_gpu_stream.py tests/test_gpu_wbfm.py tests/test_gpu_gain_epochs.py tests/test_gpu_scale.py -x -q ) > gpurun_out/r5_swp_tests.log 2>&1
echo "swp1 tests: $(tail -1 gpurun_out/r5_swp_tests.log)"
( IQD_LIB=$PWD/tmp_variants/lib_swp1.so IQD_WBFM_PATH=stream timeout 100 python3 tools/gpu_fuzz.py 60 911 ) 2>&1 | tail -1
tools/abn.sh 4 "" tmp_variants/lib_base.so tmp_variants/lib_swp1.so 2>&1 | grep median | tee gpurun_out/r5_ab_swp.log
tools/abn.sh 2 "--signal white" tmp_variants/lib_base.so tmp_variants/lib_swp1.so 2>&1 | grep median | tee -a gpurun_out/r5_ab_swp.log
tools/abn.sh 2 "--config 3" tmp_variants/lib_base.so tmp_variants/lib_swp1.so 2>&1 | grep median | tee -a gpurun_out/r5_ab_swp.log
# the whole streaming test files under the new default (rings rule)
python3 -m pytest tests/test_gpu_stream.py tests/test_gpu_stream2.py tests/test_gpu_bench_paths.py -x -q 2>&1 | tail -1
B="python3 bench.py --no-host-path --no-live-pmc --no-cpu-baseline --no-from-idle --steps 40 --warmup 5"
for shape in "256 16" "512 16" "768 16" "1024 14" "2048 14" "3072 14" "4096 14"; do
  set -- $shape
  for m in fm am usb wbfm; do
    line="$m ${1}x2^${2}:"
    for p in default tiles stream; do
      if [ $p = default ]; then out=$($B --mode $m --channels $1 --log2-samples $2 2>/dev/null | grep '"metric"')
      else out=$(IQD_WBFM_PATH=$p $B --mode $m --channels $1 --log2-samples $2 2>/dev/null | grep '"metric"'); fi
      line="$line  $p $(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2)"
    done
    echo "$line"
  done
done 2>&1 | tee gpurun_out/r5_threshold_probe.log
