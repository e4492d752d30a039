#!/bin/bash
# round 6, last pass: the streaming tests with the short lead-ins forced everywhere (IQD_D4_LEADFREE=1: wherever a family streams
# as a kernel of its own; =2: inside the one launch for several families too), then the round's profiles from the final library
mkdir -p gpurun_out/r6
for L in 1 2; do
  ( time IQD_D4_LEADFREE=$L python3 -m pytest tests/test_gpu_stream2.py tests/test_gpu_scale.py tests/test_gpu_bench_paths.py tests/test_gpu_gain_epochs.py tests/test_gpu_fuzz_pins.py -m gpu -x -q ) > gpurun_out/r6/leadfree${L}_tests.log 2>&1
  echo "IQD_D4_LEADFREE=$L: $(grep -E 'passed|failed' gpurun_out/r6/leadfree${L}_tests.log | tail -1)"
done
ROUND=r6 bash tools/profile_all.sh 2>&1 | tail -8
