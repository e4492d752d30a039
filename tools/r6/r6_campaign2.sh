#!/bin/bash
# round 6: a second campaign with other seeds on the final library, and the streaming test files once more
mkdir -p gpurun_out/r6
( python3 -m pytest tests/test_gpu_stream2.py tests/test_gpu_modes.py tests/test_gpu_bench_paths.py -x -q 2>&1 | tail -2 )
bash tools/fuzz_campaign.sh 800 2>&1 | tee gpurun_out/r6/fuzz_campaign2.txt
