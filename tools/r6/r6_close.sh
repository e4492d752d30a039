#!/bin/bash
# round 6: an FM call on the tile kernels closes its own step (fm_chain_kernel, last workgroup of a channel) - parity, then the A/B
mkdir -p gpurun_out/r6
( time python3 -m pytest tests/test_gpu_bench_paths.py tests/test_gpu_boundary.py tests/test_gpu_modes.py tests/test_gpu_gain_epochs.py tests/test_gpu_scale.py -x -q ) > gpurun_out/r6/close_tests.log 2>&1
tail -5 gpurun_out/r6/close_tests.log
{
echo "== configs[0] (1 FM channel, one 32 768-byte block per call)"
bash tools/abenv.sh 5 "--config 0" "-" "IQD_NO_CLOSE_IN_CHAIN=1"
echo "== 64 FM channels x 2^16"
bash tools/abenv.sh 5 "--channels 64 --log2-samples 16 --mode fm" "-" "IQD_NO_CLOSE_IN_CHAIN=1"
echo "== 512 FM channels x 2^16"
bash tools/abenv.sh 5 "--channels 512 --log2-samples 16 --mode fm" "-" "IQD_NO_CLOSE_IN_CHAIN=1"
echo "== 2048 FM channels x 2^13"
bash tools/abenv.sh 5 "--channels 2048 --log2-samples 13 --mode fm" "-" "IQD_NO_CLOSE_IN_CHAIN=1"
} > gpurun_out/r6/close_ab.txt 2>&1
cat gpurun_out/r6/close_ab.txt
