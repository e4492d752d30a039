#!/bin/bash
# The fuzzer over the engine's path-pinning knobs on the final sources (run on the GPU box; about 25 minutes).
run() { echo "## $*"; env "$@" 2>&1 | tail -1; }
run timeout 260 python3 tools/gpu_fuzz.py 240 171
run FUZZ_WIDE=1 timeout 200 python3 tools/gpu_fuzz.py 180 172
run FUZZ_WIDE=1 FUZZ_WIDE_RANGE=24,600 timeout 170 python3 tools/gpu_fuzz.py 150 173
run FUZZ_SHORT=1 timeout 140 python3 tools/gpu_fuzz.py 120 174
run IQD_WBFM_PATH=stream timeout 200 python3 tools/gpu_fuzz.py 180 175
run IQD_WBFM_PATH=stream FUZZ_SHORT=1 timeout 110 python3 tools/gpu_fuzz.py 90 176
run IQD_WBFM_PATH=tiles timeout 140 python3 tools/gpu_fuzz.py 120 177
run IQD_MIXED=forked FUZZ_WIDE=1 timeout 140 python3 tools/gpu_fuzz.py 120 178
run IQD_MIXED=forked FUZZ_WIDE=1 FUZZ_WIDE_RANGE=24,600 timeout 110 python3 tools/gpu_fuzz.py 90 179
run IQD_SHARES=cost FUZZ_WIDE=1 timeout 110 python3 tools/gpu_fuzz.py 90 180
run IQD_STREAM_MIN_SEG=1 FUZZ_WIDE=1 FUZZ_WIDE_RANGE=24,600 timeout 110 python3 tools/gpu_fuzz.py 90 181
run FUZZ_BIG=1 timeout 110 python3 tools/gpu_fuzz.py 90 182
run IQD_STREAM_MIN_SEG=1 FUZZ_SHORT=1 timeout 110 python3 tools/gpu_fuzz.py 90 183
run IQD_STREAM_MIN_SEG=1 timeout 110 python3 tools/gpu_fuzz.py 90 184
