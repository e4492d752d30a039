#!/bin/bash
# The fuzzer over the engine's path-pinning knobs on the final sources (run on the GPU box; about 25 minutes).
run() { echo "## $*"; env "$@" 2>&1 | tail -1; }
run timeout 169 python3 tools/gpu_fuzz.py 144 271
run FUZZ_WIDE=1 timeout 133 python3 tools/gpu_fuzz.py 108 272
run FUZZ_WIDE=1 FUZZ_WIDE_RANGE=24,600 timeout 115 python3 tools/gpu_fuzz.py 90 273
run FUZZ_SHORT=1 timeout 97 python3 tools/gpu_fuzz.py 72 274
run IQD_WBFM_PATH=stream timeout 133 python3 tools/gpu_fuzz.py 108 275
run IQD_WBFM_PATH=stream FUZZ_SHORT=1 timeout 79 python3 tools/gpu_fuzz.py 54 276
run IQD_WBFM_PATH=tiles timeout 97 python3 tools/gpu_fuzz.py 72 277
run IQD_MIXED=forked FUZZ_WIDE=1 timeout 97 python3 tools/gpu_fuzz.py 72 278
run IQD_MIXED=forked FUZZ_WIDE=1 FUZZ_WIDE_RANGE=24,600 timeout 79 python3 tools/gpu_fuzz.py 54 279
run IQD_SHARES=cost FUZZ_WIDE=1 timeout 79 python3 tools/gpu_fuzz.py 54 280
run IQD_STREAM_MIN_SEG=1 FUZZ_WIDE=1 FUZZ_WIDE_RANGE=24,600 timeout 79 python3 tools/gpu_fuzz.py 54 281
run FUZZ_BIG=1 timeout 79 python3 tools/gpu_fuzz.py 54 282
run IQD_STREAM_MIN_SEG=1 FUZZ_SHORT=1 timeout 79 python3 tools/gpu_fuzz.py 54 283
run IQD_STREAM_MIN_SEG=1 timeout 79 python3 tools/gpu_fuzz.py 54 284
