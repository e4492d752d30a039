#!/bin/bash
run() { echo "## $*"; env "$@" 2>&1 | tail -1 | cut -c1-400; }
run IQD_WBFM_PATH=stream timeout 150 python3 tools/gpu_fuzz.py 125 375
run IQD_WBFM_PATH=stream FUZZ_SHORT=1 timeout 90 python3 tools/gpu_fuzz.py 65 376
run IQD_WBFM_PATH=stream FUZZ_WIDE=1 FUZZ_WIDE_RANGE=24,600 timeout 100 python3 tools/gpu_fuzz.py 75 377
run timeout 130 python3 tools/gpu_fuzz.py 105 371
run FUZZ_WIDE=1 timeout 100 python3 tools/gpu_fuzz.py 75 372
run IQD_STREAM_MIN_SEG=1 timeout 100 python3 tools/gpu_fuzz.py 75 384
run IQD_STREAM_MIN_SEG=1 FUZZ_SHORT=1 timeout 90 python3 tools/gpu_fuzz.py 65 383
