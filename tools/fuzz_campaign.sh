#!/bin/bash
# The fuzzer over the engine's path-pinning knobs on the final sources (run on the GPU box; about 25 minutes).
OFF=${1:-0}   # added to every seed: a second campaign draws other cases
MAX=${2:-99}  # only the first MAX runs (a shorter campaign)
N=0
run() { N=$((N + 1)); [ $N -gt $MAX ] && return; echo "## $*"; env "$@" 2>&1 | tail -1; }
run timeout 260 python3 tools/gpu_fuzz.py 240 $((71 + OFF))
run FUZZ_WIDE=1 timeout 200 python3 tools/gpu_fuzz.py 180 $((72 + OFF))
run FUZZ_WIDE=1 FUZZ_WIDE_RANGE=24,600 timeout 170 python3 tools/gpu_fuzz.py 150 $((73 + OFF))
run FUZZ_SHORT=1 timeout 140 python3 tools/gpu_fuzz.py 120 $((74 + OFF))
run IQD_WBFM_PATH=stream timeout 200 python3 tools/gpu_fuzz.py 180 $((75 + OFF))
run IQD_WBFM_PATH=stream FUZZ_SHORT=1 timeout 110 python3 tools/gpu_fuzz.py 90 $((76 + OFF))
run IQD_WBFM_PATH=tiles timeout 140 python3 tools/gpu_fuzz.py 120 $((77 + OFF))
run IQD_MIXED=forked FUZZ_WIDE=1 timeout 140 python3 tools/gpu_fuzz.py 120 $((78 + OFF))
run IQD_MIXED=forked FUZZ_WIDE=1 FUZZ_WIDE_RANGE=24,600 timeout 110 python3 tools/gpu_fuzz.py 90 $((79 + OFF))
run IQD_SHARES=cost FUZZ_WIDE=1 timeout 110 python3 tools/gpu_fuzz.py 90 $((80 + OFF))
run IQD_STREAM_MIN_SEG=1 FUZZ_WIDE=1 FUZZ_WIDE_RANGE=24,600 timeout 110 python3 tools/gpu_fuzz.py 90 $((81 + OFF))
run FUZZ_BIG=1 timeout 110 python3 tools/gpu_fuzz.py 90 $((82 + OFF))
run IQD_STREAM_MIN_SEG=1 FUZZ_SHORT=1 timeout 110 python3 tools/gpu_fuzz.py 90 $((83 + OFF))
run IQD_STREAM_MIN_SEG=1 timeout 110 python3 tools/gpu_fuzz.py 90 $((84 + OFF))
# round 6: short lead-ins forced wherever a family streams / inside the one launch too / off (by default: channels of >= 8 segments)
run IQD_D4_LEADFREE=1 FUZZ_WIDE=1 FUZZ_WIDE_RANGE=24,600 timeout 140 python3 tools/gpu_fuzz.py 120 $((85 + OFF))
run IQD_D4_LEADFREE=1 IQD_STREAM_MIN_SEG=1 FUZZ_WIDE=1 FUZZ_WIDE_RANGE=24,600 timeout 140 python3 tools/gpu_fuzz.py 120 $((86 + OFF))
run IQD_D4_LEADFREE=2 FUZZ_WIDE=1 timeout 140 python3 tools/gpu_fuzz.py 120 $((87 + OFF))
run IQD_D4_LEADFREE=1 IQD_WBFM_PATH=stream timeout 110 python3 tools/gpu_fuzz.py 90 $((88 + OFF))
run IQD_D4_LEADFREE=0 FUZZ_WIDE=1 timeout 110 python3 tools/gpu_fuzz.py 90 $((89 + OFF))
