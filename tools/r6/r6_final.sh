#!/bin/bash
# round 6, final sources: the campaign's short-lead-in runs, every configuration's bench line from one box, the rocprofv3 summaries
mkdir -p gpurun_out/r6
{
run() { echo "## $*"; env "$@" 2>&1 | tail -1; }
run IQD_D4_LEADFREE=1 FUZZ_WIDE=1 FUZZ_WIDE_RANGE=24,600 timeout 140 python3 tools/gpu_fuzz.py 120 685
run IQD_D4_LEADFREE=1 IQD_STREAM_MIN_SEG=1 FUZZ_WIDE=1 FUZZ_WIDE_RANGE=24,600 timeout 140 python3 tools/gpu_fuzz.py 120 686
run IQD_D4_LEADFREE=2 FUZZ_WIDE=1 timeout 140 python3 tools/gpu_fuzz.py 120 687
run IQD_D4_LEADFREE=1 IQD_WBFM_PATH=stream timeout 110 python3 tools/gpu_fuzz.py 90 688
run IQD_D4_LEADFREE=0 FUZZ_WIDE=1 timeout 110 python3 tools/gpu_fuzz.py 90 689
} 2>&1 | tee gpurun_out/r6/fuzz_leadfree.txt
bash tools/bench_lines.sh 2>&1 | tail -20
cp gpurun_out/bench_lines.jsonl gpurun_out/r6/bench_lines.jsonl
ROUND=r6 bash tools/profile_all.sh 2>&1 | tail -12
