#!/bin/bash
# round 6, final sources: every configuration's bench line from one box, then the rocprofv3 summaries (profiles/r6_*)
mkdir -p gpurun_out/r6
bash tools/bench_lines.sh 2>&1 | tail -20
cp gpurun_out/bench_lines.jsonl gpurun_out/r6/bench_lines.jsonl
ROUND=r6 bash tools/profile_all.sh 2>&1 | tail -12
