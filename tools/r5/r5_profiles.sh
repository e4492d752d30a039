#!/bin/bash
# round 5: the round's profiles and bench lines from one box (copy gpurun_out/profiles_out/* and bench_lines.jsonl to profiles/r5_*)
mkdir -p gpurun_out
ROUND=r5 tools/profile_all.sh 2>&1 | tee gpurun_out/r5_profile_all.log
tools/bench_lines.sh 2>&1 | tee gpurun_out/r5_bench_lines.log
