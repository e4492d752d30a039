#!/bin/bash
# round 6, the last numbers from the final library on ONE box: every bench line, the driver's own command, the round's profiles
mkdir -p gpurun_out/r6
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/bench_lines.sh 2>&1 | tail -20
( time python3 bench.py --gpus 1 ) > gpurun_out/r6/final_bench.json 2> gpurun_out/r6/final_bench.err
tail -3 gpurun_out/r6/final_bench.err
ROUND=r6 bash tools/profile_all.sh 2>&1 | tail -8
