#!/bin/bash
# round 6: a third fuzz campaign, on the final library (per-family tile lead-ins and tails, write-through closing stores)
mkdir -p gpurun_out/r6
bash tools/fuzz_campaign.sh 1200 > gpurun_out/r6/fuzz3.txt 2>&1
grep -c identical gpurun_out/r6/fuzz3.txt; grep -v "identical\|^##" gpurun_out/r6/fuzz3.txt | head
python3 - <<'PY'
import re
t = open("gpurun_out/r6/fuzz3.txt").read()
print("cases:", sum(int(x) for x in re.findall(r"gpu_fuzz: (\d+) ", t)))
PY
