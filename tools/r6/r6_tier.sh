#!/bin/bash
# round 6: smoke() and the whole -m gpu tier on the final tree, then the driver's own bench command
mkdir -p gpurun_out/r6
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python3 -m pytest tests -m gpu -x -q ) > gpurun_out/r6/final_gputests.log 2>&1
tail -6 gpurun_out/r6/final_gputests.log
( time python3 bench.py --gpus 1 ) > gpurun_out/r6/final_bench.json 2> gpurun_out/r6/final_bench.err
tail -3 gpurun_out/r6/final_bench.err
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6/final_bench.json") if l.startswith('{"metric"')][-1])
print("headline", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("traffic_over_algorithmic"))
for s in d.get("other_configs", []):
    r = s.get("roofline", {})
    print("  ", s.get("argv"), s.get("ms_per_step"), r.get("frac"), r.get("kernel_ms"), r.get("traffic_over_algorithmic"), s.get("seconds"), s.get("error"))
a = d["cpu_baseline"].get("all_cores", {})
print("cpu", d["cpu_baseline"]["value"], a.get("value"), a.get("cores"), a.get("cores_effective"), a.get("cores_worth"), a.get("knee_processes"), a.get("host"))
print("series", [(e["processes"], e["value"]) for e in a.get("series", [])])
PY
