#!/bin/bash
# kernel-trace of one bench.py invocation with a given library: tools/kt.sh <lib.so> <tag> [bench args]
L=$1; TAG=$2; shift 2
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
export IQD_LIB=$PWD/$L
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -- python3 bench.py --no-cpu-baseline --no-host-path --steps 20 --warmup 3 "$@" > gpurun_out/$TAG.log 2>&1
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob("gpurun_out/$TAG/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    v.sort(); print("%-62s n=%3d avg %.4f ms  med %.4f  min %.4f" % (k, len(v), sum(v)/len(v)/1e6, v[len(v)//2]/1e6, v[0]/1e6))
PY
