#!/bin/bash
# Every bench configuration's line from ONE box, into gpurun_out/bench_lines.jsonl (copy it to profiles/<round>_bench_lines.jsonl):
# the five BASELINE configurations, the single-family AM / USB runs at 4096 x 2^16, the reference's own operating point -
# one 64 ms block per channel per call (--log2-samples 14: DataConsumer.cc:342) - for configs 2 / 3 and AM / USB (with the
# kernels that ran), and the default configuration once more without the clock-settle phase.
OUT=gpurun_out/bench_lines.jsonl
: > $OUT
run() { python3 bench.py --no-host-path --no-live-pmc --cpu-one-core-only "$@" 2>>gpurun_out/bench_lines.err | grep '"metric"' >> $OUT; }
python3 bench.py --cpu-one-core-only 2>>gpurun_out/bench_lines.err | grep '"metric"' >> $OUT     # default: live counters, host path, CPU baseline
run --config 0
run --config 2 --no-cpu-baseline
run --config 3 --no-cpu-baseline
run --config 4 --no-cpu-baseline
run --mode am --channels 4096 --log2-samples 16 --no-cpu-baseline
run --mode usb --channels 4096 --log2-samples 16 --no-cpu-baseline
run --config 2 --log2-samples 14 --no-cpu-baseline
run --config 3 --log2-samples 14 --no-cpu-baseline
run --mode am --channels 4096 --log2-samples 14 --no-cpu-baseline
run --mode usb --channels 4096 --log2-samples 14 --no-cpu-baseline
IQD_WBFM_PATH=stream run --mode am --channels 4096 --log2-samples 14 --no-cpu-baseline
IQD_WBFM_PATH=stream run --mode usb --channels 4096 --log2-samples 14 --no-cpu-baseline
run --mode mixed --channels 16384 --log2-samples 14 --no-cpu-baseline
run --mode mixed --channels 8192 --log2-samples 16 --no-cpu-baseline
run --mode mixed --channels 1400 --log2-samples 16 --no-cpu-baseline
run --no-cpu-baseline --prewarm-ms 0
python3 - <<'PY'
import json
for l in open("gpurun_out/bench_lines.jsonl"):
    d = json.loads(l)
    print("%-78s ms/step %-8s frac %-7s %s" % (d["config"]["workload"][:78], d["ms_per_step"], d["roofline"].get("frac"), d["config"].get("kernels", "")[:60]))
PY
