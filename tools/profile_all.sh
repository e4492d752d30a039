#!/bin/bash
# Round profiles of the bench configurations (run on the GPU box from the repo root); summaries land in
# gpurun_out/<tag>/summary.json and the kernel stats in gpurun_out/<tag>/kt/**/_kernel_stats.csv.
set -u
# (copy what is to be judged into profiles/: <tag>/summary.json -> profiles/<tag>_pmc.json, the newest kt/**/_kernel_stats.csv beside it)
run() { # tag needle samples args...
  local tag=$1 needle=$2 samples=$3; shift 3
  tools/profile.sh $tag "$@" > /dev/null
  python3 tools/pmc_summary.py gpurun_out/$tag "$needle" $samples gpurun_out/$tag/summary.json > /dev/null
  python3 - <<PY
import json; o=json.load(open("gpurun_out/$tag/summary.json"))
print("$tag", o.get("kernel","?")[:50], "ms", round(o.get("kernel_ms_avg",0),4), "ops/sample", round(o["derived"].get("valu_lane_ops_per_sample",0),1),
      "valu_frac", round(o["derived"].get("valu_issue_frac",0),3), "traffic/algo", round(o["derived"].get("traffic_over_algorithmic",0),3), "clk", o.get("clock_ghz"))
PY
}
run r2_wbfm_2p28 wbfm_stream_kernel 268435456
run r2_wbfm_2p28_tiles wbfm_chain_kernel 268435456 --wbfm-path tiles
run r2_wbfm_2p28_white wbfm_stream_kernel 268435456 --signal white
run r2_fm_4096 d4_stream_kernel 268435456 --config 2
run r2_am_4096 d4_stream_kernel 268435456 --mode am --channels 4096 --log2-samples 16
run r2_usb_4096 d4_stream_kernel 268435456 --mode usb --channels 4096 --log2-samples 16
run r2_ssb_8192 d4_stream_kernel 536870912 --config 4
run r2_mixed_4096 wbfm_stream_kernel 53673984 --config 3
