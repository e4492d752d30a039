#!/bin/bash
# Round profiles of the bench configurations (run on the GPU box from the repo root).  For each: the rocprofv3 kernel
# trace + the PMC passes of tools/profile.sh, condensed by tools/pmc_summary.py; the summary and the kernel stats are
# copied to profiles/<tag>_pmc.json and profiles/<tag>_kernel_stats.csv (copy them back from gpurun_out/profiles_out/).
set -u
R=${ROUND:-r5}
OUT=gpurun_out/profiles_out
mkdir -p $OUT
run() { # tag needle samples args...
  local tag=$1 needle=$2 samples=$3; shift 3
  tools/profile.sh $tag --no-live-pmc "$@" > /dev/null
  python3 tools/pmc_summary.py gpurun_out/$tag "$needle" $samples $OUT/${tag}_pmc.json > /dev/null
  cp "$(ls -t gpurun_out/$tag/kt/*/*kernel_stats.csv | head -1)" $OUT/${tag}_kernel_stats.csv
  python3 - <<PY
import json; o=json.load(open("$OUT/${tag}_pmc.json"))
print("$tag", o.get("kernel","?")[:60], "ms", round(o.get("kernel_ms_avg",0),4), "ops/sample", round(o["derived"].get("valu_lane_ops_per_sample",0),1),
      "valu_frac", round(o["derived"].get("valu_issue_frac",0),3), "traffic/algo", round(o["derived"].get("traffic_over_algorithmic",0),3), "clk", o.get("clock_ghz"), o.get("code_object",{}).get("vgpr_count"))
PY
}
run ${R}_wbfm_2p28 wbfm_stream_kernel 268435456
run ${R}_wbfm_2p28_white wbfm_stream_kernel 268435456 --signal white
run ${R}_fm_4096 d4_stream_kernel 268435456 --config 2
run ${R}_am_4096 d4_stream_kernel 268435456 --mode am --channels 4096 --log2-samples 16
run ${R}_usb_4096 d4_stream_kernel 268435456 --mode usb --channels 4096 --log2-samples 16
run ${R}_ssb_8192 d4_stream_kernel 536870912 --config 4
# the mixed configuration: the four families' pipelines as workgroup ranges of one launch
run ${R}_mixed_4096 mixed_stream_kernel 268435456 --config 3
