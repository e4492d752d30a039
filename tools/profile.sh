#!/bin/bash
# rocprofv3 passes for one bench.py invocation (run on the GPU box, from the repo root):
#   tools/profile.sh <tag> [bench.py arguments ...]
# Writes gpurun_out/<tag>/{kt,fetch,write,sq1,sq2,sq3}; tools/pmc_summary.py condenses them into profiles/<tag>.json.
# Counters go in their own passes, with no trace domain besides the kernel trace (MI355X_MICROARCH.md, PMC slots).
set -u
TAG=$1; shift
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
B="python3 bench.py --no-cpu-baseline --no-host-path --no-live-pmc $*"
P="$B --prewarm-ms 0"     # counter passes: a few launches are enough (the kernel trace keeps bench.py's clock-settle phase: settled durations)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B --steps 20 --warmup 3 > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $P --steps 3 --warmup 1 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $P --steps 3 --warmup 1 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --output-format csv -d $OUT/sq1 -- $P --steps 3 --warmup 1 > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY \
    --output-format csv -d $OUT/sq2 -- $P --steps 3 --warmup 1 > $OUT/sq2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/sq3 -- $P --steps 3 --warmup 1 > $OUT/sq3.log 2>&1
grep -h '"metric"' $OUT/kt.log | tail -1 | cut -c1-600
