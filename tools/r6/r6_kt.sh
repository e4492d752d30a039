#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats) of a list of bench invocations: tools/r6/r6_kt.sh <tag> <env assignments or -> <bench args...>
mkdir -p gpurun_out/r6
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=$1; ENVS=$2; shift 2
B="python3 bench.py --no-host-path --no-live-pmc --no-cpu-baseline --no-from-idle --steps 20 --warmup 3 $*"
if [ "$ENVS" != "-" ]; then export $ENVS; fi
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6/kt_$TAG -- $B > gpurun_out/r6/kt_$TAG.log 2>&1
f=$(find gpurun_out/r6/kt_$TAG -name '*kernel_stats.csv' | head -1)
echo "== $TAG ($ENVS) $*"; grep -v "at::native\|rocclr" $f | cut -c1-160 | head -8
cp $f gpurun_out/r6/kt_${TAG}_kernel_stats.csv; rm -rf gpurun_out/r6/kt_$TAG
