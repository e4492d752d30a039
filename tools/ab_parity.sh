#!/bin/bash
# On the GPU box: parity of a variant build (WBFM + streaming test files), then - only if it holds - interleaved timing against a base build.
#   tools/ab_parity.sh <base.so> <variant.so> ["<bench.py arguments>"] ["<pytest files>"]
BASE=$1; VAR=$2; ARGS=${3:-}; FILES=${4:-tests/test_gpu_wbfm.py tests/test_gpu_stream.py}
mkdir -p gpurun_out
{
  echo "## parity of $VAR"
  IQD_LIB=$PWD/$VAR timeout 900 python3 -m pytest $FILES -q -x -m gpu 2>&1 | tail -3 | tee /tmp/parity.txt
  if grep -q " passed" /tmp/parity.txt && ! grep -q failed /tmp/parity.txt; then
    echo "## bench.py $ARGS: $BASE / $VAR"
    bash tools/abn.sh 5 "$ARGS" $BASE $VAR
  fi
} > gpurun_out/ab_parity.txt 2>&1
grep -v "^round" gpurun_out/ab_parity.txt | tail -12
