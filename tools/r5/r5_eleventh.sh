#!/bin/bash
# Round 5, eleventh GPU session: the first FIR's two byte planes as a CHAIN of matrix instructions (low bytes, >> 8, high bytes on top:
# IQD_ST_CHAINED_MFMA) against the two independent products combined by v_lshl_add_u32.  Parity first; timing only if it holds.
mkdir -p gpurun_out
{
  echo "## parity of the chained build"
  IQD_LIB=$PWD/tmp_variants/lib_chain.so timeout 900 python3 -m pytest tests/test_gpu_wbfm.py tests/test_gpu_stream.py -q -x -m gpu 2>&1 | tail -3 | tee /tmp/parity.txt
  if grep -q " passed" /tmp/parity.txt && ! grep -q failed /tmp/parity.txt; then
    echo "## WBFM 4096 x 2^16: v_lshl_add_u32 / chained"
    bash tools/abn.sh 5 "" tmp_variants/lib_lit2.so tmp_variants/lib_chain.so
  fi
} > gpurun_out/r5_eleventh.txt 2>&1
grep -v "^round" gpurun_out/r5_eleventh.txt | tail -20
