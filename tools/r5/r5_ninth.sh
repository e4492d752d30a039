#!/bin/bash
# Round 5, ninth GPU session: the IIR lanes' taps as instruction literals (lib_lit) against taps in scalar registers (lib_cvt);
# parity of the literal build.  Run on the GPU box from the repo root.
mkdir -p gpurun_out
{
  echo "## WBFM 4096 x 2^16: fused conversions, taps in scalar registers / as literals"
  bash tools/abn.sh 5 "" tmp_variants/lib_cvt.so tmp_variants/lib_lit.so
  echo "## parity of the literal build"
  IQD_LIB=$PWD/tmp_variants/lib_lit.so timeout 900 python3 -m pytest tests/test_gpu_wbfm.py tests/test_gpu_stream.py tests/test_gpu_modes.py -q -x -m gpu 2>&1 | tail -3
} > gpurun_out/r5_ninth.txt 2>&1
grep -v "^round" gpurun_out/r5_ninth.txt | tail
