#!/bin/bash
# Round 5, tenth GPU session: what does v_cndmask_b32 with a scalar-pair mask cost (tools/ubench/classes: 12.6 cycles with vcc)?
# The P waves' lane-constant selections as v_bfi_b32 with a mask register (IQD_ST_BFI_SELECT) against v_cndmask_b32.
mkdir -p gpurun_out
{
  ./tmp_variants/classes | grep -E "opcode|cndmask|bfi_b32|add_f32 |mov_b32"
  echo "## WBFM 4096 x 2^16: selections by v_cndmask_b32 / by v_bfi_b32"
  bash tools/abn.sh 5 "" tmp_variants/lib_lit2.so tmp_variants/lib_bfi.so
  echo "## parity of the v_bfi_b32 build"
  IQD_LIB=$PWD/tmp_variants/lib_bfi.so timeout 900 python3 -m pytest tests/test_gpu_wbfm.py tests/test_gpu_stream.py -q -x -m gpu 2>&1 | tail -3
} > gpurun_out/r5_tenth.txt 2>&1
grep -v "^round" gpurun_out/r5_tenth.txt | tail -20
