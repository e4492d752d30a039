#!/bin/bash
# Round 5, eighth GPU session: wave priorities and issue rates (tools/ubench/prio.hip); the fused convert-and-pack of the IIR waves
# (cast_pack_i16_bounded) against the plain one; idle instructions added to the P waves of ONE SIMD (is the fourth SIMD, which
# carries no IIR wave, the one with issue slots to spare?).  Run on the GPU box from the repo root.
mkdir -p gpurun_out
{
  ./tmp_variants/prio
  echo "## WBFM 4096 x 2^16: plain conversions / fused / 64 idle v_add_f32 per piece in the P waves of SIMD 3 / of SIMD 0"
  bash tools/abn.sh 5 "" tmp_variants/lib_nocvt.so tmp_variants/lib_cvt.so tmp_variants/lib_burn3.so tmp_variants/lib_burn0.so
  echo "## parity of the fused build"
  IQD_LIB=$PWD/tmp_variants/lib_cvt.so timeout 900 python3 -m pytest tests/test_gpu_wbfm.py tests/test_gpu_stream.py -q -x -m gpu 2>&1 | tail -3
} > gpurun_out/r5_eighth.txt 2>&1
tail -30 gpurun_out/r5_eighth.txt
