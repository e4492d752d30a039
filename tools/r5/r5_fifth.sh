#!/bin/bash
# round 5, fifth box: the P wave with its float chain lagging a piece (IQD_ST_SWP), instantiation without magnitudes (the one that fits the registers)
mkdir -p gpurun_out
for L in base swp3; do IQD_LIB=$PWD/tmp_variants/lib_$L.so timeout 300 python3 tools/swp_check.py 2>&1 | tail -1; done | tee gpurun_out/r5_swp3.log
tools/abn.sh 4 "--no-magnitude" tmp_variants/lib_base.so tmp_variants/lib_swp3.so 2>&1 | grep median | tee -a gpurun_out/r5_swp3.log
