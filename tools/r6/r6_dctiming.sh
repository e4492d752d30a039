#!/bin/bash
# round 6: where a row's DC pass goes (tools/variant.sh dctiming iqd_kernels.hip -DIQD_DC_TIMING=1): phases in shader clocks, a few rows per launch
mkdir -p gpurun_out/r6
for M in "--mode am --channels 4096 --log2-samples 16" "--config 4"; do
  echo "== $M"
  IQD_LIB=$PWD/tmp_variants/lib_dctiming.so python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-live-pmc --no-from-idle --no-other-configs $M 2>&1 | grep "dc row" | tail -26
done > gpurun_out/r6/dc_timing.txt 2>&1
cat gpurun_out/r6/dc_timing.txt | cut -c1-260
