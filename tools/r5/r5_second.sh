#!/bin/bash
# round 5, second box: the plan-then-execute engine under the whole -m gpu tier, the opcode-class microbenchmark, bench lines
mkdir -p gpurun_out
( time python3 -m pytest tests -m gpu -x -q ) > gpurun_out/r5_gputests2.log 2>&1
tail -4 gpurun_out/r5_gputests2.log
tools/ubench/classes > gpurun_out/r5_classes.log 2>&1
cat gpurun_out/r5_classes.log
B="python3 bench.py --no-host-path --no-live-pmc --no-cpu-baseline --steps 40 --warmup 5"
for args in "" "--config 2" "--config 3" "--config 4" "--mode am --channels 4096 --log2-samples 16" "--config 3 --log2-samples 14"; do
  out=$($B $args 2>/dev/null | grep '"metric"')
  echo "[$args] $(echo "$out" | grep -o '"ms_per_step": [0-9.]*') $(echo "$out" | grep -o '"from_idle_ms_per_step": [0-9.]*') $(echo "$out" | grep -o '"kernel_ms": [0-9.]*')"
done 2>&1 | tee gpurun_out/r5_lines2.log
# what the north-star's stated +-1 LSB tolerance would buy (a timing A/B only)
cp rtlsdrdiags_amd/libiqdemod.so tmp_variants/lib_base.so
tools/abn.sh 3 "" tmp_variants/lib_base.so tmp_variants/lib_relaxed.so 2>&1 | tee gpurun_out/r5_ab_relaxed.log
