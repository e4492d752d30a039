#!/bin/bash
# round 6, first box: where a piece of the AM / USB / FM pipelines goes (in-kernel timing builds), and this box's baseline lines
mkdir -p gpurun_out/r6
B="python3 bench.py --no-host-path --no-live-pmc --no-cpu-baseline --no-from-idle"
for m in am usb fm; do
  IQD_LIB=$PWD/tmp_variants/lib_d4timing.so $B --mode $m --channels 4096 --log2-samples 16 --steps 2 --warmup 1 --prewarm-ms 30 > gpurun_out/r6/timing1_$m.txt 2>&1
done
IQD_LIB=$PWD/tmp_variants/lib_d4timing2.so $B --mode am --channels 4096 --log2-samples 16 --steps 2 --warmup 1 --prewarm-ms 30 > gpurun_out/r6/timing2_am.txt 2>&1
IQD_LIB=$PWD/tmp_variants/lib_d4waitstat.so $B --mode am --channels 4096 --log2-samples 16 --steps 2 --warmup 1 --prewarm-ms 30 > gpurun_out/r6/waitstat_am.txt 2>&1
for f in gpurun_out/r6/timing1_*.txt gpurun_out/r6/timing2_am.txt gpurun_out/r6/waitstat_am.txt; do echo "== $f: $(wc -l < $f) lines"; tail -n 64 $f > $f.tail; mv $f.tail $f; done
# baseline lines of this box
for args in "--mode am --channels 4096 --log2-samples 16" "--mode usb --channels 4096 --log2-samples 16" "--config 2" "--config 3" "--config 4" \
            "--config 2 --log2-samples 14" "--config 3 --log2-samples 14" "--mode am --channels 4096 --log2-samples 14" "--mode usb --channels 4096 --log2-samples 14" ""; do
  out=$($B $args --steps 40 --warmup 5 2>/dev/null | grep '"metric"')
  echo "base [$args] $(echo "$out" | grep -o '"ms_per_step": [0-9.]*') $(echo "$out" | grep -o '"kernel_ms": [0-9.]*')"
done 2>&1 | tee gpurun_out/r6/baseline_lines.txt
# per-kernel times of the AM step and of configs[4]
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6/kt_am -- $B --mode am --channels 4096 --log2-samples 16 --steps 20 --warmup 3 > gpurun_out/r6/kt_am.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6/kt_c4 -- $B --config 4 --steps 20 --warmup 3 > gpurun_out/r6/kt_c4.log 2>&1
for d in kt_am kt_c4; do f=$(find gpurun_out/r6/$d -name '*kernel_stats.csv' | head -1); echo "== $d"; head -8 $f; cp $f gpurun_out/r6/${d}_kernel_stats.csv; find gpurun_out/r6/$d -name '*.csv' ! -name '*stats*' -delete; done
