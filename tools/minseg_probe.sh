#!/bin/bash
# Where the streaming pipelines overtake the tile kernels (run on the GPU box): IQD_STREAM_MIN_SEG = samples a launch must
# bring per segment of the persistent workgroups before it takes them; 1 = whenever the geometry allows, 100000 = never,
# unset = the engine's own thresholds (iqd_engine.cpp: STREAM_MIN_SEG_*), which should pick the faster of the two.
run() {  # mode channels log2
  line="mode $1 ch $2 log2 $3"
  for m in auto 1 100000; do
    if [ $m = auto ]; then unset IQD_STREAM_MIN_SEG; else export IQD_STREAM_MIN_SEG=$m; fi
    out=$(python3 bench.py --mode $1 --channels $2 --log2-samples $3 --steps 20 --warmup 3 --prewarm-ms 30 --no-cpu-baseline --no-host-path --no-live-pmc 2>/dev/null)
    s=$(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    k=$(echo "$out" | grep -o '"kernels": "[^"]*"' | head -1 | cut -c13-17)
    line="$line | $m $s $k"
  done
  echo "$line"
}
if [ "$1" = full ]; then
  for mode in fm am usb wbfm; do
    for cfg in "128 16" "256 16" "512 16" "1024 16" "512 14" "1024 14" "2048 14" "2048 13" "4096 13"; do run $mode $cfg; done
  done
  for cfg in "1 22" "1 23" "1 24" "1 25" "1 26"; do run wbfm $cfg; done
  for cfg in "256 16" "512 16" "1024 16" "512 14" "1024 14" "1024 15" "2048 13" "4096 12" "4096 13"; do run mixed $cfg; done
else
  for cfg in "16 16" "64 16" "128 16" "128 14" "256 14" "512 12" "1024 12" "64 18" "5 20"; do run mixed $cfg; done
  for cfg in "768 16" "3072 13" "640 16"; do run fm $cfg; done
  for cfg in "384 16" "1536 14" "1 25"; do run wbfm $cfg; done
  for cfg in "768 16" "896 16" "768 14" "1536 13"; do run am $cfg; run usb $cfg; done
fi
