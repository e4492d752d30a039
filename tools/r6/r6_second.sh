#!/bin/bash
# round 6, second box: what the AM / USB / FM pipelines would take with their loads served from L2 (timing probe, wrong PCM),
# and with the youngest P wave of every ring at a raised priority - interleaved on one box
mkdir -p gpurun_out/r6
for m in "--mode am --channels 4096 --log2-samples 16" "--mode usb --channels 4096 --log2-samples 16" "--config 2"; do
  echo "## $m"
  tools/abn.sh 3 "$m --no-from-idle" tmp_variants/lib_base.so tmp_variants/lib_cached.so tmp_variants/lib_prio1.so tmp_variants/lib_prio3.so
done 2>&1 | tee gpurun_out/r6/probe_cached_prio.txt
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/streams tools/ubench/streams.hip && /tmp/streams 2>&1 | tee gpurun_out/r6/streams_ubench.txt | tail -30
