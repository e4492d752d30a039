#!/bin/bash
# round 6: the fuzz campaign over every path pin on the round's final kernels (warm / cold segments, in-place DC pass)
mkdir -p gpurun_out/r6
bash tools/fuzz_campaign.sh 600 2>&1 | tee gpurun_out/r6/fuzz_campaign.txt
