#!/bin/bash
for m in usb; do
  bash tools/r6/r6_kt.sh ${m}_norec "IQD_D4_LEADFREE=1 IQD_D4_NOREC=1" --mode $m --channels 4096 --log2-samples 16
  bash tools/r6/r6_kt.sh ${m}_gran256 "IQD_D4_LEADFREE=1 IQD_D4_GRAN=256" --mode $m --channels 4096 --log2-samples 16
  bash tools/r6/r6_kt.sh ${m}_lf0_gran512 "IQD_D4_LEADFREE=0 IQD_D4_GRAN=512" --mode $m --channels 4096 --log2-samples 16
  bash tools/r6/r6_kt.sh ${m}14_norec "IQD_D4_LEADFREE=1 IQD_D4_NOREC=1" --mode $m --channels 4096 --log2-samples 14
done
bash tools/r6/r6_kt.sh fm14_lf1 "IQD_D4_LEADFREE=1" --config 2 --log2-samples 14
bash tools/r6/r6_kt.sh fm14_lf0 "IQD_D4_LEADFREE=0" --config 2 --log2-samples 14
bash tools/r6/r6_kt.sh am14_lf1 "IQD_D4_LEADFREE=1" --mode am --channels 4096 --log2-samples 14
bash tools/r6/r6_kt.sh am14_lf0 "IQD_D4_LEADFREE=0" --mode am --channels 4096 --log2-samples 14
