#!/bin/bash
for m in usb am; do
  for lf in 1 0; do bash tools/r6/r6_kt.sh ${m}_lf$lf IQD_D4_LEADFREE=$lf --mode $m --channels 4096 --log2-samples 16; done
done
for lf in 1 0; do bash tools/r6/r6_kt.sh fm_lf$lf IQD_D4_LEADFREE=$lf --config 2; done
for lf in 1 0; do bash tools/r6/r6_kt.sh usb14_lf$lf IQD_D4_LEADFREE=$lf --mode usb --channels 4096 --log2-samples 14; done
