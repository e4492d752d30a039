#!/bin/bash
for mode in am usb fm; do
  echo "== $mode"
  tools/abn.sh 3 "--mode $mode --channels 4096 --log2-samples 16" tmp_variants/lib_d4base.so tmp_variants/lib_d4mqsad.so | grep median
done
