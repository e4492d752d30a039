#!/bin/bash
# Builds a variant of libiqdemod.so into tmp_variants/lib_<name>.so with extra -D flags for ONE translation unit:
#   tools/variant.sh <name> <file.hip> [-DFLAG=1 ...]
# (the other objects are compiled once into /tmp/iqd_objs and reused; for A/B runs with tools/ab.sh)
set -e
NAME=$1; UNIT=$2; shift 2
cd "$(dirname "$0")/../rtlsdrdiags_amd/csrc"
O=/tmp/iqd_objs; mkdir -p $O ../../tmp_variants
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -Wno-unused-function -I. -I../../include"
for f in iqd_kernels.hip iqd_stream.hip iqd_stream2.hip iqd_stream_mixed.hip iqd_engine.cpp iqd_host.cpp iqd_plan.cpp iqd_gather.cpp IqDataProcessor.cc; do
  if [ "$f" = "$UNIT" ]; then /opt/rocm/bin/hipcc $FL "$@" -c $f -o $O/variant_$NAME.o &
  elif [ ! -f $O/$f.o ] || [ $f -nt $O/$f.o ] || [ -n "$(find . -name '*.h' -newer $O/$f.o)" ]; then /opt/rocm/bin/hipcc $FL -c $f -o $O/$f.o &
  fi
done
wait
OBJS=""
for f in iqd_kernels.hip iqd_stream.hip iqd_stream2.hip iqd_stream_mixed.hip iqd_engine.cpp iqd_host.cpp iqd_plan.cpp iqd_gather.cpp IqDataProcessor.cc; do
  if [ "$f" = "$UNIT" ]; then OBJS="$OBJS $O/variant_$NAME.o"; else OBJS="$OBJS $O/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tmp_variants/lib_$NAME.so $OBJS -ldl
echo built tmp_variants/lib_$NAME.so
