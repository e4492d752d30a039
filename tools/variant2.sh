#!/bin/bash
# Builds a variant of libiqdemod.so into tmp_variants/lib_<name>.so with extra -D flags for the WBFM stream translation
# units (iqd_stream.hip AND iqd_stream_mixed.hip, which compiles the same bodies a second time):
#   tools/variant2.sh <name> [-DFLAG=1 ...]
set -e
NAME=$1; shift 1
cd "$(dirname "$0")/../rtlsdrdiags_amd/csrc"
O=/tmp/iqd_objs; mkdir -p $O ../../tmp_variants
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -Wno-unused-function -I. -I../../include"
for f in iqd_kernels.hip iqd_stream2.hip iqd_engine.cpp iqd_host.cpp iqd_plan.cpp iqd_gather.cpp IqDataProcessor.cc; do
  if [ ! -f $O/$f.o ] || [ $f -nt $O/$f.o ] || [ -n "$(find . -name '*.h' -newer $O/$f.o)" ]; then /opt/rocm/bin/hipcc $FL -c $f -o $O/$f.o & fi
done
/opt/rocm/bin/hipcc $FL "$@" -c iqd_stream.hip -o $O/v2_${NAME}_stream.o &
/opt/rocm/bin/hipcc $FL "$@" -c iqd_stream_mixed.hip -o $O/v2_${NAME}_mixed.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tmp_variants/lib_$NAME.so $O/iqd_kernels.hip.o $O/iqd_stream2.hip.o $O/iqd_engine.cpp.o $O/iqd_host.cpp.o $O/iqd_plan.cpp.o $O/iqd_gather.cpp.o $O/IqDataProcessor.cc.o $O/v2_${NAME}_stream.o $O/v2_${NAME}_mixed.o -ldl
echo built tmp_variants/lib_$NAME.so
