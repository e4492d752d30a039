#!/bin/bash
# round 5, third box: workgroups of one and two rings (IQD_RINGS) - correctness under the streaming tests, then the timing probe
mkdir -p gpurun_out
for r in 1 2; do
  ( IQD_RINGS=$r timeout 900 python3 -m pytest tests/test_gpu_stream.py tests/test_gpu_stream2.py tests/test_gpu_gain_epochs.py -x -q ) > gpurun_out/r5_rings${r}_tests.log 2>&1
  echo "IQD_RINGS=$r: $(tail -1 gpurun_out/r5_rings${r}_tests.log)"
done
( IQD_RINGS=1 IQD_WBFM_PATH=stream FUZZ_WIDE=1 FUZZ_WIDE_RANGE=24,600 timeout 100 python3 tools/gpu_fuzz.py 60 901 ) 2>&1 | tail -1
( IQD_RINGS=2 IQD_WBFM_PATH=stream timeout 100 python3 tools/gpu_fuzz.py 60 902 ) 2>&1 | tail -1
tools/rings_probe.sh 2>&1 | tee gpurun_out/r5_rings_probe.log
# the cold rings' lead-in 768 -> 640 / 512 (IQD_ST_COLD_HALO): time, and how often a hand-off then fails (state_repairs per 20 steps)
for sig in fm_tone white carrier; do
  for L in base cold640 cold512; do
    out=$(IQD_LIB=$PWD/tmp_variants/lib_$L.so python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-path --no-live-pmc --no-from-idle --signal $sig 2>/dev/null | grep '"metric"')
    echo "$sig $L $(echo "$out" | grep -o '"ms_per_step": [0-9.]*') $(echo "$out" | grep -o '"kernel_ms": [0-9.]*') $(echo "$out" | grep -o '"state_repairs": [0-9]*') $(echo "$out" | grep -o '"segment_repairs": [0-9]*') $(echo "$out" | grep -o '"state_checks": [0-9]*')"
  done
done 2>&1 | tee gpurun_out/r5_cold_halo.log
tools/abn.sh 3 "" tmp_variants/lib_base.so tmp_variants/lib_cold640.so tmp_variants/lib_cold512.so 2>&1 | grep median | tee -a gpurun_out/r5_cold_halo.log
# new GPU tests of this round's other changes
python3 -m pytest tests/test_gpu_boundary.py -x -q -k "demod" 2>&1 | tail -2
