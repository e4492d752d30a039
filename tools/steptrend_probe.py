"""Diagnostic: the WBFM bench step's kernel time (HIP events: stream + fix-up kernel) step by step from an idle device, then
under two seconds of sustained load, then after half a second of idling (bench.py's clock-settle phase)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from rtlsdrdiags_amd import capi, synth
n, period = 1 << 28, 1 << 24
u8 = synth.fm_tone(period, seed=1234)
eng = capi.Engine(1); eng.set_mode("wbfm")
iq = eng.dev_alloc(2 * n); pcm = eng.dev_alloc(2 * (n // 32))
eng.dev_upload(iq, u8); eng.dev_tile(iq, 2 * period, 2 * n)
eng.synchronize()
eng.set_profiling(True)
prev = eng.stats()
out = []
for k in range(80):
    eng.accept_device(iq, 2 * n, pcm)
    st = eng.stats()          # synchronises
    out.append(round((st["chain_kernel_ms"] - prev["chain_kernel_ms"]) * 1000))
    prev = st
print("kernel us per step:", out)
# ... and under sustained load: 6000 more steps (about 2 s), the mean of every 500
eng.set_profiling(True)
for blk in range(12):
    a = eng.stats()
    for k in range(500):
        eng.accept_device(iq, 2 * n, pcm)
    b = eng.stats()
    print("steps %5d..%5d: kernel %.1f us per step" % (80 + 500 * blk, 80 + 500 * blk + 499, (b["chain_kernel_ms"] - a["chain_kernel_ms"]) * 1000 / 500))
time.sleep(0.5)
out = []
for k in range(10):
    eng.accept_device(iq, 2 * n, pcm)
    st = eng.stats()
    out.append(round((st["chain_kernel_ms"] - prev["chain_kernel_ms"]) * 1000))
    prev = st
print("after 0.5 s idle:", out)
