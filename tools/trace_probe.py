"""Diagnostic: event times of one workgroup's waves, piece by piece (build with -DIQD_ST_TRACE=1).
IQD_LIB=<variant> python3 tools/trace_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtlsdrdiags_amd import capi, synth
n, period = 1 << 28, 1 << 24
u8 = synth.fm_tone(period, seed=1234)
eng = capi.Engine(1); eng.set_mode("wbfm")
iq = eng.dev_alloc(2 * n); pcm = eng.dev_alloc(2 * (n // 32))
eng.dev_upload(iq, u8); eng.dev_tile(iq, 2 * period, 2 * n)
for k in range(2):
    eng.accept_device(iq, 2 * n, pcm); eng.synchronize()
st = np.array(eng.debug_stamps_ext(64 + 16 * 256 * 4)[64:], dtype=np.int64).reshape(16, 256, 4)
t0 = st[st > 0].min()
rel = np.where(st > 0, st - t0, -1)
ring = 0
waves = [0] + [3 + pw for pw in range(12) if pw % 3 == ring]      # IIR wave 0 and the P waves of ring 0
print("times relative to the workgroup's first event; ring 0: IIR wave 0 (seen, released, done) and P waves", waves[1:], "(start, computed, ring free, signalled)")
for piece in list(range(100, 108)):
    row = ["piece %3d" % piece, "IIR " + " ".join("%7d" % x for x in rel[0, piece, :3])]
    for w in waves[1:]:
        row.append("w%-2d " % w + " ".join("%7d" % x for x in rel[w, piece]))
    print(" | ".join(row))
per = np.diff(rel[0, 60:180, 0]).mean()
print("period (IIR sees a piece): %.0f cycles" % per)
