"""Diagnostic: ring waits and per-wave phase cycles of the WBFM streaming kernel (builds with -DIQD_ST_WAITSTAT=1
-DIQD_ST_TIMING=1, see iqd_stream.hip).  IQD_LIB=<variant> python3 tools/waitstat_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdrdiags_amd import capi, synth
n, period = 1 << 28, 1 << 24
u8 = synth.fm_tone(period, seed=1234)
eng = capi.Engine(1); eng.set_mode("wbfm")
iq = eng.dev_alloc(2 * n); pcm = eng.dev_alloc(2 * (n // 32))
eng.dev_upload(iq, u8); eng.dev_tile(iq, 2 * period, 2 * n)
prev = [0] * 64
for k in range(3):
    eng.accept_device(iq, 2 * n, pcm); eng.synchronize()
    st = eng.debug_stamps_ext(64)
    d = [a - b for a, b in zip(st, prev)]; prev = st
    per_wave_pieces = d[2] / 12.0          # pieces summed over the launch's P waves / 12 waves per workgroup
    print("launch", k, "P sleeps/piece %.2f  IIR sleeps/piece %.2f" % (d[0] / max(d[2], 1), d[1] / max(d[3], 1)))
    print("  P wave (hardware wave 3 + pw; ring pw % 3):  compute / ring wait / raw wait + hand-over, cycles per piece")
    for pw in range(12):
        print("   pw %2d  wave %2d  ring %d:  %5.0f  %5.0f  %5.0f" % (pw, pw + 3, pw % 3, d[16 + pw] / per_wave_pieces, d[32 + pw] / per_wave_pieces, d[48 + pw] / per_wave_pieces))
    for r in range(3):
        print("   IIR ring %d (wave %d): wait %5.0f of %5.0f; from seeing a piece to releasing the ring %5.0f" % (r, r, d[28 + r] / per_wave_pieces, d[44 + r] / per_wave_pieces, d[60 + r] / per_wave_pieces))
