"""Diagnostic: how long each family's workgroups of the fused mixed launch run (build iqd_stream_mixed.hip with
-DIQD_MIXED_TIMING=1: tools/variant.sh mxt iqd_stream_mixed.hip -DIQD_MIXED_TIMING=1).
    IQD_LIB=tmp_variants/lib_mxt.so [IQD_FAMILY_WEIGHTS=am,fm,wbfm,ssb] python3 tools/mixed_probe.py [channels] [log2 samples]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtlsdrdiags_amd import capi, synth
n_ch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 16)
eng = capi.Engine(n_ch)
names = ["am", "fm", "wbfm", "lsb", "usb"]
for k, m in enumerate(names):
    for c in range(k, min(n_ch, 5), 5):
        pass
modes = [names[c % 5] for c in range(n_ch)]
# runs of equal settings: per channel here (set-up time does not matter)
for c in range(n_ch):
    eng.set_mode(modes[c], first=c, n=1)
u8 = synth.fm_tone(n, seed=7)
iq = eng.dev_alloc(2 * n * n_ch); pcm = eng.dev_alloc(2 * (n // 32) * n_ch)
eng.dev_upload(iq, u8); eng.dev_tile(iq, 2 * n, 2 * n * n_ch)
for k in range(3):
    eng.accept_device(iq, 2 * n, pcm); eng.synchronize()
    raw = eng.debug_stamps_ext(16 + 512 + 304)
    st, t0s = raw[16:16 + 304], [t for t in raw[16 + 512:] if t]
    fam = {0: "am", 1: "fm", 2: "wbfm", 3: "ssb"}
    rows = {}
    for v in st:
        if v:
            rows.setdefault(fam[v >> 56], []).append((v & ((1 << 56) - 1)) / 100.0)
    ends = [t + (v & ((1 << 56) - 1)) for t, v in zip(raw[16 + 512:], st) if v]
    print("launch", k, " first to last workgroup START %.1f us, first start to last END %.1f us" % ((max(t0s) - min(t0s)) / 100.0, (max(ends) - min(t0s)) / 100.0))
    print("launch", k, " workgroups, slowest / mean / fastest (us):",
          {f: (len(t), round(max(t), 1), round(sum(t) / len(t), 1), round(min(t), 1)) for f, t in rows.items()})
