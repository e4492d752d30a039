"""Diagnostic: PCM of one WBFM streaming call without squelch magnitudes (IQD_F_NO_MAGNITUDE | IQD_F_WBFM_STREAM), as an md5 -
run once per library build (IQD_LIB) and compare.  python tools/swp_check.py"""
import hashlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtlsdrdiags_amd import capi, synth
u8 = synth.fm_tone(1 << 22, seed=77)
eng = capi.Engine(1, flags=1 | 4)
eng.set_mode("wbfm")
h = hashlib.md5()
for k in range(2):
    pcm, cnt, _, _ = eng.accept(u8)
    h.update(pcm[0, :cnt[0]].tobytes())
print(os.environ.get("IQD_LIB", "default"), h.hexdigest(), eng.stats()["stream_launches"], eng.stats()["state_repairs"])
