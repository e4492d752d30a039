#!/usr/bin/env python3
"""One long WBFM row (2^28 samples = 16 384 blocks) with a running AGC: ms per step (run on the GPU box).
    IQD_LIB=tmp_variants/lib_x.so python3 tools/agc_row_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rtlsdrdiags_amd import capi, synth

n = 1 << 28
dev = torch.device("cuda:0")
period = synth.fm_tone(1 << 24, seed=1234)
iq = torch.from_numpy(period).to(dev).repeat(n // (1 << 24))
pcm = torch.zeros(n // 32, dtype=torch.int16, device=dev)
cnt = torch.zeros(1, dtype=torch.int32, device=dev)
nblk = 2 * n // 32768
mag = torch.zeros(nblk, dtype=torch.int32, device=dev)
al = torch.zeros(nblk, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for label, agc, thr, amp in (("no AGC", None, -200, None), ("Harris AGC, squelch open", 1, -200, None), ("lowpass AGC, squelch open", 0, -200, None),
                             ("Harris AGC, squelch at -60 dBFS", 1, -60, None)):
    eng = capi.Engine(1)
    eng.set_mode("wbfm")
    eng.set_squelch(thr)
    if agc is not None:
        eng.agc_set_type(agc)
        eng.agc_enable(True)
    def step():
        eng.accept_device(iq.data_ptr(), 2 * n, pcm.data_ptr(), cnt.data_ptr(), mag.data_ptr(), al.data_ptr())
    for _ in range(40):
        step()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    eng.synchronize()
    print("%-34s %.3f ms per step   (IF gain now %d dB)" % (label, (time.perf_counter() - t0) / 20 * 1e3, eng.rx_gain_db(0)))
    eng.close()
