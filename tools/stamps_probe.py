"""Diagnostic: per-phase cycle shares of the WBFM chain kernel.  Needs a library built with -DIQD_STAMPS:
   IQD_LIB=.../libiqdemod_stamps.so python tools/stamps_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rtlsdrdiags_amd import capi, synth
n = 1<<28; period = 1<<24
u8 = synth.fm_tone(period, seed=1234)
iq = torch.from_numpy(u8).cuda().repeat(n//period)
pcm = torch.zeros(n//32, dtype=torch.int16, device='cuda'); torch.cuda.synchronize()
eng = capi.Engine(1); eng.set_mode('wbfm')
for _ in range(3): eng.accept_device(iq.data_ptr(), 2*n, pcm.data_ptr())
s = eng.debug_stamps()
names = ['st2+sync+st3 | w0 slot','IIR(k)','P1(k+1)','wait at B1','Y: store+S1','-','-','loop-top']
for w in (0,1):
    tot = sum(s[8*w:8*w+8]) or 1
    print('wave', w, ' '.join('%s=%.1f%%' % (names[k], 100.0*s[8*w+k]/tot) for k in range(8)), 'total Mcycles', tot/1e6)
