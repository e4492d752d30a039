"""Diagnostic (CPU, oracle): how many samples the WBFM de-emphasis recurrence needs, started from the zero state somewhere
in a stream, before its state is bit-identical to the uninterrupted run's and stays so - the statistic behind the
768-sample lead-in of the streaming kernel's cold segments.  python3 tools/deemph_convergence.py"""
import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bindings as B
from rtlsdrdiags_amd import synth
o=B.Oracle()
f32=np.float32
b0=f32(0.0253863); a1=f32(-0.9492274)
gain=f32(256000/(2*np.pi)); K=f32(f32(gain/f32(75000))*f32(32767))
def run(name,u8,L=1024,step=64):
    s8=(u8.astype(np.int16)-128).astype(np.int8)
    rot=o.rotate(s8,1)
    st=o.wbfm_stages(rot)
    d=st["dtheta"]; ytrue=st["deemph"]
    u=(b0*(K*d).astype(f32)).astype(f32)
    n=len(u)
    starts=np.arange(2000, n-L-1, step)
    y=np.zeros(len(starts),f32); up=np.zeros(len(starts),f32)
    last_bad=np.zeros(len(starts),np.int32)
    for t in range(L):
        x=u[starts+t]
        tn=(x+up).astype(f32); r=(a1*y).astype(f32); y=(tn-r).astype(f32); up=x
        bad = y.view(np.uint32)!=ytrue[starts+t].view(np.uint32)
        last_bad[bad]=t+1
    conv=last_bad
    print(name, "starts",len(starts),"conv steps: mean %.0f p99 %d p99.9 %d max %d; >512: %d >640: %d >768: %d"%(conv.mean(), np.percentile(conv,99), np.percentile(conv,99.9), conv.max(), (conv>512).sum(), (conv>640).sum(), (conv>768).sum()))
    return conv
n=1<<22
run("fm_tone", synth.fm_tone(n,seed=1234), step=16)
run("white", synth.white_u8(n,seed=5), step=16)
run("quiet", synth.fm_tone(n,seed=7,deviation=3000.0), step=16)
run("small", synth.fm_tone(n,seed=8,amplitude=6.0,sigma=1.0), step=16)
