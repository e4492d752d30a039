"""Debug: the 1410-channel mixed call (tests/test_gpu_scale.py::test_mixed_1400_channels_at_the_share_threshold), one call,
progress printed.  argv[1]: all | wbfm | fm | am | ssb  (which families' channels get a mode; the others stay 'none')."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from rtlsdrdiags_amd import capi
import test_gpu_scale as T
which = sys.argv[1]
n_ch, n = int(sys.argv[2]) if len(sys.argv) > 2 else 1410, 1 << 16
u8 = T._mixed_rows(n_ch, n, seed=n_ch)
eng = capi.Engine(n_ch, flags=int(os.environ.get("FLAGS", "0")))
modes, rots = T._mixed_setup(eng, n_ch, 1)
if which != "all":
    keep = {"wbfm": ("wbfm",), "fm": ("fm",), "am": ("am",), "ssb": ("lsb", "usb")}[which]
    for c in range(n_ch):
        if modes[c] not in keep:
            eng.set_mode("none", first=c, n=1)
iq_d, pcm_d = eng.dev_alloc(u8.nbytes), eng.dev_alloc(n_ch * (n // 32) * 2)
nblk = 2 * n // 32768
cnt_d, mag_d, al_d = eng.dev_alloc(n_ch * 4), eng.dev_alloc(n_ch * nblk * 4), eng.dev_alloc(n_ch * nblk)
eng.dev_upload(iq_d, u8)
print("uploaded", which, flush=True)
eng.accept_device(iq_d, 2 * n, pcm_d, cnt_d, mag_d, al_d)
print("queued", flush=True)
eng.synchronize()
print("synced", eng.stats(), flush=True)
