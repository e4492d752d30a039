"""GPU (MI355X): host-pointer accepts large enough to be sliced and double-buffered (iqd_accept_iq cuts calls of
64 MiB and more into ~32 MiB slices whose uploads overlap the kernels) give exactly what the reference gives for
the unsliced stream: time slices of long rows (with and without squelch gating, pageable and page-locked host
buffers) and channel slices of many short rows."""
import numpy as np
import pytest

from rtlsdrdiags_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from rtlsdrdiags_amd import capi as c
    return c


def test_time_slices_one_long_wbfm_row(capi, oracle):
    n = 1 << 26                                     # 128 MiB of IQ: four slices
    iq = np.tile(synth.fm_tone(1 << 22, seed=77), n >> 22)
    eng = capi.Engine(1)
    eng.set_mode("wbfm")
    pcm, cnt, mag, allowed = eng.accept(iq)
    o = oracle.chain()
    o.set_mode("wbfm")
    ref, rmag, rallowed = o.accept_stream(iq)
    assert cnt[0] == len(ref) == n // 32
    assert np.array_equal(pcm[0], ref)
    assert np.array_equal(mag[0], rmag) and np.array_equal(allowed[0], rallowed)


def test_time_slices_gated_rows_from_pinned_memory(capi, oracle):
    """Three rows of 32 MiB, squelch closing on most blocks: every slice delivers fewer samples than it has room
    for and the rows are compacted behind each other."""
    nblk, bb = 1024, 32768
    rng = np.random.default_rng(5)
    modes = ["am", "usb", "fm"]
    eng = capi.Engine(3)
    iq = eng.host_array((3, nblk * bb))
    for c in range(3):
        amps = np.where(rng.random(nblk) < 0.3, 70, 2)
        amps[:2] = (2, 70)
        iq[c] = synth.stepped_amplitude(list(amps), block_samples=bb // 2, seed=40 + c)
        eng.set_mode(modes[c], first=c, n=1)
    eng.set_squelch(-40)
    pcm = eng.host_array((3, nblk * bb // 64), np.int16)
    cnt = np.zeros(3, np.uint32)
    mag = np.zeros((3, nblk), np.uint32)
    allowed = np.zeros((3, nblk), np.uint8)
    eng.accept_into(iq, pcm, cnt, mag, allowed)
    for c in range(3):
        o = oracle.chain()
        o.set_mode(modes[c])
        o.set_squelch(-40)
        ref, rmag, rallowed = o.accept_stream(np.array(iq[c]))
        assert np.array_equal(allowed[c], rallowed), c
        assert np.array_equal(mag[c], rmag), c
        assert cnt[c] == len(ref) and 0 < cnt[c] < pcm.shape[1], c
        assert np.array_equal(pcm[c, :cnt[c]], ref), c
        assert not pcm[c, cnt[c]:].any(), c
    eng.host_free(iq)
    eng.host_free(pcm)


def test_channel_slices_many_short_rows(capi, oracle):
    n_ch, n = 4096, 16384                           # 128 MiB: slices of 1024 whole rows
    order = ["am", "fm", "wbfm", "lsb", "usb"]
    base = [synth.fm_tone(n, seed=900 + k, deviation=2000.0 + 900.0 * k, amplitude=35.0 + 5 * k) for k in range(9)]
    iq = np.stack([np.roll(base[c % 9], 2 * (c % 211)) for c in range(n_ch)])
    eng = capi.Engine(n_ch)
    for c0 in range(0, n_ch, 64):                   # runs of 64 channels per mode
        eng.set_mode(order[(c0 // 64) % 5], first=c0, n=64)
    pcm, cnt, mag, allowed = eng.accept(iq)
    assert (cnt == 512).all() and allowed.all()
    for c in list(range(0, n_ch, 173)) + [1023, 1024, n_ch - 1]:
        o = oracle.chain()
        o.set_mode(order[(c // 64) % 5])
        ref, rmag, _ = o.accept_stream(iq[c])
        assert np.array_equal(pcm[c], ref), c
        assert mag[c, 0] == rmag[0], c


def test_channel_slices_of_one_short_block_per_row(capi, oracle):
    """4096 rows of 16384 bytes = ONE short block each (block_bytes is 32768), 64 MiB: the sliced path has to size its
    magnitude / flag staging by the call's own block length (it used block_bytes and got 0 blocks per row)."""
    n_ch, nbytes = 4096, 16384
    order = ["fm", "am", "wbfm", "usb"]
    base = [synth.fm_tone(nbytes // 2, seed=300 + k, amplitude=20.0 + 9 * k) for k in range(7)]
    iq = np.stack([np.roll(base[c % 7], 2 * (c % 97)) for c in range(n_ch)])
    eng = capi.Engine(n_ch)
    for c0 in range(0, n_ch, 32):
        eng.set_mode(order[(c0 // 32) % 4], first=c0, n=32)
    pcm, cnt, mag, allowed = eng.accept(iq)
    assert pcm.shape == (n_ch, nbytes // 64) and mag.shape == (n_ch, 1)
    assert (cnt == nbytes // 64).all() and allowed.all()
    for c in list(range(0, n_ch, 311)) + [2047, 2048, n_ch - 1]:
        o = oracle.chain()
        o.set_mode(order[(c // 32) % 4])
        ref, rmag, _ = o.accept_stream(iq[c], nbytes)          # one acceptIqData call of 16384 bytes
        assert np.array_equal(pcm[c], ref), c
        assert mag[c, 0] == rmag[0], c


def test_small_call_sliced_call_small_call_on_one_engine(capi, oracle):
    """One engine, three host-pointer paths in turn: a one-block call (straight out of page-locked host memory), a
    64 MiB call (sliced and double-buffered), a one-block call again.  ADVICE r3: the sliced path used to free the small
    path's pinned staging buffer without forgetting it, so the third call wrote into freed memory."""
    eng = capi.Engine(1)
    eng.set_mode("fm")
    o = oracle.chain()
    o.set_mode("fm")
    small_a = synth.fm_tone(16384, seed=601)
    big = np.tile(synth.fm_tone(1 << 21, seed=602), 16)        # 2^25 samples = 64 MiB: the sliced path
    small_b = synth.fm_tone(16384, seed=603)
    for part in (small_a, big, small_b, small_a):
        pcm, cnt, mag, allowed = eng.accept(part)
        ref, rmag, rallowed = o.accept_stream(part)
        assert cnt[0] == len(ref) and np.array_equal(pcm[0, :cnt[0]], ref)
        assert np.array_equal(mag[0], rmag) and np.array_equal(allowed[0], rallowed)
    eng.close()
