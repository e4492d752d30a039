"""GPU (MI355X): the streaming WBFM pipeline (iqd_stream.hip) pinned on by IQD_F_WBFM_STREAM, against the
oracle and the golden vectors, sample for sample.  The same cases run through the tile kernel in
test_gpu_wbfm.py / test_gpu_modes.py; both must give the reference's PCM."""
import numpy as np
import pytest

from rtlsdrdiags_amd import synth

pytestmark = pytest.mark.gpu

STREAM = 0x4   # IQD_F_WBFM_STREAM
TILES = 0x2    # IQD_F_WBFM_TILES


@pytest.fixture(scope="module")
def capi():
    from rtlsdrdiags_amd import capi as c
    return c


def oracle_run(oracle, u8, block_bytes=32768, gain=None, rotation=1):
    c = oracle.chain()
    c.set_mode("wbfm")
    c.set_rotation(rotation)
    if gain is not None:
        c.set_gain(3, gain)
    return c.accept_stream(u8, block_bytes)


@pytest.mark.parametrize("name", ["fm_tone", "am_tone", "white", "rails", "capture_excerpt"])
def test_golden(capi, golden, name):
    g = golden[name]
    eng = capi.Engine(1, flags=STREAM)
    eng.set_mode("wbfm")
    pcm, cnt, mag, allowed = eng.accept(g["iq"])
    assert eng.stats()["stream_launches"] == 1
    assert cnt[0] == len(g["pcm_wbfm"])
    assert np.array_equal(pcm[0, :cnt[0]], g["pcm_wbfm"])
    assert np.array_equal(mag[0], g["magnitude"])
    assert np.array_equal(allowed[0], g["allowed"])


@pytest.mark.parametrize("rotation", [1, 0, -1])
def test_rotations_and_white_bytes(capi, oracle, rotation):
    """Uniform random bytes hit 0x00 (= -128) at negated positions: the rotation's -(-128) = -128 quirk."""
    rng = np.random.default_rng(5 + rotation)
    u8 = rng.integers(0, 256, size=4 * 32768, dtype=np.uint8)
    u8[100:164] = 0
    ref, ref_mag, _ = oracle_run(oracle, u8, rotation=rotation)
    eng = capi.Engine(1, flags=STREAM)
    eng.set_mode("wbfm")
    eng.set_rotation(rotation)
    pcm, cnt, mag, _ = eng.accept(u8)
    assert eng.stats()["stream_launches"] == 1
    assert np.array_equal(pcm[0, :cnt[0]], ref)
    assert np.array_equal(mag[0], ref_mag)


@pytest.mark.parametrize("blocks_per_call", [1, 3, 8])
def test_calls_carry_state(capi, oracle, blocks_per_call):
    u8 = synth.fm_tone(24 * 16384, seed=77)
    ref, ref_mag, _ = oracle_run(oracle, u8)
    eng = capi.Engine(1, flags=STREAM)
    eng.set_mode("wbfm")
    out, mags = [], []
    step = blocks_per_call * 32768
    for off in range(0, len(u8), step):
        pcm, cnt, mag, _ = eng.accept(u8[off:off + step])
        out.append(pcm[0, :cnt[0]])
        mags.append(mag[0])
    assert np.array_equal(np.concatenate(out), ref)
    assert np.array_equal(np.concatenate(mags), ref_mag)
    assert eng.stats()["state_repairs"] == 0


def test_alternating_with_the_tile_kernel(capi, oracle):
    """Both kernels leave the same carried state: calls may alternate between them."""
    u8 = synth.fm_tone(16 * 16384, seed=78, deviation=60e3)
    ref, _, _ = oracle_run(oracle, u8)
    engs = [capi.Engine(1, flags=STREAM), capi.Engine(1, flags=TILES)]
    for e in engs:
        e.set_mode("wbfm")
    # one engine per path would not alternate; drive ONE stream through a state copy instead: the tile engine
    # gets every call, the stream engine gets every call too, and both must match the reference at every call
    outs = [[], []]
    for off in range(0, len(u8), 2 * 32768):
        for k, e in enumerate(engs):
            pcm, cnt, _, _ = e.accept(u8[off:off + 2 * 32768])
            outs[k].append(pcm[0, :cnt[0]])
    assert np.array_equal(np.concatenate(outs[0]), ref)
    assert np.array_equal(np.concatenate(outs[1]), ref)


def test_many_channels_short_rows(capi, oracle):
    n_ch = 200
    u8 = np.stack([synth.fm_tone(2 * 16384, seed=3000 + c, deviation=4000.0 + 350 * c) for c in range(n_ch)])
    eng = capi.Engine(n_ch, flags=STREAM)
    eng.set_mode("wbfm")
    pcm, cnt, mag, _ = eng.accept(u8)
    assert eng.stats()["stream_launches"] == 1
    for c in range(n_ch):
        ref, ref_mag, _ = oracle_run(oracle, u8[c])
        assert np.array_equal(pcm[c, :cnt[c]], ref), c
        assert np.array_equal(mag[c], ref_mag), c


def test_long_row_many_segments(capi, oracle):
    """2^22 samples in one call: several hundred segments, every hand-off verified bit for bit."""
    u8 = synth.fm_tone(1 << 22, seed=4321)
    ref, ref_mag, _ = oracle_run(oracle, u8)
    eng = capi.Engine(1, flags=STREAM)
    eng.set_mode("wbfm")
    pcm, cnt, mag, _ = eng.accept(u8)
    assert np.array_equal(pcm[0, :cnt[0]], ref)
    assert np.array_equal(mag[0], ref_mag)
    st = eng.stats()
    assert st["stream_launches"] == 1 and st["state_checks"] >= 100 and st["state_repairs"] == 0


def test_gain_change_reset_and_small_blocks(capi, oracle):
    u8 = synth.fm_tone(8 * 16384, seed=99, deviation=70e3)
    c = oracle.chain()
    c.set_mode("wbfm")
    eng = capi.Engine(1, block_bytes=4096, flags=STREAM)
    eng.set_mode("wbfm")
    ref, out = [], []
    chunks = [(0, 40960), (40960, 90112), (90112, 131072), (131072, 262144)]
    for k, (lo, hi) in enumerate(chunks):
        if k == 1:
            c.set_gain(3, 91000.0)
            eng.set_gain("wbfm", 91000.0)
        if k == 2:
            c.reset()
            eng.reset()
        if k == 3:
            c.set_gain(3, 12000.0)
            eng.set_gain("wbfm", 12000.0)
        r, _, _ = c.accept_stream(u8[lo:hi], 4096)
        pcm, cnt, _, _ = eng.accept(u8[lo:hi])
        ref.append(r)
        out.append(pcm[0, :cnt[0]])
    for k in range(len(chunks)):
        assert np.array_equal(out[k], ref[k]), k


def test_loud_audio_takes_the_clamped_path(capi, golden):
    """cast_overflow's gain drives the audio decimator into its per-MAC clamp - but K is then too large for the
    streaming kernel's bounded (int16) cast, so the engine must fall back to the tile kernel by itself."""
    g = golden["cast_overflow"]
    eng = capi.Engine(1, flags=STREAM)
    eng.set_mode("wbfm")
    eng.set_gain("wbfm", float(g["gain_wbfm"]))
    pcm, cnt, _, _ = eng.accept(g["iq"])
    assert np.array_equal(pcm[0, :cnt[0]], g["pcm_wbfm"])


def test_loud_but_bounded_gain(capi, oracle):
    """A gain 40x the default keeps (int16)y bounded but saturates the audio path: |y2| > 16061 switches the
    40-tap decimator to the reference's clamp-after-every-MAC order."""
    u8 = synth.fm_tone(6 * 16384, seed=5, deviation=75e3)
    gain = 40 * 256000 / (2 * np.pi)
    ref, _, _ = oracle_run(oracle, u8, gain=gain)
    eng = capi.Engine(1, flags=STREAM)
    eng.set_mode("wbfm")
    eng.set_gain("wbfm", gain)
    pcm, cnt, _, _ = eng.accept(u8)
    assert eng.stats()["stream_launches"] == 1
    assert np.array_equal(pcm[0, :cnt[0]], ref)


def test_channels_of_several_rotation_selectors_stream_in_one_launch(capi, oracle):
    """WBFM channels with different rotation selectors used to send the whole launch to the tile kernels:
    the channel list is sorted by selector, each group's segment ids padded to 16, a P wave takes its rounds' selector
    (StreamArgs::grouped).  210 channels, selectors mixed unevenly, own data, two calls, every channel against the oracle."""
    n_ch, n = 210, 1 << 15
    rng = np.random.default_rng(8)
    rots = [int(r) for r in rng.choice([1, 1, 1, 0, -1, -1], n_ch)]
    u8 = np.stack([np.roll(synth.fm_tone(n, seed=300 + (c % 9), amplitude=25.0 + c % 40), 2 * ((c * 37) % 1009)) for c in range(n_ch)])
    eng = capi.Engine(n_ch, flags=STREAM)
    eng.set_mode("wbfm")
    chains = []
    for c in range(n_ch):
        eng.set_rotation(rots[c], first=c, n=1)
        o = oracle.chain()
        o.set_mode("wbfm")
        o.set_rotation(rots[c])
        chains.append(o)
    for call in range(2):
        before = eng.stats()["stream_launches"]
        pcm, cnt, mag, _ = eng.accept(u8)
        assert eng.stats()["stream_launches"] - before == 1
        for c in range(n_ch):
            ref, rmag, _ = chains[c].accept_stream(u8[c])
            assert cnt[c] == len(ref) and np.array_equal(pcm[c, :cnt[c]], ref), (call, c, rots[c])
            assert np.array_equal(mag[c], rmag), (call, c)
    assert eng.stats()["state_repairs"] == 0


@pytest.mark.parametrize("gain", [None, 5.0e7])
def test_restart_state_when_the_last_streamed_segment_is_shorter_than_the_lead_in(capi, oracle, gain):
    """A channel's restart state is its de-emphasis state 768 samples before its end.  When the channel's LAST streamed segment
    is shorter than that, the point lies in the segment before it - a cold segment's state is exact from its own start on, not
    inside its lead-in, where the short one used to take the record (since round 2; a converging state there left the NEXT
    call's first PCM samples off by one now and then - round 4's fuzz campaign found it at a demodulator gain of 5e7, with the
    streaming path pinned).  Calls whose rows are k segments of 768 samples + a remainder of 128 ... 640, streaming pinned,
    three calls each, several channels with their own data; once with a squelch that closes on some 128-sample blocks, so that
    the remainders differ from channel to channel."""
    rng = np.random.default_rng(77)
    for n, thr in ((5 * 768 + 128, -200), (768 + 128, -200), (9 * 768 + 640, -200), (3 * 768 + 384, -200), (40 * 128, -38)):
        n_ch = 6
        eng = capi.Engine(n_ch, block_bytes=256, flags=4)
        chains = []
        for c in range(n_ch):
            o = oracle.chain()
            o.set_mode("wbfm")
            o.set_squelch(thr)
            if gain is not None:
                o.set_gain(3, gain)
            chains.append(o)
        eng.set_mode("wbfm")
        eng.set_squelch(thr)
        if gain is not None:
            eng.set_gain("wbfm", gain)
        for call in range(3):
            rows = np.stack([synth.fm_tone(n, seed=int(rng.integers(1 << 30)), deviation=float(rng.uniform(1e3, 7e4)),
                                           amplitude=float(rng.uniform(20, 120)), sigma=1.0) for _ in range(n_ch)])
            if thr > -200:                                   # quiet stretches of a few blocks: the squelch drops all but the first of each
                for c in range(n_ch):
                    for b0 in rng.integers(0, n // 128 - 4, 3):
                        rows[c, 256 * int(b0):256 * (int(b0) + int(rng.integers(2, 5)))] = 128
            pcm, cnt, _, allowed = eng.accept(rows)
            for c in range(n_ch):
                ref, _, rall = chains[c].accept_stream(rows[c], 256)
                assert np.array_equal(allowed[c], rall), (n, thr, call, c)
                assert cnt[c] == len(ref) and np.array_equal(pcm[c, :cnt[c]], ref), (n, thr, call, c, np.flatnonzero(pcm[c, :cnt[c]] != ref)[:8])
        assert eng.stats()["stream_launches"] == 3, eng.stats()
        eng.close()
