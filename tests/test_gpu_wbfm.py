"""GPU (MI355X): the HIP WBFM chain through the C ABI against the oracle and the golden vectors."""
import numpy as np
import pytest

from rtlsdrdiags_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from rtlsdrdiags_amd import capi as c
    return c


def oracle_pcm(oracle, u8, mode="wbfm", block_bytes=32768, gain=None, rotation=1):
    c = oracle.chain()
    c.set_mode(mode)
    c.set_rotation(rotation)
    if gain is not None:
        c.set_gain(3, gain)
    return c.accept_stream(u8, block_bytes)


@pytest.mark.parametrize("name", ["fm_tone", "am_tone", "white", "rails", "capture_excerpt"])
def test_golden_wbfm(capi, golden, name):
    g = golden[name]
    eng = capi.Engine(1)
    eng.set_mode("wbfm")
    pcm, cnt, mag, allowed = eng.accept(g["iq"])
    assert cnt[0] == len(g["pcm_wbfm"])
    assert np.array_equal(pcm[0, :cnt[0]], g["pcm_wbfm"])
    assert np.array_equal(mag[0], g["magnitude"])
    assert np.array_equal(allowed[0], g["allowed"])
    if name != "rails":   # periodic input can hold two de-emphasis trajectories 1 ulp apart for
        assert eng.stats()["state_repairs"] == 0   # ever; the exact repair path then steps in


def test_golden_cast_overflow_gain(capi, golden):
    g = golden["cast_overflow"]
    eng = capi.Engine(1)
    eng.set_mode("wbfm")
    eng.set_gain("wbfm", float(g["gain_wbfm"]))
    pcm, cnt, _, _ = eng.accept(g["iq"])
    assert np.array_equal(pcm[0, :cnt[0]], g["pcm_wbfm"])


@pytest.mark.parametrize("blocks_per_call", [1, 2, 8])
def test_streaming_calls_carry_state(capi, oracle, blocks_per_call):
    u8 = synth.fm_tone(8 * 16384, seed=31)
    ref, ref_mag, _ = oracle_pcm(oracle, u8)
    eng = capi.Engine(1)
    eng.set_mode("wbfm")
    out, mags = [], []
    step = blocks_per_call * 32768
    for off in range(0, len(u8), step):
        pcm, cnt, mag, _ = eng.accept(u8[off:off + step])
        out.append(pcm[0, :cnt[0]])
        mags.append(mag[0])
    assert np.array_equal(np.concatenate(out), ref)
    assert np.array_equal(np.concatenate(mags), ref_mag)


def test_many_channels_distinct_data(capi, oracle):
    n_ch = 64
    u8 = np.stack([synth.fm_tone(2 * 16384, seed=1000 + c, deviation=5000.0 + 900 * c) for c in range(n_ch)])
    eng = capi.Engine(n_ch)
    eng.set_mode("wbfm")
    pcm, cnt, mag, _ = eng.accept(u8)
    for c in range(n_ch):
        ref, ref_mag, _ = oracle_pcm(oracle, u8[c])
        assert np.array_equal(pcm[c, :cnt[c]], ref), c
        assert np.array_equal(mag[c], ref_mag), c


def test_long_single_channel_tiles_and_handoffs(capi, oracle):
    """2^22 samples in one call: ~70 tiles per launch, every hand-off verified bit-exact."""
    u8 = synth.fm_tone(1 << 22, seed=1234)
    ref, _, _ = oracle_pcm(oracle, u8)
    eng = capi.Engine(1)
    eng.set_mode("wbfm")
    pcm, cnt, _, _ = eng.accept(u8)
    assert np.array_equal(pcm[0, :cnt[0]], ref)
    st = eng.stats()
    assert st["state_checks"] >= 60 and st["state_repairs"] == 0


def test_rotation_and_reset(capi, oracle):
    u8 = synth.fm_tone(2 * 16384, seed=12)
    for rot in (1, 0, -1):
        eng = capi.Engine(1)
        eng.set_mode("wbfm")
        eng.set_rotation(rot)
        c = oracle.chain()
        c.set_mode("wbfm")
        c.set_rotation(rot)
        a, _, _ = c.accept_stream(u8[:32768])
        c.reset()
        b, _, _ = c.accept_stream(u8[32768:])
        pa, ca, _, _ = eng.accept(u8[:32768])
        eng.reset()
        pb, cb, _, _ = eng.accept(u8[32768:])
        assert np.array_equal(pa[0, :ca[0]], a) and np.array_equal(pb[0, :cb[0]], b), rot


def test_argument_errors(capi):
    eng = capi.Engine(2)
    with pytest.raises(capi.IqdError):
        eng.set_mode(9)
    with pytest.raises(capi.IqdError):
        eng.accept(np.zeros((2, 1000), np.uint8))       # not a multiple of block_bytes
    with pytest.raises(capi.IqdError):
        eng.set_mode("wbfm", first=1, n=5)               # range past the last channel
