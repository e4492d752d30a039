"""CPU: the WBFM kernel's phase functions (rtlsdrdiags_amd/csrc/iqd_wbfm.h), stepped on the host
by tests/emu, against the oracle: tiling, lead-in, exact hand-off, state carried between calls."""
import numpy as np
import pytest

from rtlsdrdiags_amd import synth
from tests import emu_bind


@pytest.fixture(scope="module", params=["pipelined", "serial_phases"])
def emu(request):
    """Both tile drivers: the pipelined one the kernel uses and the first, phase-by-phase one."""
    L = emu_bind.lib()
    L.emu_wbfm_driver(1 if request.param == "serial_phases" else 0)
    yield L
    L.emu_wbfm_driver(0)


def oracle_wbfm(oracle, u8, rotation=1, gain=None):
    c = oracle.chain()
    c.set_mode("wbfm")
    c.set_rotation(rotation)
    if gain is not None:
        c.set_gain(3, gain)
    return c.accept_stream(u8)


def test_lds_fits_three_workgroups_per_cu(emu):
    assert 3 * emu.emu_lds_bytes() <= 160 * 1024


@pytest.mark.parametrize("call_samples,tile_len", [(6 * 16384, 61440), (6 * 16384, 16384), (16384, 16384),
                                                   (16384, 8192), (4096, 7680), (128, 7680), (384, 128)])
def test_calls_and_tiles(emu, oracle, call_samples, tile_len):
    u8 = synth.fm_tone(6 * 16384, seed=21)
    ref, ref_mag, _ = oracle_wbfm(oracle, u8)
    ch = emu_bind.WbfmChannel(emu, tile_len)
    out = [ch.accept(u8[2 * o:2 * (o + call_samples)])[0] for o in range(0, len(u8) // 2, call_samples)]
    assert np.array_equal(np.concatenate(out), ref)
    assert ch.hand_off_mismatches == 0


@pytest.mark.parametrize("kind", ["white", "rails"])
def test_edge_inputs_and_magnitude(emu, oracle, kind):
    u8 = synth.white_u8(3 * 16384, seed=3) if kind == "white" else synth.rails_u8(2 * 16384, seed=4)
    ref, ref_mag, _ = oracle_wbfm(oracle, u8)
    ch = emu_bind.WbfmChannel(emu, 16384)
    pcm, mag = ch.accept(u8)
    assert np.array_equal(pcm, ref)
    assert np.array_equal(mag // 16384, ref_mag)
    assert ch.hand_off_mismatches == 0


@pytest.mark.parametrize("scale", [0.0, 3.0e4, -1.0e5])
def test_bad_state_guess_is_repaired_exactly(emu, oracle, scale):
    """The segment state guess only affects speed: whatever it is, the output is exact."""
    u8 = synth.fm_tone(2 * 16384, seed=5)
    ref, _, _ = oracle_wbfm(oracle, u8)
    ch = emu_bind.WbfmChannel(emu, 16384, guess_scale=scale)
    pcm, _ = ch.accept(u8)
    assert np.array_equal(pcm, ref)
    assert ch.segment_repairs > 5


@pytest.mark.parametrize("rotation", [1, 0, -1])
def test_rotation_and_cast_overflow_gain(emu, oracle, rotation):
    u8 = synth.fm_tone(2 * 16384, seed=9, amplitude=100.0)
    ref, _, _ = oracle_wbfm(oracle, u8, rotation=rotation, gain=8.0e6)
    ch = emu_bind.WbfmChannel(emu, 8192, rotation=rotation, gain=8.0e6)
    out = [ch.accept(u8[:32768])[0], ch.accept(u8[32768:])[0]]
    assert np.array_equal(np.concatenate(out), ref)


def test_unbounded_cast_path(emu, oracle):
    """A gain so large that y leaves the int32 range: the slow cast with x86 'indefinite' rules."""
    u8 = synth.fm_tone(16384, seed=10, amplitude=100.0)
    ref, _, _ = oracle_wbfm(oracle, u8, gain=6.0e9)
    ch = emu_bind.WbfmChannel(emu, 8192, gain=6.0e9)
    pcm, _ = ch.accept(u8)
    assert np.array_equal(pcm, ref)


def test_reset_keeps_deemphasis_state(emu, oracle):
    u8 = synth.fm_tone(2 * 16384, seed=12)
    c = oracle.chain()
    c.set_mode("wbfm")
    a, _, _ = c.accept_stream(u8[:32768])
    c.reset()
    b, _, _ = c.accept_stream(u8[32768:])
    ch = emu_bind.WbfmChannel(emu, 8192)
    pa, _ = ch.accept(u8[:32768])
    ch.reset()
    pb, _ = ch.accept(u8[32768:])
    assert np.array_equal(pa, a) and np.array_equal(pb, b)


def test_loud_and_quiet_sections(emu, oracle):
    """The 40-tap audio decimator switches between the clamp-capable and the clamp-free evaluation."""
    from tests.test_emu_chains import _loud_quiet
    u8 = _loud_quiet(70, tone=700.0)
    for gain in (90000.0, None):
        ref, _, _ = oracle_wbfm(oracle, u8, gain=gain)
        ch = emu_bind.WbfmChannel(emu, 8192, gain=gain)
        pcm, _ = ch.accept(u8)
        assert np.array_equal(pcm, ref)
    a = np.abs(ref.astype(np.int32))
    assert a.max() > 17000 and min(a[k:k + 40].max() for k in range(0, len(a) - 40, 20)) < 3000


# ---- calls in 64-byte units (32 samples): IqDataProcessor.cc:586 strides 8 bytes, WbFmDemodulator.cc:383-411 takes
# whatever it is given; round 4 lifts the 256-byte unit the WBFM chain used to want ---------------------------------
RAGGED = [
    [32, 32, 32, 32, 96, 160, 4128, 64, 8224, 32, 16384, 992, 32],     # from a fresh stream: restart point off the grid at once
    [16384, 96, 16384, 32, 32, 7680 + 64, 128, 2016],                  # long, short, long
    [736, 32, 32, 32, 32, 32, 32, 7040 + 96, 14080 + 32],              # across the 768-sample restart distance, across chunk ends
]


@pytest.mark.parametrize("sizes", RAGGED)
@pytest.mark.parametrize("tile_len", [7040, 16384])
def test_calls_of_any_multiple_of_32_samples(emu, oracle, sizes, tile_len):
    total = sum(sizes)
    u8 = synth.fm_tone(total, seed=33)
    c = oracle.chain()
    c.set_mode("wbfm")
    ch = emu_bind.WbfmChannel(emu, tile_len)
    off = 0
    for k, m in enumerate(sizes):
        part = u8[2 * off:2 * (off + m)]
        ref, ref_mag, _ = c.accept_stream(part, len(part) if len(part) < 32768 else 32768)
        pcm, _ = ch.accept(part)
        assert np.array_equal(pcm, ref), (k, m)
        assert ch.carry.back % 32 == 0 and 0 < ch.carry.back <= 768 + 96
        off += m
    assert ch.hand_off_mismatches == 0


def test_ragged_calls_around_a_reset(emu, oracle):
    """resetDemodulator() between calls of odd lengths: the restart point falls back to the reset point (back = what has
    been consumed since), off the 128-sample grid."""
    u8 = synth.fm_tone(40000, seed=34)
    c = oracle.chain()
    c.set_mode("wbfm")
    ch = emu_bind.WbfmChannel(emu, 7040)
    off = 0
    for k, m in enumerate([4128, 96, 32, 160, 32, 32, 7072, 352, 32, 9000 // 32 * 32]):
        if k in (2, 7):
            c.reset()
            ch.reset()
        part = u8[2 * off:2 * (off + m)]
        ref, _, _ = c.accept_stream(part, min(len(part), 32768))
        pcm, _ = ch.accept(part)
        assert np.array_equal(pcm, ref), (k, m)
        off += m


def test_restart_point_is_never_further_back_than_the_streaming_lead_in(emu, oracle):
    """The streaming kernel starts a call's first segment 768 samples back and applies the carried exact state where it
    finds it at or after that point (iqd_stream.hip: st_iir_marks).  A tile kernel that ended a call off the 128-sample
    grid used to leave its restart record at the last SEGMENT BOUNDARY at or before tlen - 768, i.e. up to 96 samples too
    far back: the streaming kernel then never met it and ran the call from a zero state (round 4's fuzzer, with the streaming
    path pinned: one PCM sample off by one in thousands of cases).  Now the record is taken exactly at tlen - 768."""
    u8 = synth.fm_tone(60000, seed=35)
    for tile_len in (7040, 2048):
        c = oracle.chain()
        c.set_mode("wbfm")
        ch = emu_bind.WbfmChannel(emu, tile_len)
        off = consumed = 0
        for m in [832, 512, 800, 1024, 32, 64, 7072, 96, 2080, 4128, 1760, 128]:
            part = u8[2 * off:2 * (off + m)]
            ref, _, _ = c.accept_stream(part, min(len(part), 32768))
            pcm, _ = ch.accept(part)
            assert np.array_equal(pcm, ref), (tile_len, m)
            off += m
            consumed += m
            assert ch.carry.back == min(consumed, 768), (tile_len, m, ch.carry.back)
