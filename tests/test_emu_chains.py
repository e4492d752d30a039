"""CPU: the FM / AM / SSB tile kernels' phase functions (rtlsdrdiags_amd/csrc/iqd_chains.h), stepped
on the host by tests/emu, against the oracle."""
import numpy as np
import pytest

from rtlsdrdiags_amd import synth
from tests import emu_bind

MODES = ["am", "fm", "lsb", "usb"]


@pytest.fixture(scope="module")
def emu():
    return emu_bind.lib()


def oracle_run(oracle, mode, u8, rotation=1, gain=None):
    c = oracle.chain()
    c.set_mode(mode)
    c.set_rotation(rotation)
    if gain is not None:
        c.set_gain({"am": 1, "fm": 2, "lsb": 4, "usb": 4}[mode], gain)
    return c.accept_stream(u8)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("call_samples,tile_len", [(4 * 16384, 65536), (4 * 16384, 16384), (16384, 8192),
                                                   (4096, 8192), (128, 8192), (1152, 384), (16384, 2048), (16384, 4096)])
def test_calls_and_tiles(emu, oracle, mode, call_samples, tile_len):
    u8 = synth.fm_tone(4 * 16384, seed=41) if mode == "fm" else synth.am_tone(4 * 16384, seed=42)
    ref, ref_mag, _ = oracle_run(oracle, mode, u8)
    ch = emu_bind.FirChannel(emu, mode, tile_len)
    out = [ch.accept(u8[2 * o:2 * (o + call_samples)])[0] for o in range(0, len(u8) // 2, call_samples)]
    assert np.array_equal(np.concatenate(out), ref)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("kind", ["white", "rails", "ssb"])
def test_edge_inputs_and_magnitude(emu, oracle, mode, kind):
    u8 = {"white": synth.white_u8(2 * 16384, seed=3), "rails": synth.rails_u8(2 * 16384, seed=4),
          "ssb": synth.ssb_tone(2 * 16384, seed=5)}[kind]
    ref, ref_mag, _ = oracle_run(oracle, mode, u8)
    ch = emu_bind.FirChannel(emu, mode, 16384)
    pcm, mag = ch.accept(u8)
    assert np.array_equal(pcm, ref)
    assert np.array_equal(mag // 16384, ref_mag)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("rotation", [0, -1])
def test_rotation_and_overflowing_gain(emu, oracle, mode, rotation):
    gain = {"am": 30000.0, "fm": 2.0e6, "lsb": 30000.0, "usb": 30000.0}[mode]
    u8 = synth.fm_tone(2 * 16384, seed=6, amplitude=100.0)
    ref, _, _ = oracle_run(oracle, mode, u8, rotation=rotation, gain=gain)
    ch = emu_bind.FirChannel(emu, mode, 8192, rotation=rotation, gain=gain)
    out = [ch.accept(u8[:32768])[0], ch.accept(u8[32768:])[0]]
    assert np.array_equal(np.concatenate(out), ref)


def test_fm_unbounded_cast(emu, oracle):
    u8 = synth.fm_tone(16384, seed=10, amplitude=100.0)
    ref, _, _ = oracle_run(oracle, "fm", u8, gain=4.0e9)
    pcm, _ = emu_bind.FirChannel(emu, "fm", 8192, gain=4.0e9).accept(u8)
    assert np.array_equal(pcm, ref)


@pytest.mark.parametrize("mode", ["am", "usb"])
@pytest.mark.parametrize("n_samples", [278528, 266368])
def test_long_stream_segmented_dc_blocker(emu, oracle, mode, n_samples):
    """More than one pass of the wave-per-channel DC-removal kernel, with a partial last pass and a
    last segment shorter than 128 PCM samples."""
    u8 = synth.am_tone(n_samples, seed=51, depth=0.8)
    c = oracle.chain()
    c.set_mode(mode)
    ref = np.concatenate([c.accept_stream(u8[o:o + 32768])[0] for o in range(0, len(u8) - 32767, 32768)]
                         + ([c.accept_stream(u8[len(u8) // 32768 * 32768:], 256)[0]] if len(u8) % 32768 else []))
    ch = emu_bind.FirChannel(emu, mode, 65536)
    pcm, _ = ch.accept(u8)
    assert np.array_equal(pcm, ref)


def _loud_quiet(seed, tone=3000.0):
    """Alternating wide- and narrow-deviation FM so that the audio decimators switch between their
    clamp-capable and clamp-free evaluation inside and across chunks."""
    parts = []
    for k, dev in enumerate([70000.0, 1500.0, 74000.0, 800.0, 1500.0, 72000.0]):
        parts.append(synth.fm_tone(5632 + 1024 * k, seed=seed + k, deviation=dev, tone=tone, amplitude=110.0, sigma=0.5))
    u8 = np.concatenate(parts)
    return u8[: (len(u8) // 256) * 256]


def test_fm_loud_and_quiet_sections(emu, oracle):
    u8 = _loud_quiet(60)
    ref, _, _ = oracle_run(oracle, "fm", u8, gain=20000.0)
    pcm, _ = emu_bind.FirChannel(emu, "fm", 16384, gain=20000.0).accept(u8)
    assert np.array_equal(pcm, ref)
    a = np.abs(ref.astype(np.int32))
    assert a.max() > 8000 and min(a[k:k + 40].max() for k in range(0, len(a) - 40, 20)) < 8000


def test_device_agc_step_on_the_host():
    """The AGC step the squelch kernel runs per block (iqd_chains.h: agc_run), compiled for the host, against the
    oracle: runs of magnitudes between the commands of the random scripts (every parameter combination)."""
    import ctypes as C
    import agc_script as A
    from oracle import bindings as B
    L = emu_bind.lib()
    u32p, f32p = C.POINTER(C.c_uint32), C.POINTER(C.c_float)
    L.emu_agc_run.argtypes = [C.c_uint32, C.c_int32, C.c_int32, C.c_float, C.c_uint32, u32p, u32p, f32p, u32p, u32p,
                              C.c_void_p, C.c_uint32, C.c_void_p]
    O = B.Oracle()
    rng = np.random.default_rng(9)
    for trial in range(60):
        typ, dead, blank = int(rng.integers(0, 2)), int(rng.integers(0, 11)), int(rng.integers(0, 11))
        alpha, op = float(np.float32(rng.choice([0.001, 0.05, 0.3, 0.8, 0.998]))), int(rng.integers(-40, 1))
        gain0 = int(rng.integers(0, 47))
        mags = np.clip(rng.normal(40, 40, 300), 0, 191).astype(np.uint32)
        o = O.chain()
        o.agc_set_type(typ); o.agc_set_deadband(dead); o.agc_set_blanking_limit(blank)
        o.agc_set_filter_coefficient(alpha); o.agc_set_operating_point(op); o.set_rx_gain_db(gain0)
        o.agc_enable(True)
        want = []
        for m in mags:
            o.agc_feed(int(m))
            want.append(o.rx_gain_db())
        rx, ifg, filt = C.c_uint32(gain0), C.c_uint32(24), C.c_float(24.0)
        bc, adj = C.c_uint32(0), C.c_uint32(0)
        got = np.zeros(len(mags), np.uint32)
        L.emu_agc_run(typ, op, dead, alpha, blank, C.byref(rx), C.byref(ifg), C.byref(filt), C.byref(bc), C.byref(adj),
                      mags.ctypes.data, len(mags), got.ctypes.data)
        assert got.tolist() == want, trial


def test_dc_pass_in_place_over_int16_rows_is_the_serial_recurrence():
    """Round 6: behind the streaming pipelines the AM / SSB DC-removal pass runs IN PLACE over int16 detector values in the PCM row,
    with a segment's LDS row holding first its input differences, then its PCM (iqd_chains.h: dc_block_wave).  Stepped on the
    host against the plain serial recurrence (dc_block_run = AmDemodulator.cc:462-465) for rows of one to five passes, ragged
    ends, gains from 1e-3 to 1e5, constant rows (the state sticks at a denormal: the tiny-state rule) and a carried state."""
    import ctypes as C
    from tests import emu_bind
    L = emu_bind.lib()

    class Dc(C.Structure):
        _fields_ = [("x_prev", C.c_float), ("y_prev", C.c_float)]
    L.emu_dc_row.restype = None
    L.emu_dc_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    L.emu_dc_serial.restype = None
    L.emu_dc_serial.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    rng = np.random.default_rng(2026)
    for trial in range(40):
        n = 4 * int(rng.integers(1, 2700))                       # PCM samples: any multiple of 4, up to five passes of 2048
        kind = trial % 4
        x = (rng.integers(-546, 547, n) if kind < 2 else np.full(n, int(rng.integers(-546, 547))) if kind == 2
             else np.round(300 * np.sin(np.arange(n) * 0.05) + rng.normal(0, 3, n))).astype(np.int32)
        gain = float(10.0 ** rng.uniform(-3, 5)) * (1 if trial % 5 else -1)
        st0 = (float(int(rng.integers(-546, 547))), float(rng.normal(0, 200))) if trial % 3 else (0.0, 0.0)
        want, got16, got32 = np.zeros(n, np.int16), x.astype(np.int16).copy(), np.zeros(n, np.int16)
        s_ref, s_a, s_b = Dc(*st0), Dc(*st0), Dc(*st0)
        L.emu_dc_serial(x.ctypes.data, want.ctypes.data, n, gain, C.byref(s_ref))
        L.emu_dc_row(1, None, got16.ctypes.data, n, gain, C.byref(s_a))              # in place over the int16 row
        L.emu_dc_row(0, x.ctypes.data, got32.ctypes.data, n, gain, C.byref(s_b))     # from the int32 stream (the tile kernels' path)
        assert np.array_equal(got16, want), (trial, n, gain, np.flatnonzero(got16 != want)[:5])
        assert np.array_equal(got32, want), (trial, n, gain)
        for s in (s_a, s_b):
            assert s.x_prev == s_ref.x_prev and (s.y_prev == s_ref.y_prev or abs(s.y_prev) < 1e-30 and abs(s_ref.y_prev) < 1e-30), (trial, s.y_prev, s_ref.y_prev)
