"""CPU: the oracle (oracle/iqd_oracle.c) against the committed golden vectors, which the
unmodified reference produced (tests/golden/make_golden.py).  Bit-exact everywhere."""
import numpy as np
import pytest

MODES = ["none", "am", "fm", "wbfm", "lsb", "usb"]
STREAMS = ["fm_tone", "am_tone", "ssb_tone", "white", "rails", "capture_excerpt"]   # the last: an off-air capture


@pytest.mark.parametrize("name", STREAMS)
@pytest.mark.parametrize("mode", MODES)
def test_pcm_matches_reference(oracle, golden, name, mode):
    g = golden[name]
    c = oracle.chain()
    c.set_mode(mode)
    pcm, mag, allowed = c.accept_stream(g["iq"], int(g["block_bytes"]))
    assert np.array_equal(pcm, g["pcm_" + mode])
    assert np.array_equal(mag, g["magnitude"])
    assert np.array_equal(allowed, g["allowed"])


@pytest.mark.parametrize("mode", MODES[1:])
def test_cast_overflow_gains(oracle, golden, mode):
    g = golden["cast_overflow"]
    c = oracle.chain()
    c.set_mode(mode)
    for which, key in [(1, "gain_am"), (2, "gain_fm"), (3, "gain_wbfm"), (4, "gain_ssb")]:
        c.set_gain(which, float(g[key]))
    pcm, _, _ = c.accept_stream(g["iq"], int(g["block_bytes"]))
    assert np.array_equal(pcm, g["pcm_" + mode])
    # the fixture really exercises the wrap: plenty of large magnitudes and both signs
    assert np.abs(pcm.astype(np.int32)).max() > 5000


@pytest.mark.parametrize("name", ["squelch_steps", "capture_gated"])
@pytest.mark.parametrize("mode", MODES)
def test_squelch_gating(oracle, golden, mode, name):
    g = golden[name]
    c = oracle.chain()
    c.set_mode(mode)
    c.set_squelch(int(g["threshold"]))
    c.set_rx_gain_db(int(g["rx_gain_db"]))
    pcm, mag, allowed = c.accept_stream(g["iq"], int(g["block_bytes"]))
    assert np.array_equal(allowed, g["allowed"])
    assert np.array_equal(mag, g["magnitude"])
    assert np.array_equal(pcm, g["pcm_" + mode])
    assert 0 < allowed.sum() < len(allowed)          # both open and closed blocks occur
    if mode != "none":
        assert len(pcm) == allowed.sum() * int(g["block_bytes"]) // 64


def test_mode_switch_without_reset(oracle, golden):
    g = golden["mode_switch"]
    c = oracle.chain()
    out = []
    for k, mode in enumerate(g["sequence"]):
        c.set_mode(str(mode))
        p, _, _ = c.accept_stream(g["iq"][k * 32768:(k + 1) * 32768])
        out.append(p)
    assert np.array_equal(np.concatenate(out), g["pcm"])


def test_block_size_invariance(oracle, golden):
    """with the squelch open the PCM does not depend on the call size."""
    g = golden["fm_tone"]
    for mode in MODES[1:]:
        for bb in (64, 4096, 8192):
            c = oracle.chain()
            c.set_mode(mode)
            pcm, _, _ = c.accept_stream(g["iq"], bb)
            assert np.array_equal(pcm, g["pcm_" + mode]), (mode, bb)


PRIMS = [("wbfm_pre", 1), ("wbfm_d1", 4), ("wbfm_d2", 4), ("audio40", 2), ("fm_tuner", 4),
         ("am_s1", 4), ("am_s2", 4), ("am_s3", 2), ("ssb_delay", 1), ("ssb_hilbert", 1)]


@pytest.mark.parametrize("name,factor", PRIMS)
def test_q15_primitives(oracle, golden, name, factor):
    g = golden["primitives"]
    h = oracle.taps_f32(name)
    hq = oracle.taps_q15(name).astype(np.int32)
    # a -32768 impulse through the reference's Q15 FIR returns -hq[k] (clamped for -32768)
    neg = g["negimp_" + name][:len(hq)].astype(np.int32)
    expect = np.minimum(-hq, 32767)
    assert np.array_equal(neg, expect)
    assert np.array_equal(oracle.decimate_q15(h, factor, g["x16"]), g["y16_" + name])
    # saturated input drives the per-MAC clamp (Decimator_int16.cc:205-218)
    assert np.array_equal(oracle.decimate_q15(h, factor, g["xsat"]), g["ysat_" + name])


def test_tap_sums(oracle):
    """SURVEY §8 a15 [verified] sums of |hq| and leading taps."""
    sums = {"am_s1": 29002, "am_s2": 34926, "am_s3": 48394, "fm_tuner": 35938, "fm_post": 36758,
            "audio40": 66852, "wbfm_pre": 54924, "wbfm_d1": 29126, "wbfm_d2": 36758,
            "ssb_hilbert": 67250, "ssb_delay": 32768}
    for name, s in sums.items():
        assert int(np.abs(oracle.taps_q15(name).astype(np.int32)).sum()) == s, name
    assert list(oracle.taps_q15("wbfm_pre")[:4]) == [-515, -1068, 305, 2036]
    assert list(oracle.taps_q15("fm_tuner")[:4]) == [135, 178, 249, 378]
    assert list(oracle.taps_q15("am_s1")[:4]) == [795, 2511, 4776, 6419]
    assert oracle.taps_q15("ssb_delay")[15] == -32768


def test_float_filters(oracle, golden):
    g = golden["primitives"]
    impulse = np.zeros(16, np.float32); impulse[0] = 1
    step = np.ones(32, np.float32)
    assert np.array_equal(oracle.fir_f32([1, 2, 3, 4, 1, 1, 1, 8], impulse), g["demo_fir_impulse"])
    assert np.array_equal(oracle.iir_f32([1.0], [0.5], step), g["demo_iir_half_step"])
    assert np.array_equal(oracle.iir_f32([1.0, -1.0], [-0.95], step), g["demo_dcblock_step"])
    assert np.array_equal(oracle.iir_f32([0.0253863, 0.0253863], [-0.9492274], g["xf"]), g["deemph_xf"])
    assert np.array_equal(oracle.iir_f32([1.0, -1.0], [-0.95], g["xf"]), g["dcblock_xf"])


def test_tables_and_casts(oracle, golden):
    import hashlib
    g = golden["primitives"]
    assert np.array_equal(np.array([oracle.dbfs(m) for m in range(300)], np.int32), g["dbfs_0_299"])
    # md5 of the libm-built 256x256 float atan2 table
    assert hashlib.md5(oracle.atan2_lut().tobytes()).hexdigest() == "e88f3b47455c059938992ead454318e6"
    assert oracle.cast_i16(40000.7) == -25536
    assert oracle.cast_i16(-40000.7) == 25536
    assert oracle.cast_i16(3.0e9) == 0 and oracle.cast_i16(-3.0e9) == 0
    assert oracle.cast_i16(float("nan")) == 0
    assert np.array_equal(oracle.rotate(g["rot_in"], +1), g["rot_up"])
    assert np.array_equal(oracle.rotate(g["rot_in"], -1), g["rot_down"])


def test_fm_theta_lut_covers_tuner_range(oracle):
    """The FM tuner decimator fed with int8 data cannot leave [-141, 141]."""
    hq = oracle.taps_q15("fm_tuner").astype(np.int64)
    assert (16384 + np.abs(hq).sum() * 128) >> 15 <= 141
    lut = oracle.fm_theta_lut(141)
    assert lut.shape == (283, 283)
    assert lut[141, 141] == 0.0                            # atan2(0, 0)
    assert lut[141, 0] == np.float32(np.pi)                # atan2(+0, -141)


# ---- AutomaticGainControl -----------------------------------------------------------
import agc_script as A  # noqa: E402


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_agc_command_scripts(oracle, golden, seed):
    """Operator commands (valid and rejected values) and block magnitudes straight into the AGC."""
    g = golden["agc"]
    flags, gains = A.replay(oracle.chain(), g["script%d_codes" % seed], g["script%d_values" % seed])
    assert np.array_equal(flags, g["script%d_flags" % seed])
    assert np.array_equal(gains, g["script%d_gains" % seed])
    assert len(set(gains.tolist())) > 10 and 0 < flags.sum() < len(flags)


def test_agc_defaults_known_answer(oracle, golden):
    """The survey's own anchor: magnitudes 5x8, 100x6, 20x4 with every default."""
    c = oracle.chain()
    c.agc_enable(True)
    out = []
    for m in [5] * 8 + [100] * 6 + [20] * 4:
        c.agc_feed(m)
        out.append(c.rx_gain_db())
    assert out == [37, 37, 46, 46, 46, 46, 46, 46, 38, 38, 30, 30, 22, 22, 25, 25, 28, 28]
    assert np.array_equal(np.array(out, np.uint32), golden["agc"]["defaults_gains"])


@pytest.mark.parametrize("case", [c[0] for c in A.STREAM_CASES])
def test_agc_inside_the_block_flow(oracle, golden, case):
    """The AGC fed by acceptIqData's magnitude callback: its gain moves the next block's squelch decision."""
    g = golden["agc"]
    cfg = dict((c[0], c[2]) for c in A.STREAM_CASES)[case]
    c = oracle.chain()
    A.configure(c, cfg)
    pcm, allowed, gains = A.stream(c, g[case + "_iq"], 4096)
    assert np.array_equal(gains, g[case + "_gains"])
    assert np.array_equal(allowed, g[case + "_allowed"])
    assert np.array_equal(pcm, g[case + "_pcm"])
    assert len(set(gains.tolist())) > 4


def test_frequency_scanner_commands_and_tuning(oracle, golden):
    """FrequencyScanner: advances on every block the squelch rejects, wraps from end to start, ignores
    parameter changes while scanning."""
    g = golden["agc"]
    flags, pcm, freq, count, final = A.scan_scenario(oracle.chain(), g["scan_iq"], 4096, A.feed_blockwise(4096))
    assert np.array_equal(flags, g["scan_flags"]) and flags.tolist() == [1, 1, 0, 0, 1, 0, 1, 1]
    assert np.array_equal(freq, g["scan_freq"]) and np.array_equal(count, g["scan_count"])
    assert np.array_equal(pcm, g["scan_pcm"]) and np.array_equal(final, g["scan_final"])
    assert count[-3] > 12 and len(set(freq.tolist())) == 4      # the scan wrapped around


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_standalone_resamplers(oracle, golden, name):
    """Float Decimator / Interpolator and Interpolator_int16: bit-identical floats, int16 clamps."""
    g = golden["resample"]
    h, f = g["h_" + name], int(g["f_" + name])
    assert np.array_equal(oracle.decimate_f32(h, f, g["x"]).view(np.uint32), g["dec_" + name].view(np.uint32))
    assert np.array_equal(oracle.interpolate_f32(h, f, g["x"]).view(np.uint32), g["int_" + name].view(np.uint32))
    assert np.array_equal(oracle.interpolate_q15(h, f, g["x16"]), g["i16_" + name])
    hs = np.clip(4 * h, -1, 0.99997).astype(np.float32)
    assert np.array_equal(oracle.interpolate_q15(hs, f, g["xsat"]), g["i16sat_" + name])


@pytest.mark.parametrize("mode", ["am", "fm", "wbfm", "lsb", "usb"])
def test_demodulator_level_entry(oracle, golden, mode):
    """{Am,Fm,WbFm,Ssb}Demodulator::acceptIqData(int8_t *, uint32_t) interleaved with the processor's acceptIqData on the
    same demodulator objects (tests/golden/make_golden.py: demod_entry)."""
    g = golden["demod_entry"]
    c = oracle.chain()
    c.set_mode(mode)
    for k, (kind, n) in enumerate(zip(g["kinds"], g["lengths"])):
        if kind == "proc":
            pcm, _, _ = c.accept_stream(g["in%d" % k], int(n))
        else:
            pcm = c.demod_accept(mode, g["in%d" % k])
        assert np.array_equal(pcm, g["pcm_%s_%d" % (mode, k)]), (mode, k)


@pytest.mark.parametrize("dtype", [1, 2, 3, 4, 5])
def test_the_reference_demod_program(oracle, golden, dtype):
    """tests/golden/demod_tool.npz = stdout of the reference's own offline harness as a program (demodulatorResearch/
    demodulators/demod.cc, built unmodified by oracle/Makefile, `-d <type>`): signed bytes, 16384 per read, into a bare
    demodulator.  The program's switch has no break, so -d 4 (LSB) falls through to USB: types 4 and 5 are both USB."""
    g = golden["demod_tool"]
    mode = {1: "am", 2: "fm", 3: "wbfm", 4: "usb", 5: "usb"}[dtype]
    c = oracle.chain()
    x = g["iq_s8"]
    pcm = np.concatenate([c.demod_accept(mode, x[o:o + 16384]) for o in range(0, len(x), 16384)])
    assert np.array_equal(pcm, g["pcm_%d" % dtype])
    if dtype == 4:
        c = oracle.chain()
        lsb = np.concatenate([c.demod_accept("lsb", x[o:o + 16384]) for o in range(0, len(x), 16384)])
        assert not np.array_equal(lsb, g["pcm_4"])          # (what -d 4 was meant to be is something else)
