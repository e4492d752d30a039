"""GPU (MI355X): every demodulator mode through the C ABI against the golden vectors (produced by the
unmodified reference) and the oracle: all modes, mixed-mode batches, mode switches, squelch gating."""
import numpy as np
import pytest

from rtlsdrdiags_amd import synth

pytestmark = pytest.mark.gpu
MODES = ["none", "am", "fm", "wbfm", "lsb", "usb"]
DEMOD_OF = {"am": "am", "fm": "fm", "wbfm": "wbfm", "lsb": "ssb", "usb": "ssb"}


@pytest.fixture(scope="module")
def capi():
    from rtlsdrdiags_amd import capi as c
    return c


@pytest.mark.parametrize("name", ["fm_tone", "am_tone", "ssb_tone", "white", "rails", "capture_excerpt"])
@pytest.mark.parametrize("mode", MODES)
def test_golden_all_modes(capi, golden, name, mode):
    g = golden[name]
    eng = capi.Engine(1)
    eng.set_mode(mode)
    pcm, cnt, mag, allowed = eng.accept(g["iq"])
    assert cnt[0] == len(g["pcm_" + mode])
    assert np.array_equal(pcm[0, :cnt[0]], g["pcm_" + mode])
    assert np.array_equal(mag[0], g["magnitude"])
    assert np.array_equal(allowed[0], g["allowed"])


@pytest.mark.parametrize("mode", MODES[1:])
def test_golden_cast_overflow_gains(capi, golden, mode):
    g = golden["cast_overflow"]
    eng = capi.Engine(1)
    eng.set_mode(mode)
    for demod, key in [("am", "gain_am"), ("fm", "gain_fm"), ("wbfm", "gain_wbfm"), ("ssb", "gain_ssb")]:
        eng.set_gain(demod, float(g[key]))
    pcm, cnt, _, _ = eng.accept(g["iq"])
    assert np.array_equal(pcm[0, :cnt[0]], g["pcm_" + mode])


@pytest.mark.parametrize("name", ["squelch_steps", "capture_gated"])
@pytest.mark.parametrize("mode", MODES)
def test_golden_squelch_gating(capi, golden, mode, name):
    """Blocks that Squelch::run() rejects are skipped entirely: no PCM, no state advance.  capture_gated is the
    reference's own recording with a real dropout of the carrier (tests/golden/make_golden.py: capture)."""
    g = golden[name]
    eng = capi.Engine(1, block_bytes=int(g["block_bytes"]))
    eng.set_mode(mode)
    eng.set_squelch(int(g["threshold"]))
    eng.set_rx_gain_db(int(g["rx_gain_db"]))
    pcm, cnt, mag, allowed = eng.accept(g["iq"])
    assert np.array_equal(allowed[0], g["allowed"])
    assert np.array_equal(mag[0], g["magnitude"])
    assert cnt[0] == len(g["pcm_" + mode])
    assert np.array_equal(pcm[0, :cnt[0]], g["pcm_" + mode])


@pytest.mark.parametrize("mode", ["wbfm", "am", "usb"])
def test_squelch_gating_across_calls(capi, golden, mode):
    """The tracker's one-block tail and the frozen filter state survive call boundaries."""
    g = golden["squelch_steps"]
    bb = int(g["block_bytes"])
    eng = capi.Engine(1, block_bytes=bb)
    eng.set_mode(mode)
    eng.set_squelch(int(g["threshold"]))
    eng.set_rx_gain_db(int(g["rx_gain_db"]))
    out, flags = [], []
    for off in range(0, len(g["iq"]), 4 * bb):
        pcm, cnt, _, allowed = eng.accept(g["iq"][off:off + 4 * bb])
        out.append(pcm[0, :cnt[0]])
        flags.append(allowed[0])
    assert np.array_equal(np.concatenate(flags), g["allowed"])
    assert np.array_equal(np.concatenate(out), g["pcm_" + mode])


def test_golden_mode_switch_without_reset(capi, golden):
    g = golden["mode_switch"]
    eng = capi.Engine(1)
    out = []
    for k, mode in enumerate(g["sequence"]):
        eng.set_mode(str(mode))
        pcm, cnt, _, _ = eng.accept(g["iq"][k * 32768:(k + 1) * 32768])
        out.append(pcm[0, :cnt[0]])
    assert np.array_equal(np.concatenate(out), g["pcm"])


def test_mixed_mode_batch(capi, oracle):
    """BASELINE config 4 in miniature: ch % 5 -> {AM, FM, WBFM, LSB, USB}, distinct data per channel,
    two consecutive calls, per-channel gains."""
    n_ch = 40
    order = ["am", "fm", "wbfm", "lsb", "usb"]
    u8 = np.stack([synth.fm_tone(4 * 16384, seed=2000 + c, deviation=3000.0 + 1500 * c) if c % 2 else
                   synth.am_tone(4 * 16384, seed=2000 + c, tone=400.0 + 50 * c) for c in range(n_ch)])
    eng = capi.Engine(n_ch)
    refs = []
    for c in range(n_ch):
        mode = order[c % 5]
        gain = {"am": 300.0, "fm": 10185.9, "wbfm": 40743.7, "lsb": 300.0, "usb": 300.0}[mode] * (0.5 + 0.05 * c)
        eng.set_mode(mode, first=c, n=1)
        eng.set_gain(DEMOD_OF[mode], gain, first=c, n=1)
        o = oracle.chain()
        o.set_mode(mode)
        o.set_gain({"am": 1, "fm": 2, "wbfm": 3, "ssb": 4}[DEMOD_OF[mode]], gain)
        refs.append(o.accept_stream(u8[c])[0])
    half = u8.shape[1] // 2
    p1, c1, _, _ = eng.accept(u8[:, :half])
    p2, c2, _, _ = eng.accept(u8[:, half:])
    for c in range(n_ch):
        got = np.concatenate([p1[c, :c1[c]], p2[c, :c2[c]]])
        assert np.array_equal(got, refs[c]), (c, order[c % 5])


@pytest.mark.parametrize("mode", ["am", "fm", "lsb"])
def test_long_stream_many_tiles(capi, oracle, mode):
    u8 = synth.am_tone(1 << 20, seed=77) if mode != "fm" else synth.fm_tone(1 << 20, seed=77)
    c = oracle.chain()
    c.set_mode(mode)
    ref, _, _ = c.accept_stream(u8)
    eng = capi.Engine(1)
    eng.set_mode(mode)
    pcm, cnt, _, _ = eng.accept(u8)
    assert np.array_equal(pcm[0, :cnt[0]], ref)


def test_reset_all_modes(capi, oracle):
    u8 = synth.fm_tone(2 * 16384, seed=12)
    for mode in MODES[1:]:
        eng = capi.Engine(1)
        eng.set_mode(mode)
        c = oracle.chain()
        c.set_mode(mode)
        a, _, _ = c.accept_stream(u8[:32768])
        c.reset()
        b, _, _ = c.accept_stream(u8[32768:])
        pa, ca, _, _ = eng.accept(u8[:32768])
        eng.reset()
        pb, cb, _, _ = eng.accept(u8[32768:])
        assert np.array_equal(pa[0, :ca[0]], a) and np.array_equal(pb[0, :cb[0]], b), mode


def test_file_tool_and_cpp_class(golden, tmp_path):
    """iqdemod_file = file source + IqDataProcessor C++ class + S16_LE sink (BASELINE config 1 plumbing,
    with the arithmetic on the GPU): PCM bytes on stdout equal the reference's, mode by mode."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "rtlsdrdiags_amd", "bin", "iqdemod_file")
    g = golden["fm_tone"]
    for mode, name in enumerate(MODES):
        out = subprocess.run([tool, str(mode)], input=g["iq"].tobytes(), stdout=subprocess.PIPE, check=True).stdout
        assert np.array_equal(np.frombuffer(out, dtype=np.int16), g["pcm_" + name]), name
    g = golden["squelch_steps"]     # the class takes 32768-byte blocks; squelch with a threshold
    out = subprocess.run([tool, "2", "-40"], input=golden["white"]["iq"].tobytes(), stdout=subprocess.PIPE,
                         check=True).stdout
    assert np.array_equal(np.frombuffer(out, dtype=np.int16), golden["white"]["pcm_fm"])


def test_file_tool_with_agc(oracle):
    """The AutomaticGainControl mirror class around the same tool: PCM and the final IF gain equal the oracle's."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "rtlsdrdiags_amd", "bin", "iqdemod_file")
    amps = [3, 3, 50, 50, 50, 110, 110, 4, 4, 4, 70, 70]
    u8 = synth.stepped_amplitude(amps, block_samples=16384, seed=5)
    for agc_type in (0, 1):
        o = oracle.chain()
        o.set_mode("fm")
        o.set_squelch(-50)
        o.agc_set_type(agc_type)
        o.agc_enable(True)
        ref, _, allowed = o.accept_stream(u8)
        r = subprocess.run([tool, "2", "-50", str(agc_type)], input=u8.tobytes(), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, check=True)
        assert np.array_equal(np.frombuffer(r.stdout, dtype=np.int16), ref)
        assert ("IF gain: %d dB" % o.rx_gain_db()) in r.stderr.decode()
        assert 0 < allowed.sum() < len(allowed)


@pytest.mark.parametrize("mode", ["am", "lsb"])
def test_many_channels_dc_blocker_batches(capi, oracle, mode):
    """200 channels (not a multiple of 64) over three calls: the batched DC-removal kernel and its
    carried state."""
    n_ch = 200
    u8 = np.stack([synth.am_tone(3 * 16384, seed=3000 + c, tone=300.0 + 11 * c, depth=0.3 + 0.003 * c)
                   for c in range(n_ch)])
    eng = capi.Engine(n_ch)
    eng.set_mode(mode)
    outs = []
    for k in range(3):
        pcm, cnt, _, _ = eng.accept(u8[:, k * 32768:(k + 1) * 32768])
        outs.append(pcm)
    got = np.concatenate(outs, axis=1)
    for c in range(0, n_ch, 7):
        o = oracle.chain()
        o.set_mode(mode)
        assert np.array_equal(got[c], o.accept_stream(u8[c])[0]), c


@pytest.mark.parametrize("mode", ["am", "usb"])
def test_long_row_dc_removal_many_waves(capi, oracle, mode):
    """One channel, 2^22 samples in two unequal calls: the 8 kS/s DC-removal recurrence is cut into tiles that
    warm up from zero and must chain up bit for bit (dc_tiled_kernel / dc_chainup_kernel)."""
    n = 1 << 22
    u8 = np.tile(synth.am_tone(1 << 20, seed=71, tone=440.0, depth=0.6), 4) if mode == "am" else \
        np.tile(synth.ssb_tone(1 << 20, seed=72), 4)
    cut = 2 * (3 << 19)           # bytes: 1.5 * 2^20 samples
    eng = capi.Engine(1)
    eng.set_mode(mode)
    pa, ca, _, _ = eng.accept(u8[:cut])
    pb, cb, _, _ = eng.accept(u8[cut:])
    o = oracle.chain()
    o.set_mode(mode)
    ref, _, _ = o.accept_stream(u8)
    assert ca[0] + cb[0] == len(ref) == n // 32
    assert np.array_equal(np.concatenate([pa[0, :ca[0]], pb[0, :cb[0]]]), ref)


def test_long_row_dc_removal_with_a_decaying_tail(capi, oracle):
    """A noiseless carrier after a modulated stretch: the detector input is constant and the true state decays
    through the denormals, where it sticks for ever, while the next tile's zero-state warm-up sits at exactly 0.
    Bit equality would fail at every later boundary (and send the row to the one-wave pass); both states are far
    below 2^-100 there - the warm-up is 2048 steps of a 0.95 contraction - so the hand-offs count as agreeing.
    Also pins denormal arithmetic on the device: the PCM must be the oracle's."""
    n8k_tiles, t0 = 4, 2 * 8192 - 2100                       # 8 kS/s index where the modulation stops
    n = 32 * 8192 * n8k_tiles
    u8 = synth.am_tone(n, seed=73, tone=700.0, depth=0.8, sigma=2.0).reshape(-1, 2).copy()
    carrier = np.array([[60, 0], [0, -60], [-60, 0], [0, 60]], np.int16)       # A * exp(-j pi n / 2)
    k0 = 32 * t0
    u8[k0:] = (128 + np.tile(carrier, ((n - k0) // 4, 1))).astype(np.uint8)
    u8 = u8.reshape(-1)
    eng = capi.Engine(1)
    eng.set_mode("am")
    eng.set_profiling(True)       # makes the call read the device counters back
    pcm, cnt, _, _ = eng.accept(u8)
    assert eng.stats()["state_repairs"] == 0          # nothing had to be redone
    o = oracle.chain()
    o.set_mode("am")
    ref, _, _ = o.accept_stream(u8)
    assert np.array_equal(pcm[0, :cnt[0]], ref)
    # a second call continues from the carried state
    more = synth.am_tone(1 << 19, seed=74)
    p2, c2, _, _ = eng.accept(more)
    r2, _, _ = o.accept_stream(more)
    assert np.array_equal(p2[0, :c2[0]], r2)


def test_long_rows_gated_dc_removal(capi, oracle):
    """Three long rows with the squelch closing on some blocks: per-channel open lengths differ."""
    nblk = 160
    rng = np.random.default_rng(8)
    eng = capi.Engine(3)
    eng.set_squelch(-42)
    rows = []
    for c, mode in enumerate(["am", "lsb", "usb"]):
        amps = [int(a) for a in np.where(rng.random(nblk) < 0.6, 80, 2)]
        rows.append(synth.stepped_amplitude(amps, block_samples=16384, seed=90 + c))
        eng.set_mode(mode, first=c, n=1)
    iq = np.stack(rows)
    pcm, cnt, mag, allowed = eng.accept(iq)
    for c, mode in enumerate(["am", "lsb", "usb"]):
        o = oracle.chain()
        o.set_mode(mode)
        o.set_squelch(-42)
        ref, rmag, rallowed = o.accept_stream(iq[c])
        assert np.array_equal(allowed[c], rallowed) and np.array_equal(mag[c], rmag)
        assert cnt[c] == len(ref) and cnt[c] > 8192 and np.array_equal(pcm[c, :cnt[c]], ref), c


def test_file_tool_scanner_and_iq_dump(oracle, tmp_path):
    """FrequencyScanner mirror class and the IQ dump tap through the tool: the rotated signed bytes equal the
    oracle's front end, the last tuning command and the command count equal the oracle's scanner."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "rtlsdrdiags_amd", "bin", "iqdemod_file")
    amps = [3, 3, 50, 50, 3, 3, 3, 110, 4, 4, 4, 70]
    u8 = synth.stepped_amplitude(amps, block_samples=16384, seed=6)
    o = oracle.chain()
    o.set_mode("usb")
    o.set_squelch(-45)
    o.scanner_set_parameters(433000000, 433075000, 25000)
    o.scanner_start()
    ref, _, allowed = o.accept_stream(u8)
    dump = tmp_path / "dump.s8"
    r = subprocess.run([tool, "5", "-45", "scan=433000000:433075000:25000", "dump=%s" % dump], input=u8.tobytes(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    assert np.array_equal(np.frombuffer(r.stdout, dtype=np.int16), ref)
    assert ("scanner: %d Hz after %d tuning commands" % o.scanner_tuned()) in r.stderr.decode()
    assert o.scanner_tuned()[1] == 1 + int((allowed == 0).sum()) > 4
    want = np.concatenate([oracle.rotate((u8[k:k + 32768] ^ 0x80).view(np.int8), +1) for k in range(0, len(u8), 32768)])
    assert np.array_equal(np.fromfile(dump, dtype=np.int8), want)


@pytest.mark.parametrize("rotation", [1, 0, -1])
def test_front_end_only(capi, oracle, golden, rotation):
    """iqd_front_end: u8 -> s8 -> rotation, interleaved; includes the -128 negation cases (rails, white)."""
    for name in ("rails", "white", "fm_tone"):
        u8 = golden[name]["iq"][:65536]
        eng = capi.Engine(2)
        eng.set_rotation(rotation, first=1, n=1)
        out = eng.front_end(np.stack([u8, u8]))
        s8 = (u8 ^ 0x80).view(np.int8)
        assert np.array_equal(out[0], np.concatenate([oracle.rotate(s8[k:k + 32768], +1) for k in (0, 32768)]))
        assert np.array_equal(out[1], np.concatenate([oracle.rotate(s8[k:k + 32768], rotation) for k in (0, 32768)]))
    if rotation == 1:
        assert np.array_equal(oracle.rotate(golden["primitives"]["rot_in"], +1), golden["primitives"]["rot_up"])


@pytest.mark.parametrize("mode", ["am", "lsb"])
def test_long_row_dc_removal_on_a_noiseless_carrier(capi, oracle, mode):
    """An unmodulated, noiseless carrier for the whole row: the detector output is constant, the true DC-removal
    state sticks at a denormal, a zero-state warm-up at 0.  The hand-offs agree (both below 2^-100), nothing is
    redone, and the PCM is the oracle's."""
    n = 1 << 21
    pat = np.array([[50, 0], [0, -50], [-50, 0], [0, 50]], np.int16)
    u8 = (128 + np.tile(pat, (n // 4, 1))).astype(np.uint8).reshape(-1)
    eng = capi.Engine(1)
    eng.set_mode(mode)
    pcm, cnt, _, _ = eng.accept(u8)
    o = oracle.chain()
    o.set_mode(mode)
    ref, _, _ = o.accept_stream(u8)
    assert np.array_equal(pcm[0, :cnt[0]], ref)
    assert eng.stats()["state_repairs"] == 0


@pytest.mark.parametrize("mode", ["fm", "wbfm", "am", "usb"])
@pytest.mark.parametrize("block_bytes", [32768, 1024])
def test_demodulator_gain_changes_between_calls(capi, oracle, mode, block_bytes):
    """setDemodulatorGain between accept calls: the histories a tile rebuilds from the raw tail lie before the change
    and must be computed with the gain that was in force then (FM / WBFM: the post-discriminator decimators and the
    de-emphasis see K; AM / SSB apply the gain behind every filter).  Short blocks put the change point inside the
    lead-in, long ones at its end."""
    which = {"am": 1, "fm": 2, "wbfm": 3, "usb": 4}[mode]
    base = {1: 300.0, 2: 10185.9, 3: 40743.7, 4: 300.0}[which]
    u8 = synth.fm_tone(24 * 16384, seed=45)
    calls = [(0, 4, 1.0), (4, 5, 0.25), (5, 9, 0.25), (9, 10, 3.0), (10, 18, 0.02), (18, 24, 1.0)]
    if block_bytes == 1024:     # the same stream in many short calls; a change every 2560+ samples
        u8 = u8[:2 * 40960]
        calls = [(k, k + 5, g) for k, g in zip(range(0, 80, 5), [1.0, 0.3, 0.3, 2.0, 0.05, 1.0, 1.0, 4.0] * 2)]
    eng = capi.Engine(2, block_bytes=block_bytes)       # channel 1 keeps its gain: must not be disturbed
    eng.set_mode(mode)
    o, o1 = oracle.chain(), oracle.chain()
    o.set_mode(mode)
    o1.set_mode(mode)
    got, ref, ref1, got1 = [], [], [], []
    for a, b, g in calls:
        eng.set_gain(which, base * g, first=0, n=1)
        o.set_gain(which, base * g)
        part = u8[a * block_bytes:b * block_bytes]
        pcm, cnt, _, _ = eng.accept(np.stack([part, part]))
        got.append(pcm[0, :cnt[0]])
        got1.append(pcm[1, :cnt[1]])
        ref.append(o.accept_stream(part, block_bytes)[0])
        ref1.append(o1.accept_stream(part, block_bytes)[0])
    assert np.array_equal(np.concatenate(got), np.concatenate(ref))
    assert np.array_equal(np.concatenate(got1), np.concatenate(ref1))


@pytest.mark.parametrize("mode", ["am", "fm", "wbfm", "usb"])
def test_no_magnitude_flag_device_path(capi, oracle, mode):
    """IQD_F_NO_MAGNITUDE with only the PCM pointer given: the chain kernels' variants without the squelch magnitude
    (the reference computes it only to feed callbacks nobody registered) - PCM unchanged, through the device entry."""
    n = 5 * 16384
    u8 = synth.fm_tone(n, seed=91)
    eng = capi.Engine(2, flags=1)
    eng.set_mode(mode)
    iq_dev, pcm_dev = eng.dev_alloc(2 * 2 * n), eng.dev_alloc(2 * 2 * (n // 32))
    eng.dev_upload(iq_dev, np.stack([u8, u8[::-1].copy()]))
    eng.accept_device(iq_dev, 2 * n, pcm_dev)
    eng.synchronize()
    got = eng.dev_download(pcm_dev, 2 * 2 * (n // 32), np.int16).reshape(2, -1)
    for c, sig in enumerate([u8, u8[::-1].copy()]):
        o = oracle.chain()
        o.set_mode(mode)
        assert np.array_equal(got[c], o.accept_stream(sig)[0]), c
    eng.dev_free(iq_dev)
    eng.dev_free(pcm_dev)


@pytest.mark.parametrize("mode", ["am", "fm", "wbfm", "lsb"])
def test_rotation_selector_changes_between_calls(capi, oracle, mode):
    """The selector changed in mid-stream: the histories keep what the old rotation produced (the engine rewrites its
    raw tails through the inverse of the new rotation once, so that every tile reads the old-rotated history back)."""
    u8 = synth.fm_tone(10 * 16384, seed=46)
    u8[:64] = 0                                   # some -128 samples: the negation's fixed point
    eng = capi.Engine(1)
    eng.set_mode(mode)
    o = oracle.chain()
    o.set_mode(mode)
    got, ref = [], []
    for k, rot in enumerate([1, -1, -1, 0, 1, 0, -1, 1, 1, 0]):
        eng.set_rotation(rot)
        o.set_rotation(rot)
        part = u8[k * 32768:(k + 1) * 32768]
        pcm, cnt, _, _ = eng.accept(part)
        got.append(pcm[0, :cnt[0]])
        ref.append(o.accept_stream(part)[0])
    assert np.array_equal(np.concatenate(got), np.concatenate(ref))
