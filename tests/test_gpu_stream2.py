"""GPU (MI355X): the FM / AM / SSB streaming pipelines (iqd_stream2.hip) pinned on by IQD_F_WBFM_STREAM, against the
oracle and the golden vectors, sample for sample - the cases the tile kernels are held to in test_gpu_modes.py."""
import numpy as np
import pytest

from rtlsdrdiags_amd import synth

pytestmark = pytest.mark.gpu

STREAM = 0x4
MODES = ["am", "fm", "lsb", "usb"]


@pytest.fixture(scope="module")
def capi():
    from rtlsdrdiags_amd import capi as c
    return c


def oracle_run(oracle, mode, u8, rotation=1, gain=None, block_bytes=32768):
    c = oracle.chain()
    c.set_mode(mode)
    c.set_rotation(rotation)
    if gain is not None:
        c.set_gain({"am": 1, "fm": 2, "lsb": 4, "usb": 4}[mode], gain)
    return c.accept_stream(u8, block_bytes)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["fm_tone", "am_tone", "white", "rails", "capture_excerpt"])
def test_golden(capi, golden, mode, name):
    g = golden[name]
    if g["iq"].size < 2 * 16384 * 2:
        pytest.skip("row shorter than two blocks: AM/SSB take the batched DC path, not the stream kernel")
    eng = capi.Engine(1, flags=STREAM)
    eng.set_mode(mode)
    pcm, cnt, mag, allowed = eng.accept(g["iq"])
    assert eng.stats()["stream_launches"] == 1
    assert np.array_equal(pcm[0, :cnt[0]], g["pcm_" + mode])
    assert np.array_equal(mag[0], g["magnitude"])


@pytest.mark.parametrize("mode", MODES + ["wbfm"])
def test_golden_capture_with_its_dropout_gated_on_the_streaming_kernels(capi, golden, mode):
    """The reference's own recording (4 blocks of demodulatorResearch/yoyo.iq, tests/golden/make_golden.py: capture) in
    4096-byte blocks with the squelch at -33 dBFS: the carrier's dropout closes it for three blocks in mid-stream; the
    gated streaming pipelines must give the reference's PCM, in one call and in two."""
    g = golden["capture_gated"]
    bb = int(g["block_bytes"])
    for cuts in ([0, len(g["iq"])], [0, 20 * bb, len(g["iq"])]):
        eng = capi.Engine(1, block_bytes=bb, flags=STREAM)
        eng.set_mode(mode)
        eng.set_squelch(int(g["threshold"]))
        eng.set_rx_gain_db(int(g["rx_gain_db"]))
        out, flags, mags = [], [], []
        for a, b in zip(cuts[:-1], cuts[1:]):
            pcm, cnt, mag, allowed = eng.accept(g["iq"][a:b])
            out.append(pcm[0, :cnt[0]])
            flags.append(allowed[0])
            mags.append(mag[0])
        assert eng.stats()["stream_launches"] == len(cuts) - 1
        assert np.array_equal(np.concatenate(flags), g["allowed"])
        assert np.array_equal(np.concatenate(mags), g["magnitude"])
        assert np.array_equal(np.concatenate(out), g["pcm_" + mode])
        eng.close()


@pytest.mark.parametrize("mode", MODES)
def test_mixed_rotations_white_bytes_and_calls(capi, oracle, mode):
    """9 channels whose rotation selector cycles +1, 0, -1 (groups padded to 16 segments inside one launch), uniform
    random bytes with runs of 0x00 (-128), three calls."""
    n_ch = 9
    rng = np.random.default_rng(11)
    u8 = rng.integers(0, 256, size=(n_ch, 6 * 32768), dtype=np.uint8)
    u8[:, 200:328] = 0
    eng = capi.Engine(n_ch, flags=STREAM)
    eng.set_mode(mode)
    for c in range(n_ch):
        eng.set_rotation((1, 0, -1)[c % 3], first=c, n=1)
    outs, mags = [], []
    for k in range(3):
        pcm, cnt, mag, _ = eng.accept(u8[:, k * 65536:(k + 1) * 65536])
        outs.append(pcm)
        mags.append(mag)
    assert eng.stats()["stream_launches"] == 3
    got, gmag = np.concatenate(outs, axis=1), np.concatenate(mags, axis=1)
    for c in range(n_ch):
        ref, ref_mag, _ = oracle_run(oracle, mode, u8[c], rotation=(1, 0, -1)[c % 3])
        assert np.array_equal(got[c], ref), (mode, c)
        assert np.array_equal(gmag[c], ref_mag), (mode, c)


@pytest.mark.parametrize("mode", MODES)
def test_many_channels_and_long_rows(capi, oracle, mode):
    n_ch = 70
    make = synth.am_tone if mode == "am" else synth.fm_tone
    u8 = np.stack([make(4 * 16384, seed=500 + c) for c in range(n_ch)])
    eng = capi.Engine(n_ch, flags=STREAM)
    eng.set_mode(mode)
    pcm, cnt, mag, _ = eng.accept(u8)
    for c in range(0, n_ch, 3):
        ref, ref_mag, _ = oracle_run(oracle, mode, u8[c])
        assert np.array_equal(pcm[c, :cnt[c]], ref), (mode, c)
        assert np.array_equal(mag[c], ref_mag), (mode, c)
    long_u8 = make(1 << 21, seed=9)
    eng1 = capi.Engine(1, flags=STREAM)
    eng1.set_mode(mode)
    pcm, cnt, _, _ = eng1.accept(long_u8)
    assert eng1.stats()["stream_launches"] == 1
    assert np.array_equal(pcm[0, :cnt[0]], oracle_run(oracle, mode, long_u8)[0]), mode


def test_fm_gain_changes_and_loud_path(capi, oracle):
    u8 = synth.fm_tone(12 * 16384, seed=21, deviation=70e3)
    c = oracle.chain()
    c.set_mode("fm")
    eng = capi.Engine(1, flags=STREAM)
    eng.set_mode("fm")
    ref, out = [], []
    for k, gain in enumerate([None, 4.0e5, 900.0, 3.0e6, None, 10185.9]):   # 3e6: |y2| far above the clamp-free bound
        if gain is not None:
            c.set_gain(2, gain)
            eng.set_gain("fm", gain)
        piece = u8[k * 65536:(k + 1) * 65536]
        ref.append(c.accept_stream(piece)[0])
        pcm, cnt, _, _ = eng.accept(piece)
        out.append(pcm[0, :cnt[0]])
    assert eng.stats()["stream_launches"] == 6
    for k in range(6):
        assert np.array_equal(out[k], ref[k]), k


def test_alternating_with_the_tile_kernels(capi, oracle):
    """Both kernels read the same carried state (raw tails, DC-removal state): the path may change from call to call."""
    u8 = synth.am_tone(8 * 16384, seed=77)
    for mode in MODES:
        ref = oracle_run(oracle, mode, u8)[0]
        a, b = capi.Engine(1, flags=STREAM), capi.Engine(1, flags=0x2)
        for e in (a, b):
            e.set_mode(mode)
        outs = [[], []]
        for off in range(0, len(u8), 4 * 32768):
            for k, e in enumerate((a, b)):
                pcm, cnt, _, _ = e.accept(u8[off:off + 4 * 32768])
                outs[k].append(pcm[0, :cnt[0]])
        assert np.array_equal(np.concatenate(outs[0]), ref) and np.array_equal(np.concatenate(outs[1]), ref), mode


@pytest.mark.parametrize("mode", MODES)
def test_rotation_selector_changed_between_calls(capi, oracle, mode):
    """The selector of some channels changes between two calls (found by tools/gpu_fuzz.py with the streaming path pinned:
    the families' channel lists are grouped by selector, and the grouping has to follow the change).  The histories keep
    the old rotation exactly, the new samples get the new one."""
    n_ch = 5
    rng = np.random.default_rng(17)
    u8 = rng.integers(0, 256, size=(n_ch, 4 * 32768), dtype=np.uint8)
    first = [1, 0, -1, 1, 1]
    second = [-1, 0, 1, 0, 1]
    eng = capi.Engine(n_ch, flags=STREAM)
    eng.set_mode(mode)
    chains = []
    for c in range(n_ch):
        eng.set_rotation(first[c], first=c, n=1)
        o = oracle.chain()
        o.set_mode(mode)
        o.set_rotation(first[c])
        chains.append(o)
    half = 2 * 32768
    pcm1, _, mag1, _ = eng.accept(u8[:, :half])
    for c in range(n_ch):
        eng.set_rotation(second[c], first=c, n=1)
    pcm2, _, mag2, _ = eng.accept(u8[:, half:])
    assert eng.stats()["stream_launches"] == 2
    for c in range(n_ch):
        r1 = chains[c].accept_stream(u8[c, :half], 32768)
        chains[c].set_rotation(second[c])
        r2 = chains[c].accept_stream(u8[c, half:], 32768)
        assert np.array_equal(pcm1[c], r1[0]) and np.array_equal(mag1[c], r1[1]), (mode, c)
        assert np.array_equal(pcm2[c], r2[0]), (mode, c)
        assert np.array_equal(mag2[c], r2[1]), (mode, c)


@pytest.mark.parametrize("mode", ["fm", "wbfm"])
def test_more_segments_than_one_round_holds(capi, oracle, mode):
    """60 000 channels of 2048 samples: one segment per channel, more than the 49 152 a round of the persistent workgroups
    takes - the second round (ring counters running on, histories rebuilt) has to be as exact as the first."""
    n_ch, n = 60000, 2048
    rng = np.random.default_rng(5)
    base = [synth.fm_tone(n, seed=900 + k) for k in range(7)]
    pick = rng.integers(0, 7, n_ch)
    u8 = np.stack([base[k] for k in pick])
    u8[:, 100:116] = rng.integers(0, 256, size=(n_ch, 16), dtype=np.uint8)     # every channel its own few bytes
    eng = capi.Engine(n_ch, flags=STREAM)
    eng.set_mode(mode)
    # (device pointers: a host-pointer call of this size would be cut into slices of a few thousand channels)
    iq_d, pcm_d = eng.dev_alloc(u8.nbytes), eng.dev_alloc(n_ch * (n // 32) * 2)
    cnt_d, mag_d = eng.dev_alloc(n_ch * 4), eng.dev_alloc(n_ch * 4)
    eng.dev_upload(iq_d, u8)
    eng.accept_device(iq_d, 2 * n, pcm_d, cnt_d, mag_d)
    eng.synchronize()
    assert eng.stats()["stream_launches"] == 1
    pcm = eng.dev_download(pcm_d, n_ch * (n // 32) * 2, np.int16).reshape(n_ch, -1)
    cnt = eng.dev_download(cnt_d, n_ch * 4, np.uint32)
    mag = eng.dev_download(mag_d, n_ch * 4, np.uint32).reshape(n_ch, 1)
    for p_ in (iq_d, pcm_d, cnt_d, mag_d):
        eng.dev_free(p_)
    for c in list(range(0, 40)) + list(range(49100, 49200)) + list(range(n_ch - 40, n_ch)):
        o = oracle.chain()
        o.set_mode(mode)
        ref = o.accept_stream(u8[c], 4096)
        assert np.array_equal(pcm[c, :cnt[c]], ref[0]), (mode, c)
        assert np.array_equal(mag[c], ref[1]), (mode, c)


@pytest.mark.parametrize("mode", ["wbfm", "fm", "usb", "am"])
def test_squelch_gated_calls_stay_on_the_streaming_kernels(capi, oracle, mode):
    """A threshold that could close the squelch makes the call a gated one (magnitude pass, decisions, open-block lists
    first).  Whether the decisions reject nothing or several blocks, the chain runs through the streaming kernel - over
    the concatenation of the channel's open blocks, like the reference's filters (IqDataProcessor.cc:793) - and the
    call stays asynchronous (round 2 read a word back and fell to the tile kernels when anything was rejected)."""
    u8 = synth.fm_tone(8 * 16384, seed=33)
    quiet = u8.copy().reshape(-1, 2)
    quiet[3 * 16384:5 * 16384] = 128                         # two silent blocks: rejected at -40 dBFS
    for data, threshold, all_open in ((u8, -60, True), (quiet.reshape(-1), -40, False)):
        eng = capi.Engine(1, flags=STREAM)
        eng.set_mode(mode)
        eng.set_squelch(threshold)
        o = oracle.chain()
        o.set_mode(mode)
        o.set_squelch(threshold)
        outs, refs = [], []
        for call in range(2):                                 # the second call continues from the gated call's state
            pcm, cnt, mag, allowed = eng.accept(data)
            ref = o.accept_stream(data, 32768)
            assert np.array_equal(allowed[0], ref[2]) and np.array_equal(mag[0], ref[1]), (mode, threshold, call)
            assert np.array_equal(pcm[0, :cnt[0]], ref[0]), (mode, threshold, call)
            assert cnt[0] == len(ref[0]) and not pcm[0, cnt[0]:].any()
        assert eng.stats()["stream_launches"] == 2, (mode, threshold)
        assert bool(np.all(ref[2] == 1)) == all_open          # the case is what it claims to be


def test_gated_streaming_many_channels_like_configs4(capi, oracle):
    """BASELINE configs[4] as bench.py --config 4 stages it, at 2048 channels: LSB / USB alternating, the rotation selector
    cycling, Harris AGC running, loud / quiet blocks per channel class, threshold -60 dBFS: a quarter of the blocks is
    rejected in the steady state and the streaming kernel compacts them away.  Three consecutive calls (the AGC moves
    the gain from call to call), every checked channel against the oracle: PCM, magnitudes, decisions, final IF gain."""
    import bench
    n_ch, n = 2048, 1 << 16
    rows = bench.gating_rows(synth, n)
    iq = np.stack([rows[c % 4] for c in range(n_ch)])
    iq[:, 5000:5016] = np.random.default_rng(3).integers(0, 256, size=(n_ch, 16), dtype=np.uint8)   # every channel a few bytes of its own
    eng = capi.Engine(n_ch)
    bench.configure(eng, "ssb_stress", n_ch, 0, None)
    iq_d, pcm_d = eng.dev_alloc(iq.nbytes), eng.dev_alloc(n_ch * (n // 32) * 2)
    cnt_d, mag_d, al_d = eng.dev_alloc(n_ch * 4), eng.dev_alloc(n_ch * 4 * 4), eng.dev_alloc(n_ch * 4)
    eng.dev_upload(iq_d, iq)
    picks = list(range(0, 24)) + list(range(1000, 1012)) + list(range(n_ch - 12, n_ch))
    chains = {}
    for c in picks:
        o = oracle.chain()
        o.set_mode("lsb" if c % 2 == 0 else "usb")
        o.set_rotation((1, 0, -1)[c % 3])
        o.set_squelch(bench.GATE_THRESHOLD_DBFS)
        o.agc_set_type(1)
        o.agc_enable(True)
        chains[c] = o
    rejected = 0
    for call in range(3):
        before = eng.stats()["stream_launches"]
        eng.accept_device(iq_d, 2 * n, pcm_d, cnt_d, mag_d, al_d)
        eng.synchronize()
        assert eng.stats()["stream_launches"] - before == 1, call
        pcm = eng.dev_download(pcm_d, n_ch * (n // 32) * 2, np.int16).reshape(n_ch, -1)
        cnt = eng.dev_download(cnt_d, n_ch * 4, np.uint32)
        mag = eng.dev_download(mag_d, n_ch * 16, np.uint32).reshape(n_ch, 4)
        al = eng.dev_download(al_d, n_ch * 4, np.uint8).reshape(n_ch, 4)
        rejected += int((al == 0).sum())
        for c in picks:
            ref, rmag, rallowed = chains[c].accept_stream(iq[c])
            assert np.array_equal(al[c], rallowed) and np.array_equal(mag[c], rmag), (call, c)
            assert cnt[c] == len(ref) and np.array_equal(pcm[c, :cnt[c]], ref), (call, c)
            assert eng.rx_gain_db(c) == chains[c].rx_gain_db(), (call, c)
    assert rejected > n_ch                                    # the squelch really closed on many blocks
    for p_ in (iq_d, pcm_d, cnt_d, mag_d, al_d):
        eng.dev_free(p_)


@pytest.mark.parametrize("mode", ["ssb_stress", "wbfm", "mixed"])
def test_prepass_one_call_ahead_gives_the_same_as_inline(capi, mode):
    """IQD_F_PREPASS_OVERLAP: the squelch pre-pass of call N + 1 on its own stream while call N's pipelines run.  Five
    calls queued back to back WITHOUT a synchronisation in between (each call its own input and output buffers, different
    data per call, an ungated call in the middle), then everything compared with an engine that runs the pre-pass inline:
    PCM, counts, magnitudes, decisions, and the AGC's final gains."""
    import bench
    n_ch, n, calls = 1024, 1 << 16, 5
    rng = np.random.default_rng(12)
    rows = bench.gating_rows(synth, n)
    inputs = []
    for k in range(calls):
        iq = np.stack([rows[(c + k) % 4] for c in range(n_ch)])
        iq[:, 7000:7016] = rng.integers(0, 256, size=(n_ch, 16), dtype=np.uint8)
        inputs.append(iq)
    results = []
    for flags in (0, 8):
        eng = capi.Engine(n_ch, flags=flags)
        if mode == "ssb_stress":
            bench.configure(eng, "ssb_stress", n_ch, 0, None)
        else:
            bench.configure(eng, mode, n_ch, 0, -50)
            eng.agc_enable(True, first=0, n=n_ch // 2)
        bufs = []
        for k in range(calls):
            d = dict(iq=eng.dev_alloc(inputs[k].nbytes), pcm=eng.dev_alloc(n_ch * (n // 32) * 2), cnt=eng.dev_alloc(n_ch * 4),
                     mag=eng.dev_alloc(n_ch * 16), al=eng.dev_alloc(n_ch * 4))
            eng.dev_upload(d["iq"], inputs[k])
            bufs.append(d)
        eng.synchronize()
        for k in range(calls):
            if k == 2:
                eng.set_squelch(-200)          # an ungated call between gated ones (its squelch pass runs on the main stream)
            if k == 3:
                eng.set_squelch(-60 if mode == "ssb_stress" else -50)
            eng.accept_device(bufs[k]["iq"], 2 * n, bufs[k]["pcm"], bufs[k]["cnt"], bufs[k]["mag"], bufs[k]["al"])
        eng.synchronize()
        got = []
        for d in bufs:
            got.append((eng.dev_download(d["pcm"], n_ch * (n // 32) * 2, np.int16), eng.dev_download(d["cnt"], n_ch * 4, np.uint32),
                        eng.dev_download(d["mag"], n_ch * 16, np.uint32), eng.dev_download(d["al"], n_ch * 4, np.uint8)))
            for p_ in d.values():
                eng.dev_free(p_)
        gains = [eng.rx_gain_db(c) for c in range(0, n_ch, 37)]
        results.append((got, gains))
        eng.close()
    for k in range(calls):
        for a, b in zip(results[0][0][k], results[1][0][k]):
            assert np.array_equal(a, b), (mode, k)
    assert results[0][1] == results[1][1]
    assert (results[0][0][0][3] == 0).any() and (results[0][0][2][3] == 1).all()   # calls 0 gated, call 2 all open
