"""GPU (MI355X): the BASELINE configurations at (reduced) scale — thousands of channels per launch — with
a sample of channels checked bit for bit against the oracle, and the WBFM hand-off repair path."""
import numpy as np
import pytest

from rtlsdrdiags_amd import synth

pytestmark = pytest.mark.gpu
ORDER = ["am", "fm", "wbfm", "lsb", "usb"]
DEMOD_ID = {"am": 1, "fm": 2, "wbfm": 3, "lsb": 4, "usb": 4}


@pytest.fixture(scope="module")
def capi():
    from rtlsdrdiags_amd import capi as c
    return c


def channel_signal(c, n):
    """A different, cheap-to-make signal per channel: one seeded FM tone, byte-rotated and offset."""
    base = synth.fm_tone(n, seed=500 + (c % 7), deviation=2500.0 + 700.0 * (c % 11), amplitude=30.0 + (c % 50))
    return np.roll(base, 2 * (c % 1013))


def test_config3_4096_fm_channels(capi, oracle):
    """BASELINE configs[2]: 4096 concurrent FM channels, two 64 ms blocks each, two calls."""
    n_ch, n = 4096, 2 * 16384
    iq = np.stack([channel_signal(c, n) for c in range(n_ch)])
    eng = capi.Engine(n_ch)
    eng.set_mode("fm")
    p1, c1, m1, _ = eng.accept(iq[:, :32768])
    p2, c2, m2, _ = eng.accept(iq[:, 32768:])
    assert (c1 == 512).all() and (c2 == 512).all()
    for c in list(range(0, n_ch, 257)) + [n_ch - 1]:
        o = oracle.chain()
        o.set_mode("fm")
        ref, mag, _ = o.accept_stream(iq[c])
        assert np.array_equal(np.concatenate([p1[c], p2[c]]), ref), c
        assert m1[c, 0] == mag[0] and m2[c, 0] == mag[1]


def test_config4_mixed_modes_2048_channels(capi, oracle):
    """BASELINE configs[3] in one GPU's share: ch % 5 -> {AM, FM, WBFM, LSB, USB}."""
    n_ch, n = 2048, 16384
    iq = np.stack([channel_signal(c, n) for c in range(n_ch)])
    eng = capi.Engine(n_ch)
    for c in range(n_ch):
        eng.set_mode(ORDER[c % 5], first=c, n=1)
    pcm, cnt, mag, allowed = eng.accept(iq)
    assert (cnt == 512).all() and allowed.all()
    for c in range(0, n_ch, 97):
        o = oracle.chain()
        o.set_mode(ORDER[c % 5])
        ref, rmag, _ = o.accept_stream(iq[c])
        assert np.array_equal(pcm[c], ref), (c, ORDER[c % 5])
        assert mag[c, 0] == rmag[0]


def test_config5_ssb_rotation_squelch(capi, oracle):
    """BASELINE configs[4] in miniature: LSB/USB alternating, per-channel rotation selector
    (none / +Fs/4 / -Fs/4), squelch threshold raised so that blocks gate, 1024 channels, 4096-byte blocks."""
    n_ch, bb, nblk = 1024, 4096, 8
    amps = [2, 60, 60, 2, 2, 2, 90, 2]
    base = synth.stepped_amplitude(amps, block_samples=bb // 2, seed=9)
    iq = np.stack([np.roll(base.reshape(nblk, bb), c % nblk, axis=0).reshape(-1) for c in range(n_ch)])
    eng = capi.Engine(n_ch, block_bytes=bb)
    eng.set_squelch(-40)
    for c in range(n_ch):
        eng.set_mode("lsb" if c % 2 == 0 else "usb", first=c, n=1)
        eng.set_rotation((0, 1, -1)[c % 3], first=c, n=1)
    pcm, cnt, mag, allowed = eng.accept(iq)
    for c in range(0, n_ch, 41):
        o = oracle.chain()
        o.set_mode("lsb" if c % 2 == 0 else "usb")
        o.set_rotation((0, 1, -1)[c % 3])
        o.set_squelch(-40)
        ref, rmag, rallowed = o.accept_stream(iq[c], bb)
        assert np.array_equal(allowed[c], rallowed), c
        assert np.array_equal(mag[c], rmag), c
        assert cnt[c] == len(ref) and np.array_equal(pcm[c, :cnt[c]], ref), c
    assert 0 < allowed.sum() < allowed.size


def test_wbfm_handoff_repair_on_periodic_input(capi, oracle):
    """Strictly periodic input can keep two de-emphasis trajectories one ulp apart for ever, so the
    zero-state lead-in of a tile never lands on the exact state: the verification must notice and the
    repair path must restore bit-exactness."""
    pattern = np.tile(np.array([255, 255, 255, 255, 0, 0, 0, 0], dtype=np.uint8), 1 << 17)   # 2^19 samples
    o = oracle.chain()
    o.set_mode("wbfm")
    ref, _, _ = o.accept_stream(pattern)
    eng = capi.Engine(1)
    eng.set_mode("wbfm")
    pcm, cnt, _, _ = eng.accept(pattern)
    assert np.array_equal(pcm[0, :cnt[0]], ref)
    st = eng.stats()
    assert st["state_repairs"] >= 1 and st["state_checks"] >= 1      # repaired on the device, then chained up


def _carrier(n, amp=60):
    """A noiseless carrier at -Fs/4: exactly constant after the front-end rotation."""
    pat = np.array([[amp, 0], [0, -amp], [-amp, 0], [0, amp]], np.int16)
    return (128 + np.tile(pat, (n // 4, 1))).astype(np.uint8).reshape(-1)


@pytest.mark.parametrize("kind", ["silence", "carrier", "bursts"])
def test_wbfm_digital_silence_and_noiseless_carrier(capi, oracle, kind):
    """With an exactly constant input the de-emphasis state does not reach zero: it sticks at a denormal
    (0.949 k rounds back to k for |k| <= 9 units of 2^-149), while a segment or tile warming up from zero sits at 0.
    Insisting on bit-equal hand-offs would then serialise the whole row (seconds for a long one) although both
    states are below anything that can reach a sample; the checks accept two sub-2^-100 states as agreeing.
    The PCM must still be the oracle's, also across calls and once a live signal returns."""
    n = 1 << 22
    if kind == "silence":
        u8 = np.concatenate([synth.fm_tone(1 << 16, seed=5), np.full(2 * (n - (1 << 16)), 128, np.uint8)])
    elif kind == "carrier":
        u8 = _carrier(n)
    else:   # modulated stretches between long noiseless ones: the trajectories merge at every burst
        parts = [_carrier(700000), synth.fm_tone(90000, seed=6), _carrier(1500000), synth.fm_tone(4096, seed=7),
                 _carrier(n - 700000 - 90000 - 1500000 - 4096)]
        u8 = np.concatenate(parts)
    o = oracle.chain()
    o.set_mode("wbfm")
    ref, _, _ = o.accept_stream(u8)
    eng = capi.Engine(1)
    eng.set_mode("wbfm")
    import time
    t0 = time.perf_counter()
    pcm, cnt, _, _ = eng.accept(u8[:2 * (3 << 20)])
    pcm2, cnt2, _, _ = eng.accept(u8[2 * (3 << 20):])        # the carried state is a stuck denormal too
    dt = time.perf_counter() - t0
    assert np.array_equal(np.concatenate([pcm[0, :cnt[0]], pcm2[0, :cnt2[0]]]), ref)
    assert eng.stats()["state_repairs"] == 0   # no tile was re-run
    assert dt < 1.0, dt                        # (serial re-runs of every tile would take seconds)
    # a live signal afterwards: the states merge at once and the stream continues exactly
    more = synth.fm_tone(1 << 18, seed=8)
    p3, c3, _, _ = eng.accept(more)
    r3, _, _ = o.accept_stream(more)
    assert np.array_equal(p3[0, :c3[0]], r3)


def test_config5_65536_ssb_channels_with_rotation_squelch_and_agc(capi, oracle):
    """BASELINE configs[4] at full channel count in one accept call: 65536 SSB channels (even LSB, odd USB), the
    rotation selector per channel, the squelch raised so that blocks gate, a Harris AGC per channel moving the
    gain the next block's squelch sees.  A sample of channels against the oracle: PCM, decisions, final gains."""
    n_ch, bb, nblk = 65536, 2048, 6
    rng = np.random.default_rng(55)
    base = [synth.stepped_amplitude([int(a) for a in rng.choice([2, 40, 90], nblk)], block_samples=bb // 2, seed=70 + k)
            for k in range(16)]
    iq = np.empty((n_ch, nblk * bb), np.uint8)
    for k in range(16):
        iq[k::16] = base[k]
    eng = capi.Engine(n_ch, block_bytes=bb)
    eng.set_squelch(-46)
    eng.set_mode("lsb")
    for c0 in range(1, 4096, 2):      # odd channels USB, in strided runs: set the first 4096 one by one, then copy the pattern
        eng.set_mode("usb", first=c0, n=1)
    # mode and rotation by residue classes need per-channel calls; keep that to a prefix and a suffix of the batch
    checked = list(range(0, 4096, 131)) + list(range(n_ch - 4096, n_ch, 257))
    for c in range(n_ch - 4096, n_ch):
        if c % 2:
            eng.set_mode("usb", first=c, n=1)
    for c in list(range(4096)) + list(range(n_ch - 4096, n_ch)):
        eng.set_rotation((0, 1, -1)[c % 3], first=c, n=1)
    eng.agc_set_operating_point(-10)
    eng.agc_enable(True)
    pcm, cnt, mag, allowed = eng.accept(iq)
    for c in checked:
        o = oracle.chain()
        o.set_mode("usb" if c % 2 else "lsb")
        o.set_rotation((0, 1, -1)[c % 3])
        o.set_squelch(-46)
        o.agc_set_operating_point(-10)
        o.agc_enable(True)
        ref, rmag, rallowed = o.accept_stream(iq[c], bb)
        assert np.array_equal(allowed[c], rallowed) and np.array_equal(mag[c], rmag), c
        assert cnt[c] == len(ref) and np.array_equal(pcm[c, :cnt[c]], ref), c
        assert eng.rx_gain_db(c) == o.rx_gain_db(), c
    assert 0 < allowed.sum() < allowed.size


def test_bench_workload_at_full_size_is_the_oracles_pcm(capi, oracle):
    """BASELINE configs[1] exactly as bench.py runs it - WBFM, one channel, 2^28 samples resident in device memory,
    two consecutive steps - with every one of the 2 x 8.4 M PCM samples compared with the oracle's (the oracle
    needs ~25 s of one host core for the two steps), and the linearity-free property the domain offers at this size:
    the second step continues the first (state carried), so it differs from it."""
    n, period = 1 << 28, 1 << 24
    u8 = synth.fm_tone(period, seed=1234)
    eng = capi.Engine(1)
    eng.set_mode("wbfm")
    iq = eng.dev_alloc(2 * n)
    pcm_dev = eng.dev_alloc(2 * (n // 32))
    eng.dev_upload(iq, u8)
    eng.dev_tile(iq, 2 * period, 2 * n)
    o = oracle.chain()
    o.set_mode("wbfm")
    outs = []
    for step in range(2):
        eng.accept_device(iq, 2 * n, pcm_dev)
        eng.synchronize()
        got = eng.dev_download(pcm_dev, 2 * (n // 32), np.int16)
        ref = np.concatenate([o.accept_stream(u8)[0] for _ in range(n // period)])
        assert len(ref) == n // 32 and np.array_equal(got, ref), step
        outs.append(got)
    assert not np.array_equal(outs[0][:4096], outs[1][:4096])     # the start of step 2 carries step 1's state
    assert np.array_equal(outs[0][period // 32:], outs[1][period // 32:])   # once the periodic input has settled
    st = eng.stats()
    assert st["state_checks"] > 1000 and st["state_repairs"] == 0
    eng.dev_free(iq)
    eng.dev_free(pcm_dev)


# ---- BASELINE configs[3] at its per-GPU size, on the path bench.py --config 3 times -------------------------------
def _mixed_rows(n_ch, n, seed):
    """n_ch distinct rows of n samples without n_ch synthesiser runs: 35 seeded FM tones, each row one of them rolled by
    its own offset with a few bytes of its own."""
    rng = np.random.default_rng(seed)
    base = [synth.fm_tone(n, seed=600 + k, deviation=2000.0 + 800.0 * (k % 11), amplitude=25.0 + 2 * k) for k in range(35)]
    u8 = np.empty((n_ch, 2 * n), np.uint8)
    for c in range(n_ch):
        u8[c] = np.roll(base[c % 35], 2 * ((c * 37) % 1009))
    u8[:, 4000:4032] = rng.integers(0, 256, size=(n_ch, 32), dtype=np.uint8)
    return u8


def _mixed_setup(eng, n_ch, wbfm_every=5):
    """channel % 5 -> {AM, FM, WBFM, LSB, USB}; the rotation selector cycles over the channels of the AM / FM / SSB
    families (the WBFM streaming kernel takes one selector per launch).  Returns (mode, rotation) per channel."""
    modes, rots = [], []
    for c in range(n_ch):
        m = ORDER[c % 5]
        if m == "wbfm" and (c // 5) % wbfm_every:
            m = "fm"                                          # (thinned-out WBFM family for the "too small to stream" case)
        r = 1 if m == "wbfm" else (1, 0, -1)[(c // 5) % 3]
        modes.append(m)
        rots.append(r)
    # runs of equal settings go down in one call each
    for arr, setter in ((modes, eng.set_mode), (rots, eng.set_rotation)):
        c0 = 0
        for c in range(1, n_ch + 1):
            if c == n_ch or arr[c] != arr[c0]:
                setter(arr[c0], first=c0, n=c - c0)
                c0 = c
    return modes, rots


def _sample_channels(modes, rots, extra=24, seed=3):
    """first and last channel of every (mode, rotation) class, plus a few at random"""
    first, last = {}, {}
    for c, key in enumerate(zip(modes, rots)):
        first.setdefault(key, c)
        last[key] = c
    rng = np.random.default_rng(seed)
    picks = set(first.values()) | set(last.values()) | set(int(x) for x in rng.integers(0, len(modes), extra))
    return sorted(picks)


def _run_mixed_on_device(capi, oracle, n_ch, log2, expect_streams, wbfm_every=1, calls=2, extra=24):
    n = 1 << log2
    u8 = _mixed_rows(n_ch, n, seed=n_ch)
    eng = capi.Engine(n_ch)
    modes, rots = _mixed_setup(eng, n_ch, wbfm_every)
    iq_d, pcm_d = eng.dev_alloc(u8.nbytes), eng.dev_alloc(n_ch * (n // 32) * 2)
    nblk = 2 * n // 32768
    cnt_d, mag_d, al_d = eng.dev_alloc(n_ch * 4), eng.dev_alloc(n_ch * nblk * 4), eng.dev_alloc(n_ch * nblk)
    eng.dev_upload(iq_d, u8)
    picks = _sample_channels(modes, rots, extra)
    chains = {}
    for c in picks:
        o = oracle.chain()
        o.set_mode(modes[c])
        o.set_rotation(rots[c])
        chains[c] = o
    for call in range(calls):
        before = eng.stats()["stream_launches"]
        eng.accept_device(iq_d, 2 * n, pcm_d, cnt_d, mag_d, al_d)
        eng.synchronize()
        assert eng.stats()["stream_launches"] - before == expect_streams, (call, eng.stats())
        pcm = eng.dev_download(pcm_d, n_ch * (n // 32) * 2, np.int16).reshape(n_ch, -1)
        cnt = eng.dev_download(cnt_d, n_ch * 4, np.uint32)
        mag = eng.dev_download(mag_d, n_ch * nblk * 4, np.uint32).reshape(n_ch, nblk)
        assert (cnt == n // 32).all()
        assert eng.dev_download(al_d, n_ch * nblk, np.uint8).all()
        for c in picks:
            ref, rmag, _ = chains[c].accept_stream(u8[c])
            assert np.array_equal(pcm[c], ref), (call, c, modes[c], rots[c])
            assert np.array_equal(mag[c], rmag), (call, c)
    assert eng.stats()["state_repairs"] == 0
    for p_ in (iq_d, pcm_d, cnt_d, mag_d, al_d):
        eng.dev_free(p_)
    return len(picks)


def test_config4_mixed_4096_channels_on_cu_shares(capi, oracle):
    """BASELINE configs[3] exactly as `bench.py --config 3` runs it on one GPU: 4096 channels x 2^16 samples,
    channel % 5 -> {AM, FM, WBFM, LSB, USB}, every channel its own data, the rotation selector varying inside the
    AM / FM / SSB families.  The four families' streaming pipelines run side by side on planned shares of the CUs, as
    ranges of one launch's workgroups (stream_launches == 4 per call proves the pipelines ran); two consecutive calls; the first and last channel of every family and rotation group and
    two dozen more against the oracle, every PCM sample and magnitude."""
    checked = _run_mixed_on_device(capi, oracle, 4096, 16, expect_streams=4, extra=40)
    assert checked >= 64


@pytest.mark.parametrize("squelch,flags", [(None, 0), (-38, 0), (None, 1)])
def test_one_launch_for_all_families_equals_a_kernel_per_family(capi, squelch, flags):
    """The mixed call runs its families' streaming pipelines as ranges of ONE launch's workgroups (iqd_stream_mixed.hip,
    stats.mixed_launches); IQD_MIXED=forked keeps the earlier arrangement, a kernel per family on side streams.  Same
    input, two engines, three calls with a gain change in between: EVERY PCM sample, magnitude and count of all 4096
    channels identical (the sampled oracle comparison is the test above).  With a squelch threshold the call is gated -
    a third of the rows have a quiet second block that closes behind the one-block tail - and the same launch runs the
    pipelines' gated instantiations on each channel's open blocks."""
    import os
    n_ch, n = 4096, 1 << 16
    u8 = _mixed_rows(n_ch, n, seed=99)
    if squelch is not None:
        quiet = np.arange(n_ch) % 3 == 1
        u8[quiet, 32768:98304] = 128 + ((u8[quiet, 32768:98304].astype(np.int16) - 128) // 32).astype(np.int16)   # blocks 1 and 2 of 4
    outs = []
    for forked in (False, True):
        if forked:
            os.environ["IQD_MIXED"] = "forked"
        try:
            eng = capi.Engine(n_ch, flags=flags)   # (the variable is read once, by iqd_create; flags 1 = IQD_F_NO_MAGNITUDE: the
                                                   #  launch's third variant, pipelines that take no squelch magnitudes)
        finally:
            os.environ.pop("IQD_MIXED", None)
        _mixed_setup(eng, n_ch)
        if squelch is not None:
            eng.set_squelch(squelch)
        iq_d, pcm_d = eng.dev_alloc(u8.nbytes), eng.dev_alloc(n_ch * (n // 32) * 2)
        eng.dev_upload(pcm_d, np.zeros(n_ch * (n // 32), np.int16))     # (rows of a gated call are only partly written)
        nblk = 2 * n // 32768
        cnt_d, mag_d = eng.dev_alloc(n_ch * 4), eng.dev_alloc(n_ch * nblk * 4)
        eng.dev_upload(mag_d, np.zeros(n_ch * nblk, np.uint32))         # (not written when the engine takes no magnitudes)
        eng.dev_upload(iq_d, u8)
        got = []
        for call in range(3):
            if call == 2:
                eng.set_gain("fm", 9000.0, first=1, n=1)          # an FM channel's gain: the next call's lead-ins reach back across it
            eng.accept_device(iq_d, 2 * n, pcm_d, cnt_d, 0 if flags & 1 else mag_d)
            eng.synchronize()
            got.append((eng.dev_download(pcm_d, n_ch * (n // 32) * 2, np.int16), eng.dev_download(cnt_d, n_ch * 4, np.uint32),
                        eng.dev_download(mag_d, n_ch * nblk * 4, np.uint32)))
        st = eng.stats()
        assert st["mixed_launches"] == (0 if forked else 3) and st["stream_launches"] == 12 and st["state_repairs"] == 0, st
        if squelch is not None:
            assert (got[0][1] < n // 32).any() and (got[0][1] > 0).all()   # some rows lost blocks, none lost everything
        outs.append(got)
        for p_ in (iq_d, pcm_d, cnt_d, mag_d):
            eng.dev_free(p_)
        eng.close()
    for call in range(3):
        for a, b in zip(outs[0][call], outs[1][call]):
            assert np.array_equal(a, b), call
    assert not np.array_equal(outs[0][0][0], outs[0][1][0])       # (the calls differ: state is carried)


def test_mixed_1400_channels_at_the_share_threshold(capi, oracle):
    """The smallest mixed call that still plans CU shares (every family brings just enough samples for its share)."""
    _run_mixed_on_device(capi, oracle, 1410, 16, expect_streams=4)


def test_mixed_call_with_a_tiny_family(capi, oracle):
    """A WBFM family of 82 channels beside thousands of others: its share is the minimum of 8 CUs, on which it runs
    segments of the minimum length - still a streaming kernel beside the others."""
    _run_mixed_on_device(capi, oracle, 4096, 16, expect_streams=4, wbfm_every=10)


def test_mixed_call_of_one_block_per_channel(capi, oracle):
    """16384 channels x 2^14 samples - one 64 ms block per channel per call, the reference's own operating point
    (DataConsumer.cc:342).  Rounds 2-3 kept AM and SSB rows of 512 PCM samples on their tile kernels; since round 4 they
    take their streaming pipelines like the other two families (the one-wave DC pass behind them), so the call plans CU
    shares and runs as one launch."""
    _run_mixed_on_device(capi, oracle, 16384, 14, expect_streams=4)


def test_short_am_ssb_rows_stream_and_match_the_tile_kernels(capi, oracle, monkeypatch):
    """AM / LSB / USB rows of 128 .. 512 PCM samples on the streaming pipelines (IQD_AM_STREAM_MIN: 128) against the oracle,
    many channels, rotation selectors mixed, two calls - and the same call with the rule of rounds 2-3 (tile kernels)."""
    n_ch = 3000
    for log2 in (12, 13, 14):
        n = 1 << log2
        u8 = _mixed_rows(n_ch, n, seed=log2)
        modes = [("am", "lsb", "usb")[c % 3] for c in range(n_ch)]
        rots = [(1, 0, -1)[(c // 3) % 3] for c in range(n_ch)]
        outs = []
        for rule in ("128", "513"):
            monkeypatch.setenv("IQD_AM_STREAM_MIN", rule)
            eng = capi.Engine(n_ch, flags=4)               # IQD_F_WBFM_STREAM: every chain streams where its rules allow
            for c in range(n_ch):
                eng.set_mode(modes[c], first=c, n=1)
                eng.set_rotation(rots[c], first=c, n=1)
            got = [eng.accept(u8)[0].copy() for _ in range(2)]
            assert (eng.stats()["stream_launches"] > 0) == (rule == "128"), (log2, rule, eng.stats())
            outs.append(got)
            eng.close()
        for k in range(2):
            assert np.array_equal(outs[0][k], outs[1][k]), (log2, k)
        for c in range(0, n_ch, 211):
            o = oracle.chain()
            o.set_mode(modes[c])
            o.set_rotation(rots[c])
            for k in range(2):
                ref, _, _ = o.accept_stream(u8[c])
                assert np.array_equal(outs[0][k][c], ref), (log2, c, k)


def test_am_streams_beside_ssb_on_the_tile_kernels(capi, oracle, monkeypatch):
    """Rows of one block: 1400 AM channels take their streaming pipeline (detector stream channel-major), 50 to 250 LSB / USB
    channels of the same call stay on the tile kernels (time-major for rows this short).  Round 4's fuzzer found the two
    families sharing one detector-stream buffer in that combination; each has its own now.  Every row of the small family
    and every 97th of the large one against the oracle, two calls.  (IQD_MIXED=forked: a kernel per family, as for every
    call that the one-launch arrangement does not take.)"""
    monkeypatch.setenv("IQD_MIXED", "forked")
    n_ch, n = 1500, 1 << 14                                # (below 64 MiB: one call, not slices)
    u8 = _mixed_rows(n_ch, n, seed=77)
    modes = ["am" if c % 29 else ("lsb", "usb")[(c // 29) % 2] for c in range(n_ch)]
    n_ssb = sum(m != "am" for m in modes)
    assert 40 <= n_ssb <= 60
    # a larger SSB family (still below its streaming threshold) in a second engine
    for extra in (0, 200):
        mm = list(modes)
        for c in range(extra):
            mm[1 + 7 * c] = "usb"
        eng = capi.Engine(n_ch)
        for c in range(n_ch):
            eng.set_mode(mm[c], first=c, n=1)
            eng.set_rotation((1, 0, -1)[c % 3], first=c, n=1)
        got = [eng.accept(u8)[0].copy() for _ in range(2)]
        st = eng.stats()
        assert st["stream_launches"] == 2 and st["mixed_launches"] == 0, st     # AM streamed (twice), SSB did not
        eng.close()
        for c in [c for c in range(n_ch) if mm[c] != "am"] + list(range(0, n_ch, 97)):
            o = oracle.chain()
            o.set_mode(mm[c])
            o.set_rotation((1, 0, -1)[c % 3])
            for k in range(2):
                ref, _, _ = o.accept_stream(u8[c])
                assert np.array_equal(got[k][c], ref), (extra, c, mm[c], k)
