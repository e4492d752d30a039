"""GPU (MI355X): the per-channel AutomaticGainControl of the engine against the oracle and the golden vectors the
unmodified reference produced: operator commands, magnitudes arriving block by block inside one
accept call, the gain moving the next block's squelch decision, thousands of channels with their own AGC."""
import numpy as np
import pytest

import agc_script as A
from rtlsdrdiags_amd import synth

pytestmark = pytest.mark.gpu
BB = 256            # bytes per block in the magnitude scripts: 128 samples


@pytest.fixture(scope="module")
def capi():
    from rtlsdrdiags_amd import capi as c
    return c


def block_with_magnitude(m):
    """128 samples whose SignalDetector average is exactly m (0..191): I = +m, or |I| = 128 with Q = 2(m-128)."""
    blk = np.empty((BB // 2, 2), np.uint8)
    if m <= 127:
        blk[:, 0], blk[:, 1] = 128 + m, 128
    else:
        blk[:, 0], blk[:, 1] = 0, 128 + 2 * (m - 128)
    return blk.reshape(-1)


class EngineChain:
    """Drives channel `ch` of an engine with agc_script.replay: runs of magnitudes become ONE multi-block accept
    call, so the gains after each block come from the in-kernel block loop (read back through the gain trace)."""

    def __init__(self, eng, ch=0):
        self.e, self.ch, self.pending = eng, ch, []
        for name in ("agc_set_type", "agc_set_deadband", "agc_set_blanking_limit", "agc_set_filter_coefficient",
                     "agc_set_operating_point", "agc_enable"):
            setattr(self, name, self._command(getattr(eng, name)))
        eng.set_gain_trace(True)

    def _command(self, fn):
        def call(v):
            self.flush()
            return fn(v, first=self.ch, n=1)
        return call

    def set_rx_gain_db(self, g):
        self.flush()
        self.e.set_rx_gain_db(g, first=self.ch, n=1)

    def agc_feed(self, m):
        self.pending.append(int(m))

    def flush(self):
        if not self.pending:
            return []
        iq = np.concatenate([block_with_magnitude(m) for m in self.pending])
        n = len(self.pending)
        _, _, mag, _ = self.e.accept(iq, first=self.ch, n=1)
        assert mag[0].tolist() == self.pending
        trace = self.e.gain_trace(n, first=self.ch, n=1)[0]
        after = list(trace[1:]) + [self.e.rx_gain_db(self.ch)]
        self.pending = []
        return after

    def rx_gain_db(self):
        return self.e.rx_gain_db(self.ch)


def replay_engine(eng, codes, values, ch=0):
    """agc_script.replay semantics, but with consecutive magnitudes batched into one accept call."""
    chain = EngineChain(eng, ch)
    flags = np.ones(len(codes), np.uint8)
    gains = np.zeros(len(codes), np.uint32)
    run_start = None
    for k, (code, v) in enumerate(zip(codes, values)):
        if int(code) == A.FEED:
            if run_start is None:
                run_start = k
            chain.agc_feed(min(int(v), 191))
            continue
        if run_start is not None:
            gains[run_start:k] = chain.flush()
            run_start = None
        f, g = A.replay(chain, [code], [v])
        flags[k], gains[k] = f[0], g[0]
    if run_start is not None:
        gains[run_start:] = chain.flush()
    return flags, gains


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_command_scripts_match_the_reference(capi, golden, seed):
    g = golden["agc"]
    codes, values = g["script%d_codes" % seed], g["script%d_values" % seed]
    eng = capi.Engine(1, block_bytes=BB)
    flags, gains = replay_engine(eng, codes, values)
    assert np.array_equal(flags, g["script%d_flags" % seed])
    assert np.array_equal(gains, g["script%d_gains" % seed])


@pytest.mark.parametrize("seed", range(200, 206))
def test_random_scripts_match_the_oracle(capi, oracle, seed):
    codes, values = A.random_script(seed, 400)
    values = np.where(codes == A.FEED, np.minimum(values, 191), values).astype(np.float32)
    fo, go = A.replay(oracle.chain(), codes, values)
    eng = capi.Engine(3, block_bytes=BB)
    flags, gains = replay_engine(eng, codes, values, ch=1)     # the neighbours must stay untouched
    assert np.array_equal(flags, fo) and np.array_equal(gains, go)
    for other in (0, 2):
        st = eng.agc_state(other)
        assert st["rx_gain_db"] == 24 and st["filtered_if_gain_db"] == 24.0 and not st["enabled"]


@pytest.mark.parametrize("case", [c[0] for c in A.STREAM_CASES])
@pytest.mark.parametrize("one_call", [True, False])
def test_agc_moves_the_squelch_inside_the_block_flow(capi, golden, case, one_call):
    g = golden["agc"]
    cfg = dict((c[0], c[2]) for c in A.STREAM_CASES)[case]
    iq = g[case + "_iq"]
    eng = capi.Engine(1, block_bytes=4096)
    eng.set_gain_trace(True)

    class Ch:
        set_mode = staticmethod(eng.set_mode)
        set_squelch = staticmethod(eng.set_squelch)
        set_rx_gain_db = staticmethod(eng.set_rx_gain_db)
    ch = Ch()
    for name in ("agc_set_type", "agc_set_deadband", "agc_set_blanking_limit", "agc_set_filter_coefficient",
                 "agc_set_operating_point", "agc_enable"):
        setattr(ch, name, getattr(eng, name))
    A.configure(ch, cfg)
    nblk = len(iq) // 4096
    if one_call:
        pcm, cnt, _, allowed = eng.accept(iq)
        trace = eng.gain_trace(nblk)[0]
        gains = np.array(list(trace[1:]) + [eng.rx_gain_db(0)], np.uint32)
        pcm, allowed = pcm[0, :cnt[0]], allowed[0]
    else:
        parts, allowed, gains = [], [], []
        for b in range(nblk):
            p, c, _, a = eng.accept(iq[b * 4096:(b + 1) * 4096])
            parts.append(p[0, :c[0]])
            allowed.append(a[0, 0])
            gains.append(eng.rx_gain_db(0))
        pcm, allowed, gains = np.concatenate(parts), np.array(allowed, np.uint8), np.array(gains, np.uint32)
    assert np.array_equal(gains, g[case + "_gains"])
    assert np.array_equal(allowed, g[case + "_allowed"])
    assert np.array_equal(pcm, g[case + "_pcm"])


def test_thousands_of_channels_each_with_its_own_agc(capi, oracle):
    """BASELINE configs[4] flavour: SSB channels, squelch raised, per-channel AGC settings; a sample of channels
    against the oracle (PCM, decisions, final AGC state)."""
    n_ch, bb, nblk = 2048, 2048, 24
    rng = np.random.default_rng(77)
    base = [synth.stepped_amplitude([int(a) for a in rng.integers(1, 120, nblk)], block_samples=bb // 2, seed=60 + k)
            for k in range(8)]
    iq = np.stack([base[c % 8] for c in range(n_ch)])
    eng = capi.Engine(n_ch, block_bytes=bb)
    eng.set_squelch(-50)
    cfgs = []
    for c in range(n_ch):
        cfg = dict(mode="lsb" if c % 2 else "usb", type=c % 2, alpha=[0.2, 0.5, 0.8][c % 3], deadband=c % 4,
                   blanking=c % 3, operating_point=-8 - (c % 9), enabled=(c % 5 != 0))
        cfgs.append(cfg)
    for c in range(n_ch):
        cfg = cfgs[c]
        eng.set_mode(cfg["mode"], first=c, n=1)
        eng.agc_set_type(cfg["type"], first=c, n=1)
        eng.agc_set_filter_coefficient(cfg["alpha"], first=c, n=1)
        eng.agc_set_deadband(cfg["deadband"], first=c, n=1)
        eng.agc_set_blanking_limit(cfg["blanking"], first=c, n=1)
        eng.agc_set_operating_point(cfg["operating_point"], first=c, n=1)
        eng.agc_enable(cfg["enabled"], first=c, n=1)
    pcm, cnt, mag, allowed = eng.accept(iq)
    for c in list(range(0, n_ch, 97)) + [5, 10, n_ch - 1]:
        cfg = cfgs[c]
        o = oracle.chain()
        o.set_mode(cfg["mode"])
        o.set_squelch(-50)
        o.agc_set_type(cfg["type"]); o.agc_set_filter_coefficient(cfg["alpha"]); o.agc_set_deadband(cfg["deadband"])
        o.agc_set_blanking_limit(cfg["blanking"]); o.agc_set_operating_point(cfg["operating_point"])
        if cfg["enabled"]:
            o.agc_enable(True)
        ref, rmag, rallowed = o.accept_stream(iq[c], bb)
        assert np.array_equal(allowed[c], rallowed), c
        assert np.array_equal(mag[c], rmag), c
        assert cnt[c] == len(ref) and np.array_equal(pcm[c, :cnt[c]], ref), c
        assert eng.rx_gain_db(c) == o.rx_gain_db(), c
    assert 0 < allowed.sum() < allowed.size


class ScanChain:
    """Engine channel 0 with the chain interface agc_script.scan_scenario drives."""

    def __init__(self, eng, one_call):
        self.e, self.one_call = eng, one_call
        self.set_mode, self.set_squelch = eng.set_mode, eng.set_squelch
        self.scanner_set_parameters = eng.scanner_set_parameters
        self.scanner_start = lambda: eng.scanner_start(True)
        self.scanner_stop = lambda: eng.scanner_start(False)
        self.scanner_tuned = lambda: eng.scanner_tuned(0)

    def feed(self, _chain, iq):
        bb = 4096
        nblk = len(iq) // bb
        if not self.one_call:
            pcm, freq, count = [], [], []
            for b in range(nblk):
                p, c, _, _ = self.e.accept(iq[b * bb:(b + 1) * bb])
                pcm.append(p[0, :c[0]])
                f, n = self.e.scanner_tuned(0)
                freq.append(f)
                count.append(n)
            return np.concatenate(pcm), np.array(freq, np.uint64), np.array(count, np.uint32)
        before = self.e.scanner_tuned(0)
        p, c, _, _ = self.e.accept(iq)
        freq = self.e.frequency_trace(nblk)[0]
        steps = np.cumsum(np.concatenate([[before[0]], freq])[1:] != np.concatenate([[before[0]], freq])[:-1])
        assert before[1] + steps[-1] == self.e.scanner_tuned(0)[1]
        return p[0, :c[0]], freq, (before[1] + steps).astype(np.uint32)


@pytest.mark.parametrize("one_call", [False, True])
def test_frequency_scanner_matches_the_reference(capi, golden, one_call):
    g = golden["agc"]
    eng = capi.Engine(1, block_bytes=4096)
    eng.set_gain_trace(True)
    ch = ScanChain(eng, one_call)
    flags, pcm, freq, count, final = A.scan_scenario(ch, g["scan_iq"], 4096, ch.feed)
    assert np.array_equal(flags, g["scan_flags"])
    assert np.array_equal(freq, g["scan_freq"]) and np.array_equal(count, g["scan_count"])
    assert np.array_equal(pcm, g["scan_pcm"]) and np.array_equal(final, g["scan_final"])


@pytest.mark.parametrize("seed", range(300, 312))
def test_long_rows_with_a_running_agc_and_a_closing_squelch(capi, oracle, seed):
    """Thousands of blocks of one channel in ONE call (the wave-per-channel block loop): levels that sit still for a few
    hundred blocks, jitter inside and outside the deadband, jumps; both AGC types, blanking, a squelch that the moving
    gain opens and closes.  Since round 4 that loop takes the runs of blocks in which the AGC does not move at once;
    gains after every block, squelch decisions and PCM against the oracle fed block by block, over three calls."""
    rng = np.random.default_rng(seed)
    nblk = int(rng.integers(700, 3000))
    mags = []
    while len(mags) < nblk:
        level = int(rng.choice([0, 1, 3, 8, 20, 45, 90, 127, 150, 191]))
        jitter = int(rng.choice([0, 0, 1, 2, 6]))
        run = int(rng.choice([1, 2, 5, 40, 70, 200, 500]))
        mags += [int(np.clip(level + rng.integers(-jitter, jitter + 1), 0, 191)) for _ in range(run)]
    mags = mags[:nblk]
    cfg = dict(mode="am", type=int(rng.integers(0, 2)), deadband=int(rng.integers(0, 5)), blanking=int(rng.choice([0, 1, 2, 7, 12])),
               alpha=float(np.float32(rng.choice([0.05, 0.3, 0.8, 0.999]))), operating_point=int(rng.integers(-30, -5)),
               gain=int(rng.integers(0, 47)), threshold=int(rng.choice([-200, -60, -40, -30])))
    iq = np.concatenate([block_with_magnitude(m) for m in mags])
    ref = oracle.chain()
    A.configure(ref, cfg)
    pcm_o, allowed_o, gains_o = A.stream(ref, iq, BB)
    eng = capi.Engine(2, block_bytes=BB)
    eng.set_gain_trace(True)

    class Ch:
        pass
    ch = Ch()
    for name in ("set_mode", "set_squelch", "set_rx_gain_db", "agc_set_type", "agc_set_deadband", "agc_set_blanking_limit",
                 "agc_set_filter_coefficient", "agc_set_operating_point", "agc_enable"):
        setattr(ch, name, (lambda fn: (lambda v=True: fn(v, first=1, n=1)))(getattr(eng, name)))
    A.configure(ch, cfg)
    cuts = sorted(int(c) for c in rng.integers(1, nblk, 2))
    pcm, allowed, gains = [], [], []
    for b0, b1 in zip([0] + cuts, cuts + [nblk]):
        if b1 == b0:
            continue
        p, c, m, a = eng.accept(iq[b0 * BB:b1 * BB], first=1, n=1)
        assert m[0].tolist() == mags[b0:b1]
        trace = eng.gain_trace(b1 - b0, first=1, n=1)[0]
        gains += list(trace[1:]) + [eng.rx_gain_db(1)]
        allowed += a[0].tolist()
        pcm.append(p[0, :c[0]])
    first_bad = np.flatnonzero(np.array(gains, np.uint32) != gains_o)
    assert first_bad.size == 0, (seed, cfg, int(first_bad[0]), mags[max(0, int(first_bad[0]) - 3):int(first_bad[0]) + 2])
    assert np.array_equal(np.array(allowed, np.uint8), allowed_o), (seed, cfg)
    assert np.array_equal(np.concatenate(pcm), pcm_o), (seed, cfg)
    assert eng.agc_state(0)["rx_gain_db"] == 24           # the neighbour stays untouched
