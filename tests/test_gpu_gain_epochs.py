"""GPU (MI355X): setDemodulatorGain() at run time is a plain store that takes effect with the next sample
(WbFmDemodulator/WbFmDemodulator.cc:341-348, 444-450; FmDemodulator/FmDemodulator.cc twin).  The engine rebuilds
its post-discriminator histories from the raw tail, so it has to remember every gain whose samples are still in
that tail: any sequence of changes, however close together, must give the reference's PCM."""
import numpy as np
import pytest

from rtlsdrdiags_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from rtlsdrdiags_amd import capi as c
    return c


GAINS = [91000.0, 700.0, 40743.665, 3.0, 250000.0, 12000.0, 1.0e6, 55.5, 18000.0]


@pytest.mark.parametrize("mode,demod", [("wbfm", 3), ("fm", 2)])
@pytest.mark.parametrize("flags", [0x2, 0x4], ids=["tiles", "stream"])
def test_gain_changes_on_consecutive_256_byte_calls(capi, oracle, mode, demod, flags):
    """block_bytes 256: a change before every one of nine consecutive calls (128 samples apart), then longer runs."""
    u8 = synth.fm_tone(40 * 128 + 3 * 16384, seed=17, deviation=50e3)
    c = oracle.chain()
    c.set_mode(mode)
    eng = capi.Engine(1, block_bytes=256, flags=flags)
    eng.set_mode(mode)
    ref, out = [], []
    off = 0
    plan = [(256, g) for g in GAINS] + [(256, None)] * 3 + [(512, GAINS[2]), (256, GAINS[5]), (2048, None), (256, GAINS[0]),
                                                           (4096, GAINS[1]), (256, None), (32768, GAINS[3]), (65536, None)]
    for nbytes, gain in plan:
        if gain is not None:
            c.set_gain(demod, gain)
            eng.set_gain(mode, gain)
        piece = u8[off:off + nbytes]
        off += nbytes
        r, _, _ = c.accept_stream(piece, 256)
        pcm, cnt, _, _ = eng.accept(piece)
        ref.append(r)
        out.append(pcm[0, :cnt[0]])
    for k in range(len(plan)):
        assert np.array_equal(out[k], ref[k]), (k, plan[k])


def test_two_changes_between_two_accepts_and_a_front_end_call_in_between(capi, oracle):
    """set_gain(B), front_end() (uploads the parameters), set_gain(C), accept: the histories were made with A."""
    u8 = synth.fm_tone(4 * 16384, seed=23, deviation=60e3)
    c = oracle.chain()
    c.set_mode("wbfm")
    eng = capi.Engine(1)
    eng.set_mode("wbfm")
    a0, _, _ = c.accept_stream(u8[:32768])
    p0, n0, _, _ = eng.accept(u8[:32768])
    c.set_gain(3, 5000.0)
    eng.set_gain("wbfm", 5000.0)
    eng.front_end(u8[:32768])
    c.set_gain(3, 77777.0)
    eng.set_gain("wbfm", 77777.0)
    a1, _, _ = c.accept_stream(u8[32768:])
    p1, n1, _, _ = eng.accept(u8[32768:])
    assert np.array_equal(p0[0, :n0[0]], a0) and np.array_equal(p1[0, :n1[0]], a1)


@pytest.mark.parametrize("mode,demod", [("wbfm", 3), ("fm", 2)])
def test_gain_change_then_64_byte_calls(capi, oracle, mode, demod):
    """A gain change, one 64-byte call, then more: the lead-in of the next call has a chunk of 32 samples with the new gain
    behind a long one with the old (round 4: the FM chain's (int16)(K dtheta) history was not shifted behind a chunk that
    short; found by the short-block fuzzer).  Then a change before EVERY one of 70 consecutive 64-byte calls - more
    changes inside the filters' reach than the 16 the engine used to keep."""
    u8 = synth.fm_tone(6 * 16384, seed=29, deviation=45e3)
    c = oracle.chain()
    c.set_mode(mode)
    eng = capi.Engine(1, block_bytes=4096)
    eng.set_mode(mode)
    off = 0
    plan = [(118784, None), (64, GAINS[0]), (64, None), (64, GAINS[1]), (192, None), (64, GAINS[2]), (4096, None)]
    plan += [(64, GAINS[k % len(GAINS)] * (1 + 0.01 * k)) for k in range(70)] + [(64, None), (8192, None), (64, GAINS[4]), (32768, None)]
    for k, (nbytes, gain) in enumerate(plan):
        if gain is not None:
            c.set_gain(demod, gain)
            eng.set_gain(mode, gain)
        piece = u8[off:off + nbytes]
        off += nbytes
        r, _, _ = c.accept_stream(piece, min(nbytes, 4096))
        pcm, cnt, _, _ = eng.accept(piece)
        assert cnt[0] == len(r) and np.array_equal(pcm[0, :cnt[0]], r), (k, nbytes, gain)


def test_gated_engine_returns_to_the_one_launch_path_after_a_gain_change_has_aged_out(capi, oracle):
    """the host's mirror of "a gain change still lies inside this channel's tail" was aged by arithmetic for
    ungated calls only, so ONE iqd_set_gain on an engine whose squelch can close kept every later call off the one-launch
    arrangement for good.  Since round 4 a gated call's WBFM tail updates report it.  Mixed engine with a squelch that can
    close (every block is loud enough to pass): one launch, gain change -> a launch per family, then one launch again;
    PCM of the changed channel and of its neighbours against the oracle throughout."""
    n_ch, n = 1500, 1 << 14
    modes = ["am", "fm", "wbfm", "lsb", "usb"]
    base = [synth.fm_tone(n, seed=700 + k, deviation=3000.0 + 500.0 * k, amplitude=60.0) for k in range(7)]
    u8 = np.stack([np.roll(base[c % 7], 2 * ((c * 37) % 1009)) for c in range(n_ch)])
    eng = capi.Engine(n_ch)
    for c in range(n_ch):
        eng.set_mode(modes[c % 5], first=c, n=1)
    eng.set_squelch(-60)
    chains = {}
    for c in (2, 7, 12, 0, 1, 3, 4, 1497):
        o = oracle.chain()
        o.set_mode(modes[c % 5])
        o.set_squelch(-60)
        chains[c] = o

    def call():
        pcm, cnt, _, allowed = eng.accept(u8)
        assert allowed.all()
        for c, o in chains.items():
            ref, _, _ = o.accept_stream(u8[c])
            assert cnt[c] == len(ref) and np.array_equal(pcm[c, :cnt[c]], ref), c
        return eng.stats()["mixed_launches"]

    assert call() == 1                                   # one launch for the four families (gated instantiation)
    eng.set_gain("wbfm", 700.0, first=7, n=1)
    chains[7].set_gain(3, 700.0)
    assert call() == 1                                   # the change lies inside channel 7's tail: a launch per family
    assert call() == 2                                   # ... it has aged out, the device said so: one launch again
    assert call() == 3
    eng.close()
