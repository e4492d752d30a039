"""CPU tier: a model of what travels from segment to segment in the FM / AM / SSB streaming pipelines with short lead-ins
(round 6; VERDICT r5 item 3: "the restart-model CPU test extended to the new carried histories").

What runs REAL code: the plan's segmentation (`plan_stream`) and the segments' geometry (`d4_geom`: start, cold / warm, what a
cold one skips) through tests/emu, and the constants of the boundary replay (iqd_stream.h: D4_REPLAY_*, D4_RAILS_FROM_PIECE - the
ones `d4_am_wave` / `d4_fm_wave` are written with).  What is MODELLED: a consumer lane, stage by stage, with the ORACLE's
decimators (`iqo_decimate_q15`, the reference's `Decimator_int16` / `FirFilter_int16` from the zero state - an empty history
is exactly what a lane starts its run with):

    a lane's run        the stages behind the P waves over [v0 - 128, v0 + tile_len) from empty histories; the P waves' values are
                        exact from the run's first sample (FM: from its third discriminator output);
    what it keeps       AM / SSB: the stage-2 pairs of pieces 4..7 (both rails); SSB: the 8 kS/s rails of pieces 8..39; FM: the
                        stage-2 pairs of pieces 4..23 - the lane's head store;
    the hand-off        the predecessor's END state = the stage histories at v0 (taken from the truth: the predecessor's own
                        outputs are held to the truth by this very test);
    the replay          the first 4 (AM) / 36 (SSB) / 20 (FM) outputs again, from what was kept, behind that state.

The truth is the same chain run once over the whole row.  Every output of every segment - warm, cold (a channel's first, every
lane 0), short last ones - must be the truth's, for plans as `plan_stream` makes them and several first segment ids.  The
test has teeth: the chains' windows reach 4 (AM), 34 (SSB), 18 (FM) outputs into a segment; with replays of 3 / 28 / 14 the
model's segments differ from the truth (the outermost taps are small - a window that misses by one or two outputs usually still
rounds to the same value, which is why the lengths are the windows' and not what a data set happens to need)."""
import ctypes as C

import numpy as np
import pytest

from tests import emu_bind

PAD = 2048          # zeros in front of the row: the kept tail of a channel that has seen nothing yet (a multiple of 128)


@pytest.fixture(scope="module")
def L():
    lib = emu_bind.lib()
    lib.emu_d4_geom.restype = None
    lib.emu_d4_geom.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emu_d4_const.restype = C.c_uint32
    lib.emu_d4_const.argtypes = [C.c_int]
    lib.emu_plan_stream2.restype = None
    lib.emu_plan_stream2.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    return lib


def segments(L, vlen, n_ch, streams, shift, sid0):
    tl, nt = C.c_uint32(), C.c_uint32()
    L.emu_plan_stream2(vlen, n_ch, streams, 128, shift, C.byref(tl), C.byref(nt))
    out = []
    for t in range(nt.value):
        v0, skip, cold = C.c_int64(), C.c_uint32(), C.c_uint32()
        L.emu_d4_geom(sid0 + t, t, tl.value, shift, C.byref(v0), C.byref(skip), C.byref(cold))
        if v0.value < vlen:
            out.append((v0.value, skip.value, cold.value))
    return tl.value, out


class Chain:
    """One family's consumer-side stages as the oracle's decimators; `truth` runs them over the whole (padded) row."""

    def __init__(self, oracle, family, rng, n):
        self.o, self.family = oracle, family
        t = oracle.taps_f32
        self.n = n
        if family == "fm":
            # the discriminator's int16 stream at Fs / 4 (any values do: the hand-off is about the two decimators behind it);
            # a few loud ones, so that the decimators' per-product clamps are in play
            e = rng.integers(-9000, 9000, (PAD + n) // 4).astype(np.int16)
            e[rng.integers(0, len(e), 12)] = 32767
            e[:PAD // 4] = 0
            self.e = e
            self.y2 = oracle.decimate_q15(t("fm_post"), 4, e)
            self.out = oracle.decimate_q15(t("audio40"), 2, self.y2).astype(np.int32)
        else:
            x = rng.integers(-128, 128, (2, PAD + n)).astype(np.int16)
            x[:, :PAD] = 0
            self.y1 = [oracle.decimate_q15(t("am_s1"), 4, r) for r in x]
            self.y2 = [oracle.decimate_q15(t("am_s2"), 4, r) for r in self.y1]
            self.y3 = [oracle.decimate_q15(t("am_s3"), 2, r) for r in self.y2]
            self.out = self.detect(self.y3[0], self.y3[1])

    def detect(self, i8k, q8k):
        if self.family == "am":      # AmDemodulator.cc:446-459 (memoryless)
            im, qm = np.abs(i8k.astype(np.int32)).astype(np.int16), np.abs(q8k.astype(np.int32)).astype(np.int16)
            return np.where(im > qm, im + (qm >> 1), qm + (im >> 1)).astype(np.int16).astype(np.int32)
        t = self.o.taps_f32          # SsbDemodulator.cc:574-588: the delayed I rail -+ the Hilbert transformer over Q (lsb: minus)
        a = self.o.decimate_q15(t("ssb_delay"), 1, i8k).astype(np.int32)
        b = self.o.decimate_q15(t("ssb_hilbert"), 1, q8k).astype(np.int32)
        return a - b if self.family == "lsb" else a + b

    def segment(self, v0, tile_len, skip, cold, replay):
        """The outputs a lane stores for its segment: positions [v0 + skip, v0 + tile_len) / 32."""
        o, t = self.o, self.o.taps_f32
        a, b, at = (v0 - 128 + PAD), (v0 + tile_len + PAD), (v0 + PAD)
        assert a >= 0
        if self.family == "fm":
            ew = self.e[a // 4:b // 4].copy()
            ew[:2] = 0                                                   # (never delivered: the consumer's `early` pair starts empty)
            y2w = o.decimate_q15(t("fm_post"), 4, ew)
            loop = o.decimate_q15(t("audio40"), 2, y2w).astype(np.int32)
            if cold:
                return loop[(128 + skip) // 32:]
            hist, head = self.y2[at // 16 - 40:at // 16], y2w[8:8 + 2 * replay]
            again = o.decimate_q15(t("audio40"), 2, np.concatenate([hist, head])).astype(np.int32)[20:]
            return np.concatenate([again, loop[4 + replay:]])
        y2w = [o.decimate_q15(t("am_s2"), 4, r[a // 4:b // 4]) for r in self.y1]
        y3w = [o.decimate_q15(t("am_s3"), 2, r) for r in y2w]
        loop = self.detect(y3w[0], y3w[1])
        if cold:
            return loop[(128 + skip) // 32:]
        pairs = self.pairs
        y3r = [o.decimate_q15(t("am_s3"), 2, np.concatenate([self.y2[k][at // 16 - 14:at // 16], y2w[k][8:8 + 2 * pairs]]))[7:]
               for k in (0, 1)]                                          # stage 3 again: the predecessor's last 7 pairs, then the kept ones
        if self.family == "am":
            return np.concatenate([self.detect(y3r[0], y3r[1])[:replay], loop[4 + replay:]])
        rails = [np.concatenate([self.y3[k][at // 32 - 30:at // 32], y3r[k], y3w[k][self.rails_from:4 + replay]]) for k in (0, 1)]
        return np.concatenate([self.detect(rails[0], rails[1])[30:], loop[4 + replay:]])


CASES = [   # samples per channel, channels, segments the launch holds (256 CUs x 192), first segment ids
    (1 << 16, 4096, 256 * 192, (0, 37, 60)),       # 12 segments per channel: the bench's shape
    (1 << 14, 4096, 256 * 192, (0, 63, 125)),      # three
    (1 << 16, 16, 256 * 192, (0, 1000)),           # one channel cut into very many (several lane-0 segments among them)
    (8192 + 3 * 128, 700, 256 * 64, (5, 62)),      # a short last segment
]


@pytest.mark.parametrize("family", ["am", "usb", "lsb", "fm"])
def test_every_segment_stores_the_truth(L, oracle, family):
    fam_const = {"am": (4, 8), "fm": (5, 9)}.get(family, (6, 10))
    shift, replay = L.emu_d4_const(fam_const[0]), L.emu_d4_const(fam_const[1])
    rng = np.random.default_rng(2026 + len(family))
    n_seg = n_warm = n_cold_inside = 0
    for vlen, n_ch, streams, sid0s in CASES:
        ch = Chain(oracle, family, rng, vlen + 4096)
        ch.pairs, ch.rails_from = L.emu_d4_const(11), L.emu_d4_const(12)
        for sid0 in sid0s:
            tile_len, segs = segments(L, vlen, n_ch, streams, shift, sid0)
            for t, (v0, skip, cold) in enumerate(segs):
                got = ch.segment(v0, tile_len, skip, cold, replay)
                first, end = v0 + skip, min(v0 + tile_len, vlen)
                want = ch.out[(first + PAD) // 32:(end + PAD) // 32]
                assert np.array_equal(got[:len(want)], want), (family, vlen, n_ch, sid0, t, v0, cold,
                                                               np.flatnonzero(got[:len(want)] != want)[:5])
                n_seg += 1
                n_warm += not cold
                n_cold_inside += bool(cold and t > 0)
    assert n_seg > 100 and n_warm > 80 and n_cold_inside >= 2


@pytest.mark.parametrize("family,reach,too_short", [("am", 4, 3), ("usb", 34, 28), ("fm", 18, 14)])
def test_the_replay_is_as_long_as_the_chains_reach(L, oracle, family, reach, too_short):
    """The shipped replay lengths cover the chains' windows, and a replay a few outputs short of them is caught by this model."""
    fam_const = {"am": (4, 8), "fm": (5, 9)}.get(family, (6, 10))
    shift, replay = L.emu_d4_const(fam_const[0]), L.emu_d4_const(fam_const[1])
    assert replay >= reach and replay % 4 == 0
    rng = np.random.default_rng(5)
    vlen = 1 << 15
    ch = Chain(oracle, family, rng, vlen + 4096)
    ch.pairs, ch.rails_from = L.emu_d4_const(11), L.emu_d4_const(12)
    tile_len, segs = segments(L, vlen, 4096, 256 * 192, shift, 1)
    warm = [(v0, skip) for v0, skip, cold in segs if not cold]
    assert len(warm) >= 4

    def all_equal(n_replayed):
        for v0, skip in warm:
            got = ch.segment(v0, tile_len, skip, 0, n_replayed)
            want = ch.out[(v0 + PAD) // 32:(min(v0 + tile_len, vlen) + PAD) // 32]
            if not np.array_equal(got[:len(want)], want):
                return False
        return True

    assert all_equal(replay) and all_equal(reach) and not all_equal(too_short)
