"""CPU tier: a model of the WBFM restart state's journey from call to call through tile-kernel and streaming calls in any
alternation (round 4's three restart-state bugs all lived here and only the GPU fuzz campaigns found them).

What runs REAL code: the tile path (the kernel's phase functions through tests/emu, `emu_wbfm_accept`), the streaming call's
segmentation (`plan_stream`), where a streamed segment takes its record and what the commit picks (`st_rec_plan`,
`wbfm_pick_carry`: iqd_wbfm.h - the very functions `st_iir_wave` and `tail_update_body` call on the GPU).  What is MODELLED here:
a streamed segment's de-emphasis lane - the recurrence from a cold (zero) or carried state over the u[n] stream, and the marks
`st_iir_marks` takes at window starts - in numpy binary32, op by op.

The oracle gives the truth: `iqo_wbfm_stages` returns the wrapped phase steps and the exact de-emphasis output of the whole
stream, i.e. the state (y, u) entering every sample.  After EVERY call the carried restart state must be that exact state at
`end - back`, bit for bit, `back` must lie within the streaming lead-in's reach, and a cold segment's warmed-up state must
chain up with its predecessor's end.

The test has teeth: with `wbfm_pick_carry` made to ignore the keeper's parked state, or `st_rec_plan` made never to appoint a
keeper (round 4's second find, either way), `test_every_pair_of_calls_in_every_alternation` fails within the first hundred
sequences (checked by hand when the test was written; the lengths 896 / 1024 / 1152 are in the set for that)."""
import ctypes as C
import itertools

import numpy as np
import pytest

from rtlsdrdiags_amd import synth
from tests import emu_bind

f32 = np.float32


class Model:
    def __init__(self, L, oracle, u8, rotation=1):
        self.L = L
        L.emu_st_rec_plan.restype = None
        L.emu_st_rec_plan.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.emu_wbfm_pick_carry.restype = None
        L.emu_wbfm_pick_carry.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.emu_restart_consts.restype = None
        L.emu_restart_consts.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.emu_plan_stream.restype = None
        L.emu_plan_stream.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        consts, ab, k = np.zeros(6, np.uint32), np.zeros(2, np.float32), C.c_float(0)
        self.gain = f32(256000 / (2 * np.pi))
        L.emu_restart_consts(consts.ctypes.data, ab.ctypes.data, self.gain, C.byref(k))
        self.FORCED_BACK, self.ST_HALO, self.MIN_TILE, self.STREAMED, self.TAIL, self.CHUNK = (int(v) for v in consts)
        self.a1, self.b0, self.k = f32(ab[0]), f32(ab[1]), f32(k.value)
        # the truth: the whole stream through the oracle's WBFM stages
        s8 = (u8 ^ 0x80).view(np.int8)
        rot = np.concatenate([oracle.rotate(s8[o:o + 32768], rotation) for o in range(0, len(s8), 32768)])
        st = oracle.wbfm_stages(rot, float(self.gain))
        self.u = (self.b0 * (self.k * st["dtheta"].astype(np.float32))).astype(np.float32)   # b0 * (K * d), each product rounded
        self.y = st["deemph"].astype(np.float32)                                             # exact output = state after sample n
        self.u8 = u8
        self.rotation = rotation

    # exact state ENTERING global sample P
    def exact(self, P):
        return (f32(0), f32(0)) if P <= 0 else (self.y[P - 1], self.u[P - 1])

    def iir_run(self, y, up, g0, n):
        """n steps of IQD_IIR_STEP from global sample g0 (u = 0 before the stream's start)."""
        a1 = self.a1
        for g in range(g0, g0 + n):
            x = self.u[g] if g >= 0 else f32(0)
            tn = f32(x + up)
            r = f32(a1 * y)
            y = f32(tn - r)
            up = x
        return y, up

    @staticmethod
    def agree(a, b):
        return a.tobytes() == b.tobytes() or (abs(float(a)) < 2.0 ** -100 and abs(float(b)) < 2.0 ** -100)

    def stream_call(self, pos0, vlen, carry):
        """One streamed call of `vlen` samples starting at global sample pos0; carry = (y, u, back).  Returns the new carry, or
        a string saying what the streaming kernel could not have done."""
        tl, nt = C.c_uint32(), C.c_uint32()
        self.L.emu_plan_stream(vlen, 1, 256 * 192, C.byref(tl), C.byref(nt))
        tile_len, tiles = tl.value, nt.value
        recs = np.zeros((tiles, 8), np.uint32)
        prev_end = None
        for t in range(tiles):
            v0 = t * tile_len
            tlen = min(tile_len, vlen - v0)
            plan = np.zeros(4, np.int32)
            self.L.emu_st_rec_plan(1, t, v0, tlen, vlen, carry[2] if t == 0 else self.FORCED_BACK, plan.ctypes.data)
            rec_pos, back_out, park_pos, keeps = (int(v) for v in plan)
            if t == 0:
                if carry[2] > self.ST_HALO:
                    return "the carried restart point lies %d samples back: beyond the streaming lead-in (%d)" % (carry[2], self.ST_HALO)
                y, up, pos = carry[0], carry[1], -carry[2]
                y_in = carry[0]
            else:
                y, up, pos = f32(0), f32(0), -self.ST_HALO
                y_in = f32(0)
            y_out, u_out = (carry[0], carry[1]) if t == 0 else (f32(0), f32(0))   # (what the record holds until a mark overwrites it)
            pad = None
            y_end = u_end = f32(0)
            while True:                                       # marks at window starts (st_iir_marks), then 16 samples
                if t > 0 and pos == 0:
                    y_in = y
                if pos == park_pos:
                    if keeps:
                        pad = (y, up)
                    else:
                        y_out, u_out = y, up
                if keeps and pos == tile_len - self.FORCED_BACK:
                    y_out, u_out = y, up
                if pos == tlen:
                    y_end, u_end = y, up
                    break
                y, up = self.iir_run(y, up, pos0 + v0 + pos, 16)
                pos += 16
            if t > 0 and not self.agree(y_in, prev_end):
                return None                                   # (the hand-off check would flag it and the tile kernel repair it: not modelled)
            prev_end = y_end
            last = not (vlen - v0 > tlen)
            words = [y_in, y_out, u_out, None, y_end, u_end]
            recs[t, :3] = np.array(words[:3], np.float32).view(np.uint32)
            recs[t, 3] = np.uint32(np.int32(back_out).view(np.uint32))
            recs[t, 4:6] = np.array(words[4:6], np.float32).view(np.uint32)
            if keeps:
                recs[t, 6:8] = np.array(pad, np.float32).view(np.uint32)
            elif last:
                recs[t, 6] = self.STREAMED
        out = emu_bind.Carry()
        before = recs[tiles - 2] if tiles >= 2 else recs[0]
        self.L.emu_wbfm_pick_carry(recs[tiles - 1].ctypes.data, np.ascontiguousarray(before).ctypes.data, tiles, vlen, tile_len, 1, C.byref(out))
        return (f32(out.y), f32(out.u), int(out.back))


def run_sequence(m, L, lengths, paths):
    """lengths in samples; paths: 't' tile kernel, 's' streaming kernel.  Returns None or a description of what went wrong."""
    ch = emu_bind.WbfmChannel(L, m.CHUNK, rotation=m.rotation, gain=m.gain)
    pos = 0
    for k, (n, path) in enumerate(zip(lengths, paths)):
        if path == "s":
            got = m.stream_call(pos, n, (f32(ch.carry.y), f32(ch.carry.u), int(ch.carry.back)))
            if got is None:
                return None                                   # a hand-off that needs the repair path: outside the model
            if isinstance(got, str):
                return "call %d (%d samples, streamed): %s" % (k, n, got)
            ch.carry.y, ch.carry.u, ch.carry.back = float(got[0]), float(got[1]), got[2]
            # the kept tail, as tail_update_body leaves it
            raw = m.u8[2 * pos:2 * (pos + n)]
            joined = np.concatenate([ch.tail, raw])
            ch.tail[:] = joined[-4096:]
            ye, ue = m.exact(pos + n)
            ch.carry.y_end, ch.carry.u_end = float(ye), float(ue)
        else:
            ch.accept(m.u8[2 * pos:2 * (pos + n)])
        pos += n
        back = int(ch.carry.back)
        if back > m.ST_HALO or back > pos or back < 0:
            return "call %d (%d samples, %s): back = %d (stream so far %d, lead-in %d)" % (k, n, path, back, pos, m.ST_HALO)
        ye, ue = m.exact(pos - back)
        if not (m.agree(f32(ch.carry.y), ye) and m.agree(f32(ch.carry.u), ue)):
            return "call %d (%d samples, %s): carried (%r, %r) at end - %d is not the exact state (%r, %r)" % (
                k, n, path, ch.carry.y, ch.carry.u, back, float(ye), float(ue))
    if ch.hand_off_mismatches:
        return "tile hand-off mismatches: %d" % ch.hand_off_mismatches
    return None


@pytest.fixture(scope="module")
def model(oracle):
    L = emu_bind.lib()
    L.emu_wbfm_driver(0)
    u8 = synth.fm_tone(4 * 5696 // 2 * 2 + 4096, seed=71)     # enough for the longest sequence
    m = Model(L, oracle, u8)
    # the model's recurrence is the oracle's: re-run it over the whole stream from the zero state
    y, up = m.iir_run(f32(0), f32(0), 0, 3000)
    assert y.tobytes() == m.y[2999].tobytes() and up.tobytes() == m.u[2999].tobytes()
    return m, L


def paths_for(lengths):
    """every alternation of tile and streaming calls (a call streams only if it is whole 128-sample units)"""
    return itertools.product(*[("t", "s") if n % 128 == 0 else ("t",) for n in lengths])


# bytes / 2: 32 .. 768 samples in steps of 32, the two lengths around the bench's segment, and three that cut a
# streamed call into a full segment and a SHORT last one (896 = 768 + 128 ...: round 4's second find needs a last segment too short
# for a cold start to have converged)
ALL_LENGTHS = [b // 2 for b in list(range(64, 1536 + 1, 64)) + [5632, 5696, 1792, 2048, 2304]]


def test_every_pair_of_calls_in_every_alternation(model):
    m, L = model
    n = 0
    for lengths in itertools.product(ALL_LENGTHS, repeat=2):
        for paths in paths_for(lengths):
            bad = run_sequence(m, L, lengths, paths)
            assert bad is None, (lengths, paths, bad)
            n += 1
    assert n > 1200


def test_triples_and_quadruples_around_the_lead_in(model):
    """Depth 3 exhaustively over the lengths that sit on and around the 768-sample restart distance and the 128-sample grid,
    depth 4 over a seeded sample of everything."""
    m, L = model
    core = [32, 96, 128, 384, 640, 736, 768, 896, 1024, 2816, 2848]
    n = 0
    for lengths in itertools.product(core, repeat=3):
        for paths in paths_for(lengths):
            bad = run_sequence(m, L, lengths, paths)
            assert bad is None, (lengths, paths, bad)
            n += 1
    rng = np.random.default_rng(5)
    for _ in range(400):
        lengths = [int(x) for x in rng.choice(ALL_LENGTHS, 4)]
        paths = [("s" if x % 128 == 0 and rng.random() < 0.6 else "t") for x in lengths]
        bad = run_sequence(m, L, lengths, paths)
        assert bad is None, (lengths, paths, bad)
        n += 1
    assert n > 1500


def test_the_model_sees_round_4s_first_find(model):
    """Round 4, campaign find (1): a tile kernel that ended a call off the 128-sample grid took its restart record up to 96
    samples further back than the streaming lead-in reaches.  The model refuses such a carry - the check that would have
    caught it."""
    m, L = model
    got = m.stream_call(2000, 1024, (f32(0.5), f32(0.1), m.ST_HALO + 96))
    assert isinstance(got, str) and "beyond the streaming lead-in" in got
