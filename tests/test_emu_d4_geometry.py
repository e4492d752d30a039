"""CPU tier: the segment geometry of the FM / AM / SSB streaming pipelines with short lead-ins (round 6; iqd_stream.h: d4_geom,
iqd_host.cpp: plan_stream) - the one function the kernels (d4_segment) and the plan share, compiled for the host (tests/emu).

A warm segment runs 128 samples of lead-in and replays its first outputs with the end state of its predecessor, the lane below
in its consumer wave; a cold one (a channel's first segment; a wave's lane 0 = every segment id that is a multiple of 64) runs
the family's full lead-in out of its own length.  Checked here, for plans as plan_stream makes them and for every first segment
id a channel can have: the segments' output ranges tile the row exactly once, every warm segment's predecessor ends exactly
where it starts and sits in the lane below, every run stays inside the kept tail's reach, and the plan's segment count is
enough whatever the first id."""
import ctypes as C

import numpy as np
import pytest

from tests import emu_bind


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


def geom(L, sid, tile, tile_len, shift):
    v0, skip, cold = C.c_int64(), C.c_uint32(), C.c_uint32()
    L.emu_d4_geom(sid, tile, tile_len, shift, C.byref(v0), C.byref(skip), C.byref(cold))
    return v0.value, skip.value, cold.value


def test_segments_tile_every_row_once_whatever_the_channels_first_id(L):
    rng = np.random.default_rng(77)
    tail = L.emu_d4_const(7)
    n_cases = 0
    for shift, keep in ((L.emu_d4_const(4), L.emu_d4_const(13)), (L.emu_d4_const(5), L.emu_d4_const(14)),
                        (L.emu_d4_const(6), L.emu_d4_const(15))):                     # AM 256, FM 640, SSB 1152; what the family keeps of a channel's stream
        assert shift + 128 + 32 <= keep <= tail
        for trial in range(120):
            vlen = 128 * int(rng.integers(1, 1 << int(rng.integers(1, 12))))
            n_ch = int(rng.integers(1, 5000))
            streams = 64 * int(rng.integers(1, 4)) * int(rng.integers(1, 257))
            tl, nt = C.c_uint32(), C.c_uint32()
            L.emu_plan_stream2(vlen, n_ch, streams, 128, shift, C.byref(tl), C.byref(nt))
            tile_len, n_tiles = tl.value, nt.value
            assert tile_len % 128 == 0 and tile_len >= shift + 128
            for sid0 in {0, 1, 63, 64, int(rng.integers(0, 1 << 20)), 64 * int(rng.integers(0, 1000)) + 64 - min(n_tiles, 63)}:
                n_cases += 1
                at = 0                                              # the next output sample nobody has produced yet
                for t in range(n_tiles):
                    v0, skip, cold = geom(L, sid0 + t, t, tile_len, shift)
                    assert cold == (1 if (t == 0 or (sid0 + t) % 64 == 0) else 0) and skip == (shift if cold else 0)
                    if v0 >= vlen:                                  # not there (the kernel: invalid) - then nothing behind it is either
                        assert at >= vlen, (vlen, tile_len, n_tiles, sid0, t)
                        continue
                    first, end = v0 + skip, min(v0 + tile_len, vlen)
                    assert first == at, (vlen, tile_len, sid0, t, first, at)          # outputs start where the predecessor's ended
                    assert v0 - 128 - 32 >= -keep                                     # the run (lead-in + the piece before it) stays inside what the closing launch kept (tail_keep)
                    if not cold:                                                      # its predecessor: the lane below, ending at v0
                        pv0, _, _ = geom(L, sid0 + t - 1, t - 1, tile_len, shift)
                        assert pv0 + tile_len == v0 and (sid0 + t) % 64 != 0
                    at = max(at, end)
                assert at == vlen, (vlen, tile_len, n_tiles, sid0, at)                # the row is covered, by the plan's segment count
    assert n_cases > 1500


def test_no_shift_is_the_geometry_of_rounds_2_to_5(L):
    for sid, t in ((0, 0), (64, 3), (1000, 7)):
        assert geom(L, sid, t, 5504, 0) == (t * 5504, 0, 1)
