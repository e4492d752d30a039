"""CPU tier: the engine's host-side planning for the streaming kernels (iqd_host.cpp), through the emulation library."""
import ctypes as C

import numpy as np
import pytest

import emu_bind


@pytest.fixture(scope="module")
def L():
    lib = emu_bind.lib()
    lib.emu_plan_family_shares.restype = C.c_int
    lib.emu_plan_family_shares.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
    lib.emu_plan_stream.restype = None
    lib.emu_plan_stream.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    return lib


def shares(L, cost, n_cus=256):
    cost = np.asarray(cost, np.float32)
    out = np.zeros(len(cost), np.uint32)
    ok = L.emu_plan_family_shares(cost.ctypes.data, len(cost), n_cus, out.ctypes.data)
    return bool(ok), out


def test_family_shares_fit_side_by_side_on_every_xcd():
    rng = np.random.default_rng(3)
    L_ = emu_bind.lib()
    L_.emu_plan_family_shares.restype = C.c_int
    L_.emu_plan_family_shares.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
    for _ in range(2000):
        n = 4
        cost = rng.uniform(0.0, 10000.0, n).astype(np.float32)
        cost[rng.random(n) < 0.3] = 0.0                      # absent families
        if not np.any(cost > 0):
            continue
        ok, s = shares(L_, cost)
        present = cost > 0
        assert np.all(s[~present] == 256)                    # absent: untouched (all CUs)
        if not ok:
            assert np.all(s == 256)
            continue
        assert np.all(s[present] % 8 == 0) and np.all(s[present] >= 8)
        assert s[present].sum() <= 240                       # two CUs per XCD stay unplanned
        assert s[present].sum() > 240 - 8 or np.count_nonzero(present) == 0   # nothing left lying around
        due = 240 * cost[present] / cost[present].sum()
        assert np.all(s[present] >= np.minimum(8, due) - 1e-3)
        assert np.all(np.abs(s[present].astype(np.float64) - due) < 16 + 8 * np.count_nonzero(present)), (cost, s)


def test_fused_shares_hand_out_every_workgroup_in_proportion():
    """Several families as ranges of ONE launch (iqd_stream_mixed.hip): plain integers, all workgroups given out."""
    lib = emu_bind.lib()
    lib.emu_plan_fused_shares.restype = None
    lib.emu_plan_fused_shares.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
    rng = np.random.default_rng(5)
    for _ in range(2000):
        cost = rng.uniform(0.0, 10000.0, 4).astype(np.float32)
        cost[rng.random(4) < 0.3] = 0.0
        if rng.random() < 0.1:
            cost[cost > 0] *= rng.choice([1e-4, 1.0], size=np.count_nonzero(cost > 0)).astype(np.float32)   # a tiny family beside big ones
        n_wgs = int(rng.choice([64, 240, 256, 304]))
        out = np.full(4, 77, np.uint32)
        lib.emu_plan_fused_shares(cost.ctypes.data, 4, n_wgs, out.ctypes.data)
        present = cost > 0
        assert np.all(out[~present] == 0)
        if not present.any():
            continue
        assert out.sum() == n_wgs and np.all(out[present] >= 1)
        due = n_wgs * cost[present].astype(np.float64) / cost[present].astype(np.float64).sum()
        assert np.all(np.abs(out[present] - due) < 1.0 + np.count_nonzero(present)), (cost, out)
    out = np.zeros(4, np.uint32)
    cost = np.array([3.4 * 819, 6.3 * 819, 10.8 * 820, 3.6 * 1638], np.float32)   # the mixed bench configuration
    lib.emu_plan_fused_shares(cost.ctypes.data, 4, 256, out.ctypes.data)
    assert out.tolist() == [31, 58, 100, 67]


def test_family_shares_of_the_mixed_bench_configuration(L):
    # 820 AM, 820 FM, 820 WBFM, 1638 SSB channels with the engine's weights 3.4 / 5.2 / 9.0 / 2.9
    ok, s = shares(L, [3.4 * 820, 5.2 * 820, 9.0 * 820, 2.9 * 1638])
    assert ok and s.tolist() == [32, 56, 96, 56]


def test_no_plan_on_a_small_device_or_without_work(L):
    assert shares(L, [1.0, 2.0, 0.0, 0.0], n_cus=32) == (False, pytest.approx(np.array([32, 32, 32, 32])))
    ok, s = shares(L, [0.0, 0.0, 0.0, 0.0])
    assert not ok and s.tolist() == [256] * 4


def test_stream_plan_covers_the_row_in_whole_quads_of_pieces(L):
    tile, tiles = C.c_uint32(), C.c_uint32()
    for vlen, n_ch, streams in [(1 << 28, 1, 49152), (65536, 4096, 49152), (65536, 820, 96 * 192), (2048, 60000, 49152),
                                (6 * 16384, 9, 49152), (128, 3, 49152), (1 << 21, 1, 49152)]:
        L.emu_plan_stream(vlen, n_ch, streams, C.byref(tile), C.byref(tiles))
        assert tile.value % 128 == 0 and tile.value >= 768          # whole quads of 32-sample pieces, at least the minimum tile
        assert tile.value * tiles.value >= vlen                     # the tiles cover the row
        assert tile.value * (tiles.value - 1) < vlen or tiles.value == 1   # and none of them is empty
