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


def _by_time(vlen, counts, n_wgs=256, ns=(21.7, 38.2, 58.3, 21.0)):
    """counts: per family (AM, FM, WBFM, SSB) the channels per rotation selector."""
    lib = emu_bind.lib()
    lib.emu_plan_fused_by_time.restype = C.c_int
    lib.emu_plan_fused_by_time.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    halo, gran = (384, 768, 768, 1280), (128, 128, 512, 128)
    fam = np.array([list(counts[f]) + [halo[f], gran[f], 0] for f in range(4)], np.uint32)
    nsv = np.array(ns, np.float32)
    out = np.full(4, 77, np.uint32)
    ok = lib.emu_plan_fused_by_time(vlen, 4, fam.ctypes.data, nsv.ctypes.data, n_wgs, out.ctypes.data)
    return bool(ok), out


def _family_time(vlen, rot_counts, wgs, halo, gran, ns):
    """What the engine then does with a share (plan_stream: the shortest segments that fit one round; once more with 48
    segments fewer if the rotation groups' padding overflows it) -> ns, or None if it takes a second round."""
    n = sum(rot_counts)
    for spare in (0, 48):
        per = max(1, (wgs * 192 - spare) // n)
        ln = -(-vlen // per)
        ln = max(768, -(-ln // gran) * gran)
        if ln > 768 and -(-vlen // 768) <= per:
            ln = 768
        tiles = -(-vlen // ln)
        ids = sum(-(-(c * tiles) // 16) * 16 for c in rot_counts)
        if ids <= wgs * 192:
            return ns * (ln + halo)
    return None


def test_shares_by_time_fit_one_round_and_beat_proportional_shares_where_rows_are_short():
    halo, gran, ns = (384, 768, 768, 1280), (128, 128, 512, 128), (21.7, 38.2, 58.3, 21.0)
    lib = emu_bind.lib()
    lib.emu_plan_fused_shares.restype = None
    lib.emu_plan_fused_shares.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
    rng = np.random.default_rng(11)
    gains = []
    for case in range(600):
        vlen = int(rng.choice([4096, 8192, 16384, 32768, 65536, 1 << 18]))
        n_ch = int(rng.integers(40, 20000))
        fams = rng.integers(0, 5, n_ch)
        counts = []
        for f in range(4):
            k = int(np.count_nonzero(fams == f) + (np.count_nonzero(fams == 4) if f == 3 else 0))
            if f == 2:
                counts.append((k, 0, 0))                              # WBFM: one selector in the one-launch arrangement
            else:
                a = int(rng.integers(0, k + 1)); b = int(rng.integers(0, k - a + 1))
                counts.append((a, b, k - a - b))
        if rng.random() < 0.2:
            counts[int(rng.integers(0, 4))] = (0, 0, 0)               # a family that is not there
        ok, share = _by_time(vlen, counts)
        present = [sum(c) > 0 for c in counts]
        if not ok:
            assert share.tolist() == [0, 0, 0, 0]
            # then not even one segment per channel fits a round of 256 workgroups
            assert sum(sum(-(-c // 16) * 16 for c in counts[f]) for f in range(4)) > 256 * 192 - 3 * 192, (vlen, counts)
            continue
        assert share.sum() == 256 and all((share[f] > 0) == present[f] for f in range(4)), (counts, share)
        times = [_family_time(vlen, counts[f], int(share[f]), halo[f], gran[f], ns[f]) for f in range(4) if present[f]]
        assert all(t is not None for t in times), (vlen, counts, share)          # every family fits ONE round of its share
        # against the proportional shares the engine used before
        cost = np.array([(3.4, 6.3, 10.8, 3.6)[f] * sum(counts[f]) for f in range(4)], np.float32)
        prop = np.zeros(4, np.uint32)
        lib.emu_plan_fused_shares(cost.ctypes.data, 4, 256, prop.ctypes.data)
        pt = [_family_time(vlen, counts[f], int(prop[f]), halo[f], gran[f], ns[f]) for f in range(4) if present[f]]
        if all(t is not None for t in pt):
            assert max(times) <= max(pt) * 1.0001, (vlen, counts, share, prop)   # never worse than the proportional plan
            gains.append(max(times) / max(pt))
    assert len(gains) > 200 and min(gains) < 0.8                                 # and clearly better in some cases
    # 16 384 channels x 2^14 (one block per channel and call): AM and SSB get two segments per channel instead of one
    ok, share = _by_time(16384, [(3276, 0, 0), (3276, 0, 0), (3277, 0, 0), (6555, 0, 0)])
    assert ok and share.tolist() == [38, 56, 90, 72]      # (35 / 52 / 86 / 69 meet the deadline; the 14 left over go round in proportion)
