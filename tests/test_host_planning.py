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


# ---- plan_call (iqd_plan.cpp): the whole call's plan, decided before anything is queued ---------------------------------
FAMS = ("am", "fm", "wbfm", "ssb")
F_TILES, F_STREAM = 0x2, 0x4
PLAN_TILES, PLAN_STREAM = 0, 1


_PLAN_LIB = []


def _plan_lib():
    if not _PLAN_LIB:          # (emu_bind.lib() runs make: once)
        lib = emu_bind.lib()
        lib.emu_plan_call.restype = None
        lib.emu_plan_call.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.emu_plan_const.restype = C.c_uint32
        _PLAN_LIB.append(lib)
    return _PLAN_LIB[0]


def plan_call(vlen, fams, flags=0, n_cus=256, stream_ok=True, env_path=0, min_seg=0, am_min=0, mixed_forked=False, by_cost=False,
              stream_wgs=0, full_grid=False, gated=False, rings=0, full_lead_ins=False, leadfree=None):
    """fams: {family: (rot_count tuple | channels, cast_bounded, epochs_in_reach)} -> the plan as a dict."""
    lib = _plan_lib()
    knobs = np.array([flags, n_cus, int(stream_ok), {0: 0, 1: 1, -1: 2}[env_path], min_seg, am_min, int(mixed_forked), int(by_cost),
                      stream_wgs, int(full_grid), rings, 1 if full_lead_ins else (0 if leadfree is None else 1 + leadfree)], np.uint32)
    fam = np.zeros((4, 6), np.uint32)
    for f, name in enumerate(FAMS):
        if name in fams:
            rc, bounded, epochs = fams[name]
            rc = (rc, 0, 0) if isinstance(rc, int) else rc
            fam[f] = [sum(rc), rc[0], rc[1], rc[2], int(bounded), int(epochs)]
    out = np.zeros(9 + 4 * 21, np.uint32)
    lib.emu_plan_call(knobs.ctypes.data, vlen, vlen // 32, int(gated), fam.ctypes.data, out.ctypes.data)
    keys = ("present", "path", "lane", "wgs", "tile_len", "tiles_per_ch", "grouped", "gs0", "gs1", "gs2", "gs3", "gn0", "gn1", "gn2",
            "grid", "rounds", "wg_first", "epochs", "rings", "halo", "lead_shift")
    plan = {"n_fams": int(out[0]), "forked": bool(out[1]), "shares_on": bool(out[2]), "fused": bool(out[3]), "mix_wgs": int(out[4]),
            "order": out[5:9].tolist(), "fam": {}}
    for f, name in enumerate(FAMS):
        plan["fam"][name] = dict(zip(keys, (int(v) for v in out[9 + 21 * f: 9 + 21 * f + 21])))
        plan["fam"][name]["n"] = int(fam[f][0])
        plan["fam"][name]["rot_count"] = fam[f][1:4].tolist()
    return plan


def check_plan(p, vlen, flags, env_path, stream_ok, n_cus, fams, rings_knob=0):
    lib = _plan_lib()
    st_segs, min_tile, dc_tile = lib.emu_plan_const(0), lib.emu_plan_const(1), lib.emu_plan_const(3)
    present = [n for n in FAMS if p["fam"][n]["n"]]
    assert p["n_fams"] == len(present) and p["forked"] == (len(present) > 1)
    assert sorted(p["order"]) == [0, 1, 2, 3]
    if p["fused"]:
        assert p["shares_on"] and p["forked"]
    at = 0
    ranges = []
    for name in FAMS:
        q = p["fam"][name]
        assert bool(q["present"]) == (q["n"] > 0)
        if not q["n"]:
            continue
        # the tiles / segments cover the row - FM / AM / SSB pipelines with short lead-ins (round 6): a channel's cold segments (its
        # first one, and every lane 0 of a consumer wave: at most 1 + ceil((n - 1) / 64) of its n segments) spend `lead_shift`
        # samples of their length on their lead-in (tests/test_emu_d4_geometry.py walks the geometry itself)
        max_cold = 1 + (q["tiles_per_ch"] + 62) // 64 if q["lead_shift"] else 0
        assert q["tile_len"] > 0 and q["tile_len"] * q["tiles_per_ch"] >= vlen + q["lead_shift"] * max_cold, (name, q)
        assert 0 <= q["lane"] <= 3 and (q["lane"] == 0 or (p["forked"] and not p["fused"]))
        rc, bounded, epochs = fams[name]
        if q["path"] == PLAN_STREAM:
            assert vlen % 128 == 0 and not (flags & F_TILES) and env_path >= 0, (name, q)
            assert q["tile_len"] >= min_tile and q["tile_len"] % 128 == 0
            if name == "wbfm":
                assert q["lead_shift"] == 0
            else:   # short lead-ins: 128 samples, the rest of the family's full lead-in as the cold segments' shift, segments longer than
                full = {"am": 384, "fm": 768, "ssb": 1280}[name]   # the shift; by default only where a channel has 8 segments or more and never in the one launch
                assert (q["halo"], q["lead_shift"]) in ((128, full - 128), (full, 0)), (name, q)
                if q["lead_shift"]:
                    assert q["tile_len"] >= full and not p["fused"] and q["tiles_per_ch"] >= 8, (name, q)
            assert q["grid"] >= 1 and q["rounds"] >= 1 and q["grid"] <= max(q["wgs"], 1)
            if name == "wbfm":
                assert stream_ok and bounded
                assert bool(q["epochs"]) == bool(epochs)
            if name == "fm":
                assert bounded
            if name in ("am", "ssb"):
                assert vlen // 32 >= 128
            if q["grouped"]:
                starts = [q["gs0"], q["gs1"], q["gs2"], q["gs3"]]
                nseg = [q["gn0"], q["gn1"], q["gn2"]]
                assert starts[0] == 0 and all(s % 16 == 0 for s in starts)
                assert nseg == [c * q["tiles_per_ch"] for c in q["rot_count"]]
                assert all(starts[r + 1] - starts[r] == -(-nseg[r] // 16) * 16 for r in range(3))
                ids = starts[3]
            else:
                assert name == "wbfm" and sum(1 for c in q["rot_count"] if c) == 1
                ids = q["n"] * q["tiles_per_ch"]
            assert 1 <= q["rings"] <= 3 and (q["rings"] == 3 or ((q["rounds"] == 1 or rings_knob) and not p["fused"]))   # fewer rings: launches of one round
            wg_segs = 64 * q["rings"]
            assert wg_segs == st_segs or q["rings"] < 3
            assert q["grid"] * q["rounds"] * wg_segs >= ids                                  # every segment id has a workgroup and a round
            assert (q["grid"] - 1) * q["rounds"] * wg_segs < ids or q["rounds"] > 1 or q["grid"] == 1   # and no workgroup is launched for nothing
        else:
            if name == "wbfm":
                assert not q["epochs"]
        if flags & F_TILES or env_path < 0:
            assert q["path"] == PLAN_TILES
        if p["fused"]:
            assert q["path"] == PLAN_STREAM                  # a fused plan in which a family fell off its pipeline cannot exist
            ranges.append((q["wg_first"], q["grid"]))
            if name == "wbfm":
                assert not q["grouped"] and not q["epochs"]
            if name in ("am", "ssb"):
                assert vlen // 32 <= dc_tile
    if p["fused"]:
        ranges.sort()
        for first, count in ranges:
            assert first == at and count >= 1
            at += count
        assert at == p["mix_wgs"] <= n_cus                   # contiguous, disjoint, every workgroup a CU of its own
    if not p["forked"]:
        assert not p["shares_on"] and not p["fused"]


def test_plan_call_over_everything_the_fuzzers_draw():
    """Every family subset x row length x pin (flags, IQD_WBFM_PATH, IQD_MIXED, IQD_SHARES, IQD_STREAM_MIN_SEG) x gain / epoch
    condition, with channel counts from one to thousands: the plan is whole (every family has a path, its segments cover the
    row, a streaming grid covers its ids), a fused plan holds streaming families only - so the launch loop's old guard has
    nothing left to guard - and the pins pin."""
    import itertools
    rng = np.random.default_rng(17)
    n_plans = n_fused = n_stream = 0
    vlens = [32, 96, 128, 1024, 2048, 4096, 8192, 16384, 65536, 1 << 18, 1 << 22]
    for subset in range(1, 16):
        names = [FAMS[f] for f in range(4) if subset >> f & 1]
        for vlen, (flags, env_path), (mixed_forked, by_cost, min_seg) in itertools.product(
                vlens, [(0, 0), (F_TILES, 0), (F_STREAM, 0), (0, 1), (0, -1)], [(False, False, 0), (True, False, 0), (False, True, 0), (False, False, 1)]):
            for scale in (1, 24, 700, 4096):
                fams = {}
                for name in names:
                    n = max(1, int(rng.integers(max(1, scale // 2), scale + 1)))
                    a = int(rng.integers(0, n + 1)); b = int(rng.integers(0, n - a + 1))
                    rc = (n, 0, 0) if rng.random() < 0.5 else (a, b, n - a - b)
                    fams[name] = (rc, rng.random() > 0.1, name == "wbfm" and rng.random() < 0.15)
                stream_ok = rng.random() > 0.05
                p = plan_call(vlen, fams, flags=flags, env_path=env_path, mixed_forked=mixed_forked, by_cost=by_cost, min_seg=min_seg, stream_ok=stream_ok)
                check_plan(p, vlen, flags, env_path, stream_ok, 256, fams)
                assert p == plan_call(vlen, fams, flags=flags, env_path=env_path, mixed_forked=mixed_forked, by_cost=by_cost, min_seg=min_seg,
                                      stream_ok=stream_ok)   # a pure function of its inputs (the engine keeps the plan while the shape repeats)
                n_plans += 1
                n_fused += p["fused"]
                n_stream += any(q["path"] == PLAN_STREAM for q in p["fam"].values())
                if mixed_forked:
                    assert not p["fused"]
                if flags & F_STREAM and vlen % 128 == 0 and vlen >= 4096:
                    assert all(q["path"] == PLAN_STREAM for n_, q in p["fam"].items() if q["n"] and (n_ != "wbfm" or (stream_ok and fams[n_][1]))
                               and (n_ != "fm" or fams[n_][1])), (vlen, fams, p)
    assert n_plans > 10000 and n_fused > 150 and n_stream > 3000, (n_plans, n_fused, n_stream)


def test_plan_call_known_configurations():
    # BASELINE configs[1]: one WBFM row of 2^28 samples - one streaming launch over the whole chip, 5632-sample segments
    p = plan_call(1 << 28, {"wbfm": (1, True, False)})
    q = p["fam"]["wbfm"]
    assert (q["path"], q["tile_len"], q["grid"], q["rounds"], q["grouped"]) == (PLAN_STREAM, 5632, 249, 1, 0) and not p["forked"]
    # configs[2]: 4096 FM channels x 2^16
    q = plan_call(1 << 16, {"fm": (4096, True, False)})["fam"]["fm"]
    assert q["path"] == PLAN_STREAM and q["rounds"] == 1 and q["grid"] <= 256 and q["rings"] == 3
    # the reference's operating point with a thousand channels (one 64 ms block per channel and call): workgroups of ONE ring on
    # every CU instead of 118 full ones - longer segments, faster pieces (tools/rings_probe.sh: AM 0.076 -> 0.064 ms)
    q = plan_call(1 << 14, {"am": (1024, True, False)}, full_lead_ins=True)["fam"]["am"]
    assert (q["path"], q["rings"], q["tile_len"], q["grid"], q["rounds"], q["lead_shift"]) == (PLAN_STREAM, 1, 1024, 256, 1, 0), q
    # ... and with round 6's short lead-ins (128 samples; a segment takes its predecessor's state from the lane below)
    q = plan_call(1 << 14, {"am": (1024, True, False)})["fam"]["am"]
    assert (q["path"], q["rounds"], q["halo"], q["lead_shift"]) == (PLAN_STREAM, 1, 128, 256) and q["tile_len"] * q["tiles_per_ch"] >= (1 << 14) + 2 * 256, q
    # configs[2] / USB at 4096 x 2^16: twelve segments per channel, 5632 + 128 / 5760 + 128 samples each (room for two cold segments
    # per channel) where they were 5504 + 768 / 5504 + 1280
    q = plan_call(1 << 16, {"fm": (4096, True, False)})["fam"]["fm"]
    assert (q["tile_len"], q["tiles_per_ch"], q["halo"], q["lead_shift"]) == (5632, 12, 128, 640), q
    q = plan_call(1 << 16, {"ssb": (4096, True, False)})["fam"]["ssb"]
    assert (q["tile_len"], q["tiles_per_ch"], q["halo"], q["lead_shift"]) == (5760, 12, 128, 1152), q
    q = plan_call(1 << 16, {"am": (4096, True, False)})["fam"]["am"]
    assert (q["tile_len"], q["tiles_per_ch"], q["halo"], q["lead_shift"]) == (5504, 12, 128, 256), q
    # the reference's operating point, 4096 channels x one 64 ms block: FM 1536 + 128 samples per segment where it was 1408 + 768
    q = plan_call(1 << 14, {"fm": (4096, True, False)})["fam"]["fm"]
    assert (q["path"], q["tile_len"], q["tiles_per_ch"]) == (PLAN_STREAM, 1536, 12), q
    q = plan_call(1 << 14, {"am": (1024, True, False)}, rings=3)["fam"]["am"]          # (IQD_RINGS pins it)
    assert (q["rings"], q["tile_len"]) == (3, 768) and q["grid"] < 128
    # configs[3]: the four families as ranges of one launch
    p = plan_call(1 << 16, {"am": (819, True, False), "fm": (819, True, False), "wbfm": (820, True, False), "ssb": (1638, True, False)})
    assert p["fused"] and p["mix_wgs"] <= 256 and all(q["path"] == PLAN_STREAM and q["rounds"] == 1 for q in p["fam"].values())
    # the same with a WBFM gain change in reach of a lead-in, or WBFM channels of two selectors: kernels of their own, still streaming
    for wb in ((820, True, True), ((400, 420, 0), True, False)):
        p = plan_call(1 << 16, {"am": (819, True, False), "fm": (819, True, False), "wbfm": wb, "ssb": (1638, True, False)})
        assert not p["fused"] and p["shares_on"] and p["fam"]["wbfm"]["path"] == PLAN_STREAM
        assert len({q["lane"] for q in p["fam"].values()}) >= 3          # side by side on the side streams
    # a WBFM gain so large that (int16)y can hit the integer-indefinite value: that family on the tile kernel, and no share for anyone
    p = plan_call(1 << 16, {"am": (819, True, False), "wbfm": (820, False, False)})
    assert not p["fused"] and p["fam"]["wbfm"]["path"] == PLAN_TILES
    # a row that is not whole 128-sample units (one short block of 96 samples): tile kernels
    assert plan_call(96, {"wbfm": (3, True, False), "am": (2, True, False)})["fam"]["am"]["path"] == PLAN_TILES


def test_planning_a_long_row_is_cheap_now():
    """plan_fused_by_time walked vlen / 768 values of k per call (2 ms at 2^28 samples, on the submit path of a
    call whose kernels take 0.3 ms); it visits only the k that shorten the segment now."""
    import time
    fams = {"fm": (1, True, False), "wbfm": (1, True, False)}
    plan_call(1 << 28, fams)
    t0 = time.perf_counter()
    for _ in range(20):
        p = plan_call(1 << 28, fams)
    per_call = (time.perf_counter() - t0) / 20
    assert p["fused"] and per_call < 1.5e-3, per_call       # (ctypes and numpy around it included; it was 2 ms in C alone)


def test_the_taps_the_kernels_carry_as_literals_are_the_ones_the_host_quantises():
    """Round 5: the IIR lanes of the WBFM streaming kernel take their decimator taps as instruction literals (iqd_taps.h:
    STREAM_TAPS, quantised at compile time) - the same 30 words build_consts() + build_stream_taps() put into the kernel
    arguments (which the fix-up kernel and the tile kernels keep using)."""
    lib = _plan_lib()
    lit, host = np.zeros(30, np.uint32), np.zeros(30, np.uint32)
    lib.emu_stream_taps.restype = None
    lib.emu_stream_taps.argtypes = [C.c_void_p, C.c_void_p]
    lib.emu_stream_taps(lit.ctypes.data, host.ctypes.data)
    assert host.any() and np.array_equal(lit, host), (lit, host)
