"""CPU tier: the boundary fix-up of the FM / AM / SSB streaming pipelines (rtlsdrdiags_amd/csrc/iqd_d4_fix.h, round 6) stepped on
the host (tests/emu) against a plain numpy restatement of the stages behind the records.

The pipelines' segments run 128 samples of lead-in instead of 384 / 768 / 1280 and leave boundary records; the launch that
closes a step recomputes each later segment's first 4 / 18 / 34 outputs from the records of the two segments that meet there.
Here a channel's intermediate streams (y2 at 16 kS/s, for SSB the 8 kS/s rails) are random, the records are cut out of them
exactly where the consumer lanes take them (d4_am_wave / d4_fm_wave: pieces 4.. and the run's last pieces), every output the
fix-up is responsible for is poisoned, and after the fix-up the whole row must be the direct computation's - for segment
lengths, shifts and row lengths as the plan produces them, short last segments and more boundaries than one batch included.
Since the pipelines fix the boundaries inside a consumer wave themselves (the lane below holds the predecessor's end state),
the closing launch is left with every 64th segment id: the odd trials poison and fix only the boundaries t_first, t_first + 64, ...
"""
import ctypes as C
import os

import numpy as np
import pytest

from tests import emu_bind

FAM_AM, FAM_FM, FAM_SSB = 0, 1, 3


@pytest.fixture(scope="module")
def L():
    lib = emu_bind.lib()
    lib.emu_d4_fix.restype = None
    lib.emu_d4_fix.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32]
    lib.emu_d4_const.restype = C.c_uint32
    lib.emu_d4_const.argtypes = [C.c_int]
    lib.emu_taps.restype = C.c_int
    lib.emu_taps.argtypes = [C.c_int, C.c_void_p]
    return lib


def taps(L, which):
    buf = np.zeros(40, np.int16)
    n = L.emu_taps(which, buf.ctypes.data)
    return buf[:n].astype(np.int64)


def q15(h, x, newest, clamp=False):
    """Decimator_int16.cc:176-238 at the positions `newest` (array of indices into x): acc = 2^14; acc += h[k] x[n-k] (clamped per MAC); >> 15."""
    acc = np.full(len(newest), 1 << 14, np.int64)
    for k in range(len(h)):
        acc = acc + h[k] * x[newest - k]
        if clamp:
            acc = np.clip(acc, -(1 << 30), (1 << 30) - 1)
    return acc >> 15


def geometry(tile_len, shift, vlen):
    n_tiles = (vlen + shift + tile_len - 1) // tile_len
    return n_tiles, (128 + tile_len) // 32


def pairs(x):
    """int16 stream -> uint32 pairs, older sample in the low half"""
    x = x.astype(np.int64) & 0xffff
    return (x[0::2] | (x[1::2] << 16)).astype(np.uint32)


PAD = 64      # pieces of history in front of the call's first sample (the first segment's lead-in lies there)


def run_piece0(t, tile_len, shift):
    """absolute piece index (piece 0 = the call's samples 0..31) of segment t's run piece 0"""
    return (t * tile_len - shift - 128) // 32


@pytest.mark.parametrize("family", [FAM_AM, FAM_SSB, FAM_FM])
def test_fixup_restores_every_output_the_short_lead_in_leaves_open(L, family):
    rng = np.random.default_rng(60 + family)
    shift = L.emu_d4_const({FAM_AM: 4, FAM_FM: 5, FAM_SSB: 6}[family])
    assert L.emu_d4_const(3) == 128 and shift == {FAM_AM: 256, FAM_FM: 640, FAM_SSB: 1152}[family]
    rec_bytes = L.emu_d4_const({FAM_AM: 0, FAM_FM: 1, FAM_SSB: 2}[family])
    n_fix = {FAM_AM: 4, FAM_FM: 18, FAM_SSB: 34}[family]
    cases = 0
    for trial in range(60):
        tile_len = 128 * int(rng.integers((shift + 128) // 128, 64))
        vlen = 128 * int(rng.integers(1, 400))
        if trial % 7 == 0:
            vlen = tile_len * int(rng.integers(13, 30)) - shift       # whole segments, more boundaries than one batch
        if trial % 11 == 5:
            tile_len = shift + 128
            vlen = tile_len * int(rng.integers(80, 200)) - shift      # hundreds of segments: the every-64th stepping over several batches
        n_tiles, N = geometry(tile_len, shift, vlen)
        if n_tiles < 2:
            continue
        cases += 1
        n_pieces = vlen // 32                                          # output samples of the row
        total = PAD + n_pieces + N + 64                                # pieces, with room behind the row for a last segment's run
        if family == FAM_FM:
            loud = trial % 3 == 0                                       # full-scale values: the per-MAC clamp fires
            y2 = rng.integers(-32768 if loud else -12000, 32768 if loud else 12000, 2 * total).astype(np.int64)
            h40 = taps(L, 5)
            newest = 2 * (PAD + np.arange(n_pieces)) + 1
            truth = q15(h40, y2, newest, clamp=True).astype(np.int16)
            rec = np.zeros((n_tiles, rec_bytes // 4), np.uint32)
            for t in range(n_tiles):
                p0 = PAD + run_piece0(t, tile_len, shift)
                rec[t, 0:20] = pairs(y2[2 * (p0 + 4): 2 * (p0 + 24)])
                rec[t, 20:40] = pairs(y2[2 * (p0 + N - 20): 2 * (p0 + N)])
            out = truth.copy()
        else:
            y2 = rng.integers(-121, 122, (2, 2 * total)).astype(np.int64)
            h16 = taps(L, 1)
            newest_all = 2 * np.arange(8, total) + 1
            rails = np.zeros((2, total), np.int64)
            for r in range(2):
                rails[r, 8:] = q15(h16, y2[r], newest_all)
            at = PAD + np.arange(n_pieces)
            lsb = int(trial & 1)
            if family == FAM_AM:
                im, qm = np.abs(rails[0, at]), np.abs(rails[1, at])
                truth = np.where(im > qm, im + (qm >> 1), qm + (im >> 1)).astype(np.int32)
            else:
                idl = q15(taps(L, 2), rails[0], at)
                qh = q15(taps(L, 3), rails[1], at)
                truth = (idl - qh if lsb else idl + qh).astype(np.int32)
            rec = np.zeros((n_tiles, rec_bytes // 4), np.uint32)
            for t in range(n_tiles):
                p0 = PAD + run_piece0(t, tile_len, shift)
                for r in range(2):
                    rec[t, 4 * r: 4 * r + 4] = pairs(y2[r, 2 * (p0 + 4): 2 * (p0 + 8)])
                    rec[t, 8 + 8 * r: 8 + 8 * r + 7] = pairs(y2[r, 2 * (p0 + N - 7): 2 * (p0 + N)])
                    if family == FAM_SSB:
                        rec[t, 24 + 16 * r: 24 + 16 * r + 16] = pairs(rails[r, p0 + 8: p0 + 40])
                        rec[t, 56 + 16 * r: 56 + 16 * r + 16] = pairs(rails[r, p0 + N - 32: p0 + N])
            out = truth.copy()
        # poison what the fix-up must write: the first n_fix outputs of every segment but the first (as far as the segment goes) -
        # or, as the closing launch is asked, of the segments t_first, t_first + 64, ...
        t_first, t_step = (1, 1) if trial % 2 == 0 or n_tiles < 4 else (int(rng.integers(1, min(n_tiles, 65))), 64 if n_tiles > 70 else 3)
        for t in range(t_first, n_tiles, t_step):
            v0 = t * tile_len - shift
            tlen = min(tile_len, vlen - v0)
            out[v0 // 32: v0 // 32 + min(n_fix, tlen // 32)] = 0x5a5a if family != FAM_FM else 0x5a5a
        assert not np.array_equal(out, truth)
        L.emu_d4_fix(family, rec.ctypes.data, n_tiles, tile_len, shift, vlen, lsb if family != FAM_FM else 0, out.ctypes.data, t_first, t_step)
        bad = np.flatnonzero(out != truth)
        assert bad.size == 0, (family, trial, tile_len, vlen, n_tiles, bad[:8], out[bad[:8]], truth[bad[:8]])
    assert cases > 40
