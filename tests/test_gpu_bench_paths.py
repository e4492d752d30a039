"""GPU (MI355X): exactly the paths `bench.py` times, at their sizes, with EVERY channel held to the oracle:
--config 2 (4096 FM channels x 2^16), --mode am and --mode usb at 4096 x 2^16 - device pointers, one whole-chip streaming
launch per call, the bench's own per-channel data (bench.per_channel_rows) - and --config 3 / --config 4 with all
channels compared instead of a sample.  The oracle runs in a process pool over the host's cores (tests/oracle_pool.py)."""
import numpy as np
import pytest

from rtlsdrdiags_amd import synth
from oracle_pool import bench_rows, oracle_all_channels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from rtlsdrdiags_amd import capi as c
    return c


def _run_on_device(capi, rows, modes, rots=None, calls=2, threshold=None, agc=None, expect_streams=None, expect_mixed=None,
                   expect_launches=None):
    n_ch, nbytes = rows.shape
    nblk = nbytes // 32768
    eng = capi.Engine(n_ch)
    for arr, setter in ((modes, eng.set_mode), (rots or [1] * n_ch, eng.set_rotation)):
        c0 = 0
        for c in range(1, n_ch + 1):
            if c == n_ch or arr[c] != arr[c0]:
                setter(arr[c0], first=c0, n=c - c0)
                c0 = c
    if threshold is not None:
        eng.set_squelch(threshold)
    if agc is not None:
        eng.agc_set_type(agc)
        eng.agc_enable(True)
    iq_d, pcm_d = eng.dev_alloc(rows.nbytes), eng.dev_alloc(n_ch * (nbytes // 64) * 2)
    cnt_d, mag_d, al_d = eng.dev_alloc(n_ch * 4), eng.dev_alloc(n_ch * nblk * 4), eng.dev_alloc(n_ch * nblk)
    eng.dev_upload(iq_d, rows)
    got = []
    for call in range(calls):
        before = eng.stats()
        eng.accept_device(iq_d, nbytes, pcm_d, cnt_d, mag_d, al_d)
        eng.synchronize()
        after = eng.stats()
        if expect_streams is not None:
            assert after["stream_launches"] - before["stream_launches"] == expect_streams, (call, after)
        if expect_mixed is not None:
            assert after["mixed_launches"] - before["mixed_launches"] == expect_mixed, (call, after)
        if expect_launches is not None:
            assert after["device_launches"] - before["device_launches"] == expect_launches, (call, before, after)
        got.append((eng.dev_download(pcm_d, n_ch * (nbytes // 64) * 2, np.int16).reshape(n_ch, -1),
                    eng.dev_download(cnt_d, n_ch * 4, np.uint32),
                    eng.dev_download(mag_d, n_ch * nblk * 4, np.uint32).reshape(n_ch, nblk),
                    eng.dev_download(al_d, n_ch * nblk, np.uint8).reshape(n_ch, nblk)))
    assert eng.stats()["state_repairs"] == 0
    for p_ in (iq_d, pcm_d, cnt_d, mag_d, al_d):
        eng.dev_free(p_)
    eng.close()
    return got


def _compare_all(got, rows, modes, rots=None, threshold=None, agc=None):
    calls = len(got)
    pcm, mag, allowed = oracle_all_channels(rows, modes, rots, calls=calls, threshold=threshold, agc=agc)
    for k, (g_pcm, g_cnt, g_mag, g_al) in enumerate(got):
        for c in range(len(rows)):
            assert g_cnt[c] == len(pcm[k][c]), (k, c, modes[c])
            assert np.array_equal(g_pcm[c, :g_cnt[c]], pcm[k][c]), (k, c, modes[c])
            assert np.array_equal(g_mag[c], mag[k][c]) and np.array_equal(g_al[c], allowed[k][c]), (k, c)


@pytest.mark.parametrize("mode", ["fm", "am", "usb"])
def test_bench_single_family_paths_at_size_every_channel(capi, mode):
    """`bench.py --config 2` (FM) / `--mode am` / `--mode usb`, 4096 channels x 2^16 samples, two calls: one streaming
    launch per call, every PCM sample, magnitude and flag of all 4096 channels against the oracle."""
    n_ch, n = 4096, 1 << 16
    rows = bench_rows(synth.fm_tone(n, seed=1234), n_ch, 0, 2 * n)
    modes = [mode] * n_ch
    got = _run_on_device(capi, rows, modes, expect_streams=1, expect_mixed=0)
    _compare_all(got, rows, modes)


@pytest.mark.parametrize("n_ch,blocks", [(1, 1), (3, 1), (700, 2), (64, 4), (257, 3)])
def test_small_fm_calls_on_the_tile_kernels(capi, n_ch, blocks):
    """configs[0] (one FM channel, one 32 768-byte block per call - the reference's own operating point) and its neighbours on the
    tile kernels: three consecutive calls give the oracle's PCM, magnitudes, flags and counts for every channel (the second and
    third calls read the tails and the zeroed sums the first one's closing launch left)."""
    n = blocks * 16384
    rows = bench_rows(synth.fm_tone(n, seed=77 + n_ch), n_ch, 0, 2 * n)
    modes = ["fm"] * n_ch
    got = _run_on_device(capi, rows, modes, calls=3, expect_streams=0, expect_launches=2)
    _compare_all(got, rows, modes)


# What ranks 0, 1 and 7 of an 8-GPU job run: first_global = rank * channels per GPU, so the mode mix
# starts at another residue of g % 5 (4096 % 5 = 1), the rotation selectors at another residue of g % 3 (8192 % 3 = 2), the
# gating classes and the data rolls at other offsets.  One GPU can run each rank's slice.
@pytest.mark.parametrize("rank", [0, 1, 7])
def test_bench_config3_mixed_at_size_every_channel(capi, rank):
    """`bench.py --config 3`: 4096 channels x 2^16 per GPU, job-wide channel g % 5 -> {AM, FM, WBFM, LSB, USB}, the bench's
    per-channel data: the four families' pipelines as ranges of one launch, all 4096 channels against the oracle, two calls."""
    import bench
    n_ch, n = 4096, 1 << 16
    first_global = rank * n_ch
    rows = bench_rows(synth.fm_tone(n, seed=1234), n_ch, first_global, 2 * n)
    modes, rots = bench.channel_plan("mixed", n_ch, first_global)
    assert modes[0] == ["am", "fm", "wbfm", "lsb", "usb"][first_global % 5]
    got = _run_on_device(capi, rows, modes, rots, expect_streams=4, expect_mixed=1)
    _compare_all(got, rows, modes, rots)


@pytest.mark.parametrize("rank", [0, 1, 7])
def test_bench_config4_gated_ssb_at_size_every_channel(capi, rank):
    """`bench.py --config 4` on one GPU of eight: 8192 LSB / USB channels x 2^16, rotation selector by job-wide channel,
    Harris AGC, squelch at -60 dBFS over the bench's loud / quiet block classes (a quarter of the blocks rejected in the
    steady state): the gated streaming pipeline, three calls (the AGC moves), every channel against the oracle."""
    import bench
    from oracle_pool import gated_rows
    n_ch, n = 8192, 1 << 16
    first_global = rank * n_ch
    rows = gated_rows(synth.fm_tone(n, seed=1234), synth.fm_tone(n, seed=1234, amplitude=2.0, sigma=1.0), n_ch, first_global,
                      2 * n, bench.GATE_PATTERNS)
    modes, rots = bench.channel_plan("ssb_stress", n_ch, first_global)
    assert rots[0] == (1, 0, -1)[first_global % 3]
    got = _run_on_device(capi, rows, modes, rots, calls=3, threshold=bench.GATE_THRESHOLD_DBFS, agc=1, expect_streams=1)
    rejected = 1.0 - float(np.mean(got[-1][3]))
    assert 0.15 < rejected < 0.35, rejected
    _compare_all(got, rows, modes, rots, threshold=bench.GATE_THRESHOLD_DBFS, agc=1)
