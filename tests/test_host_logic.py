"""CPU: host-side logic of the engine and the C-ABI surface (no compute without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wrap_delta_float_only():
    """iqd_prims.h::wrap_delta == the reference's double-precision branch-cut handling
    (WbFmDemodulator.cc:472-480) for EVERY float in [pi, 2*pi] and its mirror image."""
    f32 = np.float32
    pi_f, hi = f32(np.pi), f32(2 * np.pi)
    lo = f32(2 * np.pi - float(hi))
    assert float(hi) / 2 == float(pi_f) and float(pi_f) > np.pi
    bits = np.arange(np.array([pi_f]).view(np.uint32)[0] - 8, np.array([hi]).view(np.uint32)[0] + 1, dtype=np.uint32)
    d = bits.view(np.float32)
    for sign in (1.0, -1.0):
        x = (d * f32(sign)).astype(np.float32)
        xd = x.astype(np.float64)
        ref = x.copy()
        m = xd > np.pi
        ref[m] = (xd[m] - 2 * np.pi).astype(np.float32)
        m = xd < -np.pi
        ref[m] = (xd[m] + 2 * np.pi).astype(np.float32)
        assert not np.any(np.abs(ref.astype(np.float64)) > np.pi)          # one step is enough
        mine = np.where(x >= pi_f, ((x - hi).astype(np.float32) - lo).astype(np.float32),
                        np.where(x <= -pi_f, ((x + hi).astype(np.float32) + lo).astype(np.float32), x))
        assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32))
        # the kernels pick the branch with k = rint(d * c) (one v_mul_f32 + one v_rndne_f32): it must be
        # sign(d) exactly where |d| >= pi_f and 0 below, over this whole interval ...
        c = f32(0.15915495157241821)
        k = np.rint((x * c).astype(np.float32))
        assert np.array_equal(k, np.where(np.abs(x) >= pi_f, f32(sign), f32(0.0)))
    # ... and, the product being monotone in d, everywhere in [0, 2*pi]: zero up to the float below pi_f, one from pi_f on
    c = f32(0.15915495157241821)
    below = np.nextafter(pi_f, f32(0))
    assert f32(below * c) == f32(0.5) and np.rint(f32(below * c)) == 0      # the tie goes to the even 0
    assert np.rint(f32(pi_f * c)) == 1 and np.rint(f32(hi * c)) == 1 and f32(hi * c) < 1.5
    lowbits = np.arange(0, np.array([pi_f]).view(np.uint32)[0], 4099, dtype=np.uint32).view(np.float32)
    assert not np.rint((lowbits * c).astype(np.float32)).any()


def test_library_exports_every_declared_symbol():
    from rtlsdrdiags_amd import capi
    header = open(os.path.join(ROOT, "include", "iqdemod.h")).read()
    declared = set(re.findall(r"\b(iqd_[a-z_0-9]+)\s*\(", header))
    declared -= {"iqd_t"}
    lib = C.CDLL(capi.LIB)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(capi.EXPORTS)


def test_create_fails_loudly_without_a_gpu_or_with_bad_config():
    import torch
    from rtlsdrdiags_amd import capi
    lib = capi._lib()
    h = C.c_void_p()
    cfg = capi.Config(lib.iqd_abi_version(), 0, 0, -1, 0)
    assert lib.iqd_create(C.byref(cfg), C.byref(h)) == -1            # n_channels == 0
    cfg = capi.Config(lib.iqd_abi_version() + 1, 1, 0, -1, 0)
    assert lib.iqd_create(C.byref(cfg), C.byref(h)) == -1            # ABI mismatch
    cfg = capi.Config(lib.iqd_abi_version(), 1, 1000, -1, 0)
    assert lib.iqd_create(C.byref(cfg), C.byref(h)) == -1            # block_bytes % 256
    if not torch.cuda.is_available():
        with pytest.raises(capi.IqdError):
            capi.Engine(1)                                           # no CPU fallback exists
    assert lib.iqd_strerror(-2) == b"no usable HIP device"


def test_product_never_touches_the_oracle():
    """The product tree must not import, link or load anything under oracle/ or tests/."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rtlsdrdiags_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", ".cc")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                if re.search(r"(from|import)\s+oracle|oracle/|libiqd_oracle|libiqd_ref|libiqd_emu|tests/emu/", text) \
                        and "tests/emu" not in text.split("IQD_HOST_EMU")[0][-400:]:
                    hits = [l for l in text.splitlines()
                            if re.search(r"(from|import)\s+oracle|libiqd_oracle|libiqd_ref|libiqd_emu|#include.*oracle", l)]
                    if hits:
                        bad.append((f, hits))
    assert not bad, bad


def test_file_tool_fails_loudly_without_gpu():
    import subprocess
    import torch
    tool = os.path.join(ROOT, "rtlsdrdiags_amd", "bin", "iqdemod_file")
    assert os.path.exists(tool)
    if not torch.cuda.is_available():
        r = subprocess.run([tool, "3"], input=b"\x80" * 32768, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 1 and b"no usable HIP device" in r.stderr and r.stdout == b""


def test_many_channel_tool_fails_loudly_without_gpu(tmp_path):
    import subprocess
    import torch
    tool = os.path.join(ROOT, "rtlsdrdiags_amd", "bin", "iqdemod_multi")
    assert os.path.exists(tool)
    r = subprocess.run([tool, "nonsense"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"usage:" in r.stderr
    if not torch.cuda.is_available():
        (tmp_path / "cap_0.iq").write_bytes(b"\x80" * 32768)
        r = subprocess.run([tool, "channels=1", "in=" + str(tmp_path / "cap_%d.iq"), "out=" + str(tmp_path / "pcm_%d.s16")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 1 and b"no usable HIP device" in r.stderr
        assert not (tmp_path / "pcm_0.s16").exists()       # nothing is written, there is no CPU path


def test_bench_rows_numpy_twin_is_the_benchs_own_data():
    """tests/oracle_pool.bench_rows feeds the oracle what bench.per_channel_rows feeds the engine."""
    import torch
    import bench
    from oracle_pool import bench_rows
    base = np.random.default_rng(3).integers(0, 256, 5000, dtype=np.uint8)
    for first, n_ch, row_bytes in ((0, 7, 12000), (4096, 300, 5000), (11, 3, 4096)):
        a = bench.per_channel_rows(torch, torch.from_numpy(base), n_ch, first, row_bytes, chunk=64).numpy()
        assert np.array_equal(a, bench_rows(base, n_ch, first, row_bytes))
    assert not np.array_equal(a[0], a[1])


def test_gated_rows_and_channel_plan_at_the_offsets_of_other_ranks():
    """configs[3] / configs[4] as ranks 1 and 7 of an 8-GPU job stage them (first_global = rank * channels per GPU):
    tests/oracle_pool.gated_rows is bench.gated_rows, and bench.channel_plan is a function of the job-wide index only."""
    import torch
    import bench
    from oracle_pool import gated_rows
    rng = np.random.default_rng(4)
    loud, quiet = rng.integers(0, 256, 6000, dtype=np.uint8), rng.integers(0, 256, 6000, dtype=np.uint8)
    for rank in (0, 1, 7):
        first = rank * 8192
        a = bench.gated_rows(torch, torch.from_numpy(loud), torch.from_numpy(quiet), 13, first, 6 * 1024, block_bytes=1024).numpy()
        assert np.array_equal(a, gated_rows(loud, quiet, 13, first, 6 * 1024, bench.GATE_PATTERNS, block_bytes=1024))
    whole_m, whole_r = bench.channel_plan("ssb_stress", 8 * 64, 0)
    whole3, _ = bench.channel_plan("mixed", 8 * 64, 0)
    for rank in range(8):
        m, r = bench.channel_plan("ssb_stress", 64, rank * 64)
        assert m == whole_m[rank * 64:(rank + 1) * 64] and r == whole_r[rank * 64:(rank + 1) * 64]
        assert bench.channel_plan("mixed", 64, rank * 64)[0] == whole3[rank * 64:(rank + 1) * 64]
