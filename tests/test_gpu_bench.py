"""GPU (MI355X): bench.py itself, as the driver calls it - one JSON line on stdout with the contract's fields, figures that
agree with each other, the streaming kernels named as what ran.  Small step counts; the timed numbers themselves are not
asserted (boxes differ), their consistency is."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def check_contract(d, steps, warmup):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["unit"] == "MSamples/s" and d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    return d


def test_default_workload_line_small():
    """configs[1] shape at 2^26 samples: WBFM streaming kernel + fix-up timed by HIP events, algorithmic bytes = 2.0625 B x samples."""
    d = check_contract(run_bench("--steps", "4", "--warmup", "1", "--prewarm-ms", "20", "--log2-samples", "26", "--no-cpu-baseline",
                                 "--no-host-path", "--no-live-pmc"), 4, 1)
    r = d["roofline"]
    n = 1 << 26
    assert r["algorithmic_bytes_per_launch"] == 2.0625 * n and "wbfm_stream_kernel" in r["kernel"]
    assert 0 < r["kernel_ms"] <= d["ms_per_step"] * 1.02                      # the timed kernels are part of the step
    assert abs(r["achieved"] - 2.0625 * n / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 2e-3
    assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 2e-3
    assert d["config"]["kernels"] == "streaming pipelines" and d["state_repairs"] == 0 and "untimed steps" in d["config"]["prewarm"]


def test_mixed_configuration_line():
    """configs[3]: the one-launch arrangement is what ran, and the line says so."""
    d = check_contract(run_bench("--config", "3", "--steps", "4", "--warmup", "1", "--prewarm-ms", "20", "--no-cpu-baseline",
                                 "--no-host-path", "--no-live-pmc"), 4, 1)
    assert "one launch" in d["config"]["kernels"] and "mixed_stream_kernel" in d["roofline"]["kernel"]
    assert 0 < d["roofline"]["kernel_ms"] <= d["ms_per_step"] * 1.02
    assert d["config"]["squelch"]["blocks_rejected_frac"] == 0.0


def test_cpu_baseline_and_live_counters_present():
    """The full default form on a small workload: cpu_baseline (the reference or its port, on this box's host cores) and the
    counters collected by the run's own rocprofv3 passes (or a reason why not)."""
    d = check_contract(run_bench("--steps", "3", "--warmup", "1", "--prewarm-ms", "10", "--log2-samples", "24", "--no-host-path"), 3, 1)
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "MSamples/s" and cb["sample"]
    r = d["roofline"]
    assert "traffic" in r
    if r["traffic"] is not None:                                              # rocprofv3 present: HBM bytes per launch, corrected
        assert 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 3.0 and "rocprofv3" in r["counters"]


def check_sub_line(s):
    """One `other_configs` entry: the same internal consistency the headline is held to."""
    assert "error" not in s, s
    for key in ("argv", "metric", "value", "unit", "ms_per_step", "config", "roofline"):
        assert key in s, (key, s.get("argv"))
    r = s["roofline"]
    assert s["unit"] == "MSamples/s" and s["value"] > 0 and s["ms_per_step"] > 0 and "workload" in s["config"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    ch, log2 = s["config"]["channels_per_gpu"], s["config"].get("log2_samples_per_channel")
    if log2 is None:                              # configs[0]: one 32768-byte block per step, the CPU chain beside it
        assert s["argv"] == "--config 0" and s["parity"] == "PCM identical to the CPU chain" and s["block_latency_us"]["p50"] > 0
        assert abs(s["value"] - 16384 / (s["ms_per_step"] * 1e3)) / s["value"] < 2e-3
        return
    n = ch << log2
    assert abs(s["value"] - n / (s["ms_per_step"] * 1e-3) / 1e6) / s["value"] < 2e-3
    assert r["algorithmic_bytes_per_launch"] == 2.0625 * n
    assert 0 < r["kernel_ms"] <= s["ms_per_step"] * 1.02                      # the timed kernels are part of the step
    assert r["achieved"] <= 2.0625 * n / (r["kernel_ms"] * 1e-3) / 1e9 * 1.002  # never more than the timed kernels alone would give
    assert "streaming" in s["config"]["kernels"] or "one launch" in s["config"]["kernels"]
    assert s["state_repairs"] == 0
    if r.get("traffic") is not None:
        assert 0.8 < r["traffic_over_algorithmic"] < 3.0 and abs(r["traffic_over_algorithmic"] - r["traffic"] / r["algorithmic_bytes_per_launch"]) < 2e-3


def test_default_run_carries_every_other_configuration():
    """`python bench.py --gpus 1 --steps K --warmup W`, the driver's own command: the headline line of configs[1] with its usual
    fields, and `other_configs` - configs[0], [2], [3], [4] and AM / USB at 4096 x 2^16, a few steps each, a child process each
     - each sub-line consistent in itself.  (Full size: this IS the driver's run, about three minutes.)"""
    import bench
    d = check_contract(run_bench("--gpus", "1", "--steps", "6", "--warmup", "2", "--cpu-one-core-only", "--no-live-pmc"), 6, 2)
    assert "wbfm_stream_kernel" in d["roofline"]["kernel"] and d["config"]["log2_samples_per_channel"] == 28
    subs = d["other_configs"]
    assert [s["argv"] for s in subs] == [" ".join(a) for a in bench.OTHER_CONFIGS]
    for s in subs:
        check_sub_line(s)
    by = {s["argv"]: s for s in subs}
    assert "mixed_stream_kernel" in by["--config 3"]["roofline"]["kernel"]
    assert by["--config 4"]["config"]["squelch"]["blocks_rejected_frac"] > 0.2


# ---- the N > 1 code path on the hardware there is: one rank under a real launcher ----------------------
ONE_RANK_ARGS = ["--config", "3", "--gather", "--steps", "2", "--warmup", "1", "--prewarm-ms", "10", "--no-cpu-baseline", "--no-host-path",
                 "--no-live-pmc"]


def check_one_rank_line(d):
    check_contract(d, 2, 1)
    assert len(d["per_rank_ms"]) == 1 and d["per_rank_ms"][0] > 0
    r = d["rccl"]                                   # what the collective library itself reports
    assert r["backend"] == "nccl" and r["ranks"] == 1 and int(r["version"]) >= 20000, r
    assert "iqd_gather" in r["communicator"] and "iqd_gather_pcm" in d["config"]["sharding"]
    assert "other_configs" not in d and "mixed_stream_kernel" in d["roofline"]["kernel"]


def test_one_rank_under_torch_distributed_run():
    """The driver's multi-GPU command line with one process: `python -m torch.distributed.run --nproc-per-node 1 bench.py --gather` -
    the real nccl (= RCCL) backend with device_id=, torch's barriers and all_gather on the device, the engine's own gatherer
    (unique id broadcast, iqd_gather_create, grouped transfers on the engine's stream) and the `rccl` block of the line."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1"] + ONE_RANK_ARGS,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    check_one_rank_line(json.loads(lines[0]))


def test_bench_starts_its_own_single_rank():
    """The same through bench.py's own launcher (`self_launch`: torch.distributed.run as a CHILD process, never an exec), forced
    for one GPU: what `python bench.py --gpus 8` does on a node, at the world size this box has."""
    check_one_rank_line(run_bench("--gpus", "1", "--force-launch", *ONE_RANK_ARGS))
