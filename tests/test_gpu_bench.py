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
