"""GPU (MI355X): a fixed-seed slice of tools/gpu_fuzz.py - random channel counts, modes, gains, thresholds, AGC and
scanner settings, block sizes, signal kinds, call boundaries, channel sub-ranges and operator commands between calls
- every case compared with the oracle (PCM, decisions, magnitudes, IF gain, tuned frequency)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("seed", [9001, 9002])
def test_random_configurations_match_the_oracle(oracle, seed):
    import gpu_fuzz
    rng = np.random.default_rng(seed)
    for case in range(200):
        assert gpu_fuzz.one_case(rng, oracle), (seed, case)


def test_random_long_rows_match_the_oracle(oracle, monkeypatch):
    import gpu_fuzz
    monkeypatch.setenv("FUZZ_BIG", "1")
    rng = np.random.default_rng(9003)
    for case in range(40):
        assert gpu_fuzz.one_case(rng, oracle), case


def test_random_wide_batches_match_the_oracle(oracle):
    """Hundreds to thousands of channels per launch from a few templates: every row of every case against the oracle
    (nothing randomised covered the many-channel launch geometry)."""
    import gpu_fuzz
    rng = np.random.default_rng(9004)
    for case in range(10):
        assert gpu_fuzz.wide_case(rng, oracle), case


def test_random_short_block_calls_match_the_oracle(oracle):
    """Every call one short block of any multiple of 64 bytes (or a few whole blocks), WBFM included, operator changes in
    between (tools/gpu_fuzz.py: short_case)."""
    import gpu_fuzz
    rng = np.random.default_rng(9005)
    for case in range(60):
        bad = gpu_fuzz.short_case(rng, oracle)
        assert bad is None, (case, bad)
