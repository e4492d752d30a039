"""GPU (MI355X): bounded slices of tools/gpu_fuzz.py under each path-pinning knob that found a bug in round 4
(tools/fuzz_campaign.sh; VERDICT r4 item 4: the campaigns' evidence belongs where the driver runs it).  The knobs are read
by iqd_create from the environment, so every pin is a subprocess of its own: a FIXED seed for a fixed time - a regression gate:
the same cases every run.  With FUZZ_CLOCK=1 in the environment each pin then runs a second slice with a seed taken from the
clock (round 6, VERDICT r5: a green that depends on the clock is evidence, not a gate - so it is opt-in, and the tier is
two minutes shorter; the seed is printed on failure, together with the fuzzer's own description of the failing case; the
generator state of the case is left in gpurun_out/fuzz_fail_state.json for `FUZZ_REPLAY=<file> python tools/gpu_fuzz.py`).
The unbounded campaigns stay in tools/fuzz_campaign.sh (profiles/r6_fuzz_campaign.txt has this round's)."""
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (name, environment, seconds with the fixed seed, fixed seed)
PINS = [
    ("default_paths", {}, 18, 501),
    ("wbfm_stream", {"IQD_WBFM_PATH": "stream"}, 18, 502),          # found: restart record further back than the lead-in reaches
    ("wbfm_tiles", {"IQD_WBFM_PATH": "tiles"}, 16, 503),
    ("short_blocks", {"FUZZ_SHORT": "1"}, 18, 504),                 # found: DC pass past odd PCM counts, fm_shift_e
    ("stream_short_blocks", {"IQD_WBFM_PATH": "stream", "FUZZ_SHORT": "1"}, 16, 505),   # found: short last segment's restart state
    ("wide", {"FUZZ_WIDE": "1"}, 18, 506),                          # found: AM / SSB detector-stream buffer shared
    ("wide_small_calls", {"FUZZ_WIDE": "1", "FUZZ_WIDE_RANGE": "24,600"}, 16, 507),
    ("mixed_forked", {"IQD_MIXED": "forked", "FUZZ_WIDE": "1"}, 16, 508),
    ("shares_by_cost", {"IQD_SHARES": "cost", "FUZZ_WIDE": "1"}, 16, 509),
    ("min_seg_1_wide", {"IQD_STREAM_MIN_SEG": "1", "FUZZ_WIDE": "1", "FUZZ_WIDE_RANGE": "24,600"}, 16, 510),   # found: repaired keeper re-runs the short last segment
    ("min_seg_1_short", {"IQD_STREAM_MIN_SEG": "1", "FUZZ_SHORT": "1"}, 16, 511),
    # round 6: FM / AM / SSB segments with short lead-ins (iqd_stream.h: d4_geom) - by default only where a channel is cut into 8
    # segments or more; forced wherever a family streams (also for rows of a few segments and inside the one launch), and off
    ("short_lead_ins_always", {"IQD_D4_LEADFREE": "1", "FUZZ_WIDE": "1", "FUZZ_WIDE_RANGE": "24,600"}, 16, 512),
    ("short_lead_ins_min_seg_1", {"IQD_D4_LEADFREE": "1", "IQD_STREAM_MIN_SEG": "1", "FUZZ_WIDE": "1", "FUZZ_WIDE_RANGE": "24,600"}, 16, 513),
    ("short_lead_ins_one_launch", {"IQD_D4_LEADFREE": "2", "FUZZ_WIDE": "1"}, 16, 514),
    ("full_lead_ins", {"IQD_D4_LEADFREE": "0", "FUZZ_WIDE": "1"}, 16, 515),
]
CLOCK_SECONDS = 8


def _run(env_extra, seconds, seed):
    env = dict(os.environ)
    for k in ("IQD_WBFM_PATH", "IQD_MIXED", "IQD_SHARES", "IQD_STREAM_MIN_SEG", "IQD_D4_LEADFREE", "FUZZ_SHORT", "FUZZ_WIDE", "FUZZ_WIDE_RANGE",
              "FUZZ_BIG", "FUZZ_REPLAY"):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz.py"), str(seconds), str(seed)], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=seconds + 240)
    return r.returncode, r.stdout.decode(errors="replace")


@pytest.mark.parametrize("name,env_extra,seconds,seed", PINS, ids=[p[0] for p in PINS])
def test_fuzz_slice_under_pin(name, env_extra, seconds, seed):
    rc, out = _run(env_extra, seconds, seed)
    assert rc == 0 and "identical to the oracle" in out, "pin %s %r seed %d:\n%s" % (name, env_extra, seed, out[-3000:])
    cases = int(out.strip().splitlines()[-1].split("gpu_fuzz: ")[1].split()[0])
    assert cases >= 3, (name, out[-500:])     # (a slice that ran no cases proves nothing)
    if os.environ.get("FUZZ_CLOCK", "0") in ("", "0"):
        return
    clock_seed = int(time.time()) % 1000000007
    rc, out = _run(env_extra, CLOCK_SECONDS, clock_seed)
    assert rc == 0 and "identical to the oracle" in out, "pin %s %r CLOCK SEED %d:\n%s" % (name, env_extra, clock_seed, out[-3000:])
