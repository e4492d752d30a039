"""CPU, world_size 2 over gloo: the N>1 path of the benchmark / sharded deployment.

The rank body is bench.py's own (`bench.rank_body`: staging, warm-up, barriers, K timed steps, the PCM gather with
its persistent buffers, the max-over-ranks clock, the result line); only the engine is a stand-in that works on
host memory, so the control flow the driver's multi-GPU runs depend on is exactly what runs here.  The data path
itself has no collective."""
import ctypes
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from rtlsdrdiags_amd import shard  # noqa: E402


def test_channel_ranges_cover_everything():
    for world in (1, 2, 3, 8):
        for n in (1, 7, 8, 4096, 32768, 65537):
            got = []
            for r in range(world):
                first, cnt = shard.channel_range(r, world, n)
                got.extend(range(first, first + cnt))
            assert got == list(range(n))
            sizes = [shard.channel_range(r, world, n)[1] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


class FakeEngine:
    """Stand-in for capi.Engine over host memory: 'PCM' of channel c is a function of the channel's global index,
    its mode and its input bytes, so that the gathered result proves which rank produced which rows."""

    def __init__(self, n_channels, flags, rank):
        self.n, self.flags, self.rank = n_channels, flags, rank
        self.modes = ["none"] * n_channels
        self.rot = [1] * n_channels
        self.calls = 0
        self.profiling = False
        self.agc = False

    def set_mode(self, mode, first=0, n=None):
        for c in range(first, first + (self.n - first if n is None else n)):
            self.modes[c] = mode

    def set_rotation(self, r, first=0, n=None):
        for c in range(first, first + (self.n - first if n is None else n)):
            self.rot[c] = r

    def set_squelch(self, t, first=0, n=None):
        pass

    def agc_set_type(self, t, first=0, n=None):
        pass

    def agc_enable(self, on=True, first=0, n=None):
        self.agc = on

    def accept_device(self, iq_ptr, bytes_per_ch, pcm_ptr, cnt_ptr=0, mag_ptr=0, allowed_ptr=0):
        n_pcm = bytes_per_ch // 64
        iq = np.frombuffer((ctypes.c_uint8 * (self.n * bytes_per_ch)).from_address(iq_ptr), dtype=np.uint8).reshape(self.n, -1)
        pcm = np.frombuffer((ctypes.c_int16 * (self.n * n_pcm)).from_address(pcm_ptr), dtype=np.int16).reshape(self.n, -1)
        for c in range(self.n):
            tagv = ["none", "am", "fm", "wbfm", "lsb", "usb"].index(self.modes[c]) * 1000 + int(iq[c, :64].sum()) % 1000
            pcm[c, :] = np.int16(tagv % 30000)
            pcm[c, 0] = np.int16(self.rot[c])
        if cnt_ptr:
            cnt = np.frombuffer((ctypes.c_int32 * self.n).from_address(cnt_ptr), dtype=np.int32)
            cnt[:] = n_pcm
        self.calls += 1

    def synchronize(self):
        pass

    def set_profiling(self, on):
        self.profiling = on

    def stats(self):
        return {"chain_kernel_ms": 0.25 * self.calls, "chain_kernel_count": self.calls, "state_checks": 0,
                "state_repairs": 0, "segment_repairs": 0, "stream_launches": 0}

    def close(self):
        pass


def _worker(rank, world, port, argv, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    args = bench.parse_args(argv)
    engines = []

    def make_engine(n_channels, flags):
        engines.append(FakeEngine(n_channels, flags, rank))
        return engines[-1]

    captured = {}
    real_gatherer = shard.PcmGatherer

    class Spy(real_gatherer):   # keeps a handle on the gatherer bench.py builds, to read what arrived on rank 0
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            captured["g"] = self

    shard.PcmGatherer = Spy
    out = bench.rank_body(args, rank, world, torch.device("cpu"), make_engine, dist, torch)
    shard.PcmGatherer = real_gatherer
    eng = engines[0]
    assert eng.calls == args.warmup + args.steps
    if rank == 0:
        pcm, cnts = captured["g"].result()
        rows = torch.cat(pcm).numpy()
        q.put((out, rows[:, 0].tolist(), rows[:, 1].tolist(), torch.cat(cnts).numpy().tolist(),
               captured["g"].sizes, eng.modes, eng.agc))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def _run(argv, world=2):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, argv, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    return res


def test_bench_rank_body_mixed_channels_world_size_2():
    """configs[3] shape (mixed modes), 5 channels per rank: channel g of the job runs mode g % 5 whichever rank owns it."""
    argv = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--mode", "mixed", "--channels", "5", "--log2-samples", "12",
            "--gather", "--no-cpu-baseline", "--no-host-path", "--prewarm-ms", "0"]
    out, rot_col, tag_col, counts, sizes, modes0, agc0 = _run(argv)
    assert sizes == [5, 5] and len(tag_col) == 10
    names = ["am", "fm", "wbfm", "lsb", "usb"]
    assert modes0 == names                                     # rank 0 owns job channels 0..4
    idx = {"am": 1, "fm": 2, "wbfm": 3, "lsb": 4, "usb": 5}
    assert [t // 1000 for t in tag_col] == [idx[names[g % 5]] for g in range(10)]   # rank 1's rows continue the pattern
    assert counts == [(1 << 12) // 32] * 10 and not agc0
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["channels_per_gpu"] == 5 and "gathered" in out["config"]["sharding"]
    # value = samples of ALL ranks / the slowest rank's time
    assert abs(out["value"] - 2 * 5 * 4096 * 3 / (out["ms_per_step"] * 3e-3) / 1e6) / out["value"] < 2e-2   # (both figures are rounded in the line)
    assert out["roofline"]["kernel_ms"] == 0.25 and "cpu_baseline" not in out


def test_bench_rank_body_ssb_stress_world_size_2():
    """configs[4] shape: LSB/USB alternate and the rotation selector cycles with the JOB-wide channel index."""
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--mode", "ssb_stress", "--channels", "3", "--log2-samples", "12",
            "--gather", "--no-cpu-baseline", "--no-host-path", "--prewarm-ms", "0"]
    out, rot_col, tag_col, counts, sizes, modes0, agc0 = _run(argv)
    assert [t // 1000 for t in tag_col] == [4 if g % 2 == 0 else 5 for g in range(6)]
    assert rot_col == [(1, 0, -1)[g % 3] for g in range(6)]
    assert agc0 and out["metric"].startswith("IQ MSamples/s through SSB_STRESS chains")


def test_prewarm_runs_the_same_number_of_steps_on_every_rank():
    """The untimed clock-settle phase in front of the warm-up steps is a step COUNT derived from the workload alone - a
    wall-clock loop would leave the ranks with different numbers of gather collectives."""
    import bench
    a = bench.parse_args(["--gpus", "2", "--steps", "2", "--warmup", "1", "--mode", "mixed", "--channels", "5", "--log2-samples", "12",
                          "--gather", "--no-cpu-baseline", "--no-host-path", "--prewarm-ms", "0.1"])
    eng = FakeEngine(5, 0, 0)
    out = bench.rank_body(a, 0, 1, torch.device("cpu"), lambda n_channels, flags: eng, None, torch)
    # (W + K steps for the from-idle figure, the settle phase, then W + K again)
    assert eng.calls == (1 + 2) + 5 + 1 + 2 and "5 untimed steps" in out["config"]["prewarm"], (eng.calls, out["config"]["prewarm"])
    assert out["from_idle_ms_per_step"] is not None
    a.no_from_idle = True
    eng = FakeEngine(5, 0, 0)
    out = bench.rank_body(a, 0, 1, torch.device("cpu"), lambda n_channels, flags: eng, None, torch)
    assert eng.calls == 5 + 1 + 2 and out["from_idle_ms_per_step"] is None


def test_presets_name_the_baseline_configurations():
    import bench
    a = bench.parse_args([])
    assert (a.mode, a.channels, a.log2_samples) == ("wbfm", 1, 28) and "configs[1]" in a.what
    a = bench.parse_args(["--config", "2"])
    assert (a.mode, a.channels, a.log2_samples) == ("fm", 4096, 16) and "configs[2]" in a.what
    a = bench.parse_args(["--config", "4"])
    assert (a.mode, a.channels) == ("ssb_stress", 8192) and "65536" in a.what
    a = bench.parse_args(["--mode", "am", "--channels", "64"])
    assert a.what is None and a.tag is None                    # not a BASELINE configuration: labelled as such


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts torch.distributed.run as a child process,
    relays rank 0's one JSON line and the exit code (the driver calls it exactly like this).  Here with the
    stand-in engine over gloo; on a GPU box the same entry runs one rank per GPU over RCCL."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                           "TORCHELASTIC_RUN_ID")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--mode", "mixed", "--channels", "5", "--log2-samples", "12", "--gather", "--no-cpu-baseline",
                        "--no-host-path", "--prewarm-ms", "0", "--standin", "tests.test_shard_gloo:FakeEngine"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                               # ONE line on stdout, whatever the ranks printed
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert "STAND-IN" in out["data"] and out["config"]["channels_per_gpu"] == 5


def test_bench_relays_a_failing_rank():
    """A rank that dies must not look like success: the launcher's exit code comes back."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                           "TORCHELASTIC_RUN_ID")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--channels", "2", "--log2-samples", "12", "--no-cpu-baseline", "--no-host-path",
                        "--standin", "tests.test_shard_gloo:NoSuchEngine"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert r.returncode != 0 and not r.stdout.decode().strip()


# ---- shard.NativeGatherer's control flow against a stand-in for capi.Gatherer --------------------
class FakeCapiGatherer:
    """What csrc/iqd_gather.cpp does, over host memory and gloo point-to-point transfers: the root posts one receive per
    peer into that peer's row, every other rank one send.  Keeps a log of the calls, so that the test can hold the
    id hand-over, the per-rank byte tables and the two gathers per step to what the C ABI expects."""
    UID = bytes(range(128))
    made = []

    def __init__(self, engine, unique_id, rank, world, root=0):
        assert bytes(unique_id) == self.UID, "every rank must be handed the ROOT's id"
        self.engine, self.rank, self.world, self.root = engine, rank, world, root
        self.calls = []
        FakeCapiGatherer.made.append(self)

    @staticmethod
    def unique_id():
        assert dist.get_rank() == FakeCapiGatherer.expect_root, "only the root makes the id"
        return FakeCapiGatherer.UID

    def gather(self, send_dev, bytes_per_rank, recv_dev, row_stride):
        assert len(bytes_per_rank) == self.world and all(b <= row_stride for b in bytes_per_rank)
        assert (recv_dev != 0) == (self.rank == self.root)
        self.calls.append((list(bytes_per_rank), row_stride))
        mine = bytes_per_rank[self.rank]
        src = torch.frombuffer((ctypes.c_uint8 * max(mine, 1)).from_address(send_dev), dtype=torch.uint8)[:mine]
        if self.rank == self.root:
            for p in range(self.world):
                n = bytes_per_rank[p]
                row = torch.frombuffer((ctypes.c_uint8 * max(n, 1)).from_address(recv_dev + p * row_stride), dtype=torch.uint8)[:n]
                if p == self.root:
                    row.copy_(src)
                elif n:
                    dist.recv(row, src=p)
        elif mine:
            dist.send(src.clone(), dst=self.root)

    def info(self):
        return {"version": 22606, "ranks": self.world, "library_reused": True}

    def close(self):
        self.calls.append("closed")


def _native_worker(rank, world, port, root, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rtlsdrdiags_amd import capi
    real = capi.Gatherer
    capi.Gatherer = FakeCapiGatherer
    FakeCapiGatherer.expect_root = root
    try:
        n_local, row = (3, 8) if rank == 0 else (5, 8)          # uneven channel counts: rows are padded to the largest
        engine = object()
        g = shard.NativeGatherer(engine, n_local, row, torch.device("cpu"), dst=root)
        assert g.sizes == [3, 5] and g.n_max == 5 and FakeCapiGatherer.made[-1].engine is engine
        for step in range(3):
            pcm = (torch.arange(n_local * row, dtype=torch.int16).view(n_local, row) + 1000 * rank + 100 * step).contiguous()
            cnt = torch.full((n_local,), row - rank - step, dtype=torch.int32)
            g.gather(pcm, cnt)
        fake = FakeCapiGatherer.made[-1]
        assert fake.calls == [([3 * row * 2, 5 * row * 2], 5 * row * 2), ([3 * 4, 5 * 4], 5 * 4)] * 3   # PCM then counts, every step
        assert g.info()["ranks"] == world
        if rank == root:
            pcm_r, cnt_r = g.result()
            got = [(p.numpy().copy(), c.numpy().copy()) for p, c in zip(pcm_r, cnt_r)]
            q.put(got)
        else:
            assert g.result() == (None, None)
        g.close()
        assert fake.calls[-1] == "closed"
    finally:
        capi.Gatherer = real
    dist.barrier()
    dist.destroy_process_group()


def test_native_gatherer_control_flow_with_a_standin_communicator():
    """shard.NativeGatherer - id broadcast from the root, per-rank byte table, two gathers per step on persistent
    buffers, a root that is not rank 0 - in two gloo processes with capi.Gatherer replaced by a host-memory stand-in:
    what is left untested without a multi-GPU node is RCCL itself."""
    world, root = 2, 1
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_native_worker, args=(r, world, port, root, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    step, row = 2, 8                                           # the last step's rows, each rank's trimmed to its channel count
    for rank, (n_local, (pcm, cnt)) in enumerate(zip((3, 5), got)):
        want = np.arange(n_local * row, dtype=np.int16).reshape(n_local, row) + 1000 * rank + 100 * step
        assert pcm.shape == (n_local, row) and np.array_equal(pcm, want), rank
        assert np.array_equal(cnt, np.full(n_local, row - rank - step, np.int32))


def test_bench_refuses_more_ranks_than_gpus():
    """`--gpus N` with N above the visible device count stops before any rank is started, with a message."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                           "TORCHELASTIC_RUN_ID")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert r.returncode != 0 and not r.stdout.decode().strip()
    assert "--gpus 64" in r.stderr.decode() and "nothing was started" in r.stderr.decode()
