"""CPU, world_size 2 over gloo: the N>1 path of the benchmark / sharded deployment — channel ranges,
the PCM gather and the max-over-ranks clock.  (The data path itself has no collective.)"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rtlsdrdiags_amd import shard


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


def _worker(rank, world, port, n_channels, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, cnt = shard.channel_range(rank, world, n_channels)
    # stand-in for the engine's output: channel c's PCM row is filled with c, count = c + 1
    rows = torch.stack([torch.full((16,), c, dtype=torch.int16) for c in range(first, first + cnt)])
    counts = torch.tensor([c + 1 for c in range(first, first + cnt)], dtype=torch.int32)
    pcm, cnts = shard.gather_pcm(rows, counts, dst=0)
    slow = shard.max_over_ranks(0.5 + rank, torch.device("cpu"))
    if rank == 0:
        allp = torch.cat(pcm).numpy()
        allc = torch.cat(cnts).numpy()
        q.put((allp[:, 0].tolist(), allc.tolist(), slow))
    else:
        assert pcm is None and cnts is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_and_clock_world_size_2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n_channels = 5   # uneven split: 3 + 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_channels, q)) for r in range(2)]
    for p in procs:
        p.start()
    first_col, counts, slow = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert first_col == list(range(n_channels))
    assert counts == [c + 1 for c in range(n_channels)]
    assert abs(slow - 1.5) < 1e-9
