"""Channel sharding across the GPUs of one node (one process per GPU, torch.distributed).

Channels are independent: rank r owns a contiguous channel range and runs its own
engine on its own GPU; nothing is exchanged on the data path.  The only collectives are the optional
gather of PCM to rank 0 (RCCL over xGMI with the "nccl" backend; "gloo" in the CPU tests) and the
max-over-ranks of the benchmark clock.
"""
import torch
import torch.distributed as dist


def channel_range(rank, world, n_channels):
    """Contiguous, balanced split: the first n_channels % world ranks get one channel more."""
    base, extra = divmod(n_channels, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def gather_pcm(pcm_rows, counts, dst=0):
    """Gathers per-rank PCM ([n_local, row] int16) and valid counts to rank dst.

    Returns (list of per-rank PCM tensors, list of per-rank count tensors) on dst, (None, None)
    elsewhere.  Ranks may own different numbers of channels (row length is the same everywhere)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n_local = torch.tensor([pcm_rows.shape[0]], dtype=torch.int64, device=pcm_rows.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    n_max = int(max(int(s.item()) for s in sizes))
    row = pcm_rows.shape[1]
    pad_pcm = torch.zeros((n_max, row), dtype=pcm_rows.dtype, device=pcm_rows.device)
    pad_pcm[:pcm_rows.shape[0]] = pcm_rows
    pad_cnt = torch.zeros(n_max, dtype=counts.dtype, device=counts.device)
    pad_cnt[:counts.shape[0]] = counts
    # collectives move raw bytes: int16 is not a gloo dtype, and bytes are what the sink wants
    pcm_bytes = pad_pcm.view(torch.uint8)
    out_pcm = [torch.empty_like(pcm_bytes) for _ in range(world)] if rank == dst else None
    out_cnt = [torch.empty_like(pad_cnt) for _ in range(world)] if rank == dst else None
    dist.gather(pcm_bytes, out_pcm, dst=dst)
    dist.gather(pad_cnt, out_cnt, dst=dst)
    if rank != dst:
        return None, None
    return ([p.view(pcm_rows.dtype)[:int(s.item())] for p, s in zip(out_pcm, sizes)],
            [c[:int(s.item())] for c, s in zip(out_cnt, sizes)])


def max_over_ranks(seconds, device):
    """The benchmark clock: the slowest rank's elapsed time."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
