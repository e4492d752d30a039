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


class PcmGatherer:
    """PCM of every rank's channels to rank `dst`, once per step, with nothing but the collective in the step.

    Set-up (once): the ranks exchange their channel counts, every rank allocates one padded send buffer and `dst`
    one receive buffer per rank.  A step then is two `gather` calls over persistent tensors: no size exchange, no
    host synchronisation, no allocation.  Collectives move raw bytes (int16 is not a gloo dtype, and bytes are what
    a sink wants).  PCM is 1/32 of the input volume, so this is far from the xGMI rate; it exists to deliver the
    audio, not to be fast.
    """

    def __init__(self, n_local, row, device, dst=0, count_dtype=torch.int32):
        self.world, self.rank, self.dst = dist.get_world_size(), dist.get_rank(), dst
        n = torch.tensor([n_local], dtype=torch.int64, device=device)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(sizes, n)
        self.sizes = [int(s.item()) for s in sizes]            # the only host read, at set-up
        self.n_local, self.row, self.n_max = n_local, row, max(self.sizes)
        self.send_pcm = torch.zeros((self.n_max, row), dtype=torch.int16, device=device)
        self.send_cnt = torch.zeros(self.n_max, dtype=count_dtype, device=device)
        is_dst = self.rank == dst
        self.recv_pcm = [torch.empty((self.n_max, 2 * row), dtype=torch.uint8, device=device)
                         for _ in range(self.world)] if is_dst else None
        self.recv_cnt = [torch.empty_like(self.send_cnt) for _ in range(self.world)] if is_dst else None

    def gather(self, pcm_rows, counts):
        """pcm_rows [n_local, row] int16, counts [n_local]; both are copied into the persistent send buffers on the
        current stream, so the caller orders its producer before this call (see bench.py: the engine's stream)."""
        self.send_pcm[:self.n_local].copy_(pcm_rows, non_blocking=True)
        self.send_cnt[:self.n_local].copy_(counts, non_blocking=True)
        dist.gather(self.send_pcm.view(torch.uint8), self.recv_pcm, dst=self.dst)
        dist.gather(self.send_cnt, self.recv_cnt, dst=self.dst)

    def result(self):
        """On dst: (per-rank PCM tensors, per-rank count tensors) of the last gather, trimmed to each rank's channels."""
        if self.rank != self.dst:
            return None, None
        return ([p.view(torch.int16)[:n] for p, n in zip(self.recv_pcm, self.sizes)],
                [c[:n] for c, n in zip(self.recv_cnt, self.sizes)])


class NativeGatherer:
    """The same gather through the engine's own C ABI (iqd_gather_*: RCCL point-to-point transfers on the ENGINE's
    stream, csrc/iqd_gather.cpp) - what a C++ host would call.  Nothing of torch is on the data path: the transfers
    read the engine's output buffers where they lie (the accept that filled them and the next accept that will overwrite
    them are on the same stream).  torch.distributed is used once, to hand the communicator id to the other ranks."""

    def __init__(self, engine, n_local, row, device, dst=0):
        from . import capi
        self.world, self.rank, self.dst = dist.get_world_size(), dist.get_rank(), dst
        n = torch.tensor([n_local], dtype=torch.int64, device=device)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(sizes, n)
        self.sizes = [int(s.item()) for s in sizes]
        self.n_local, self.row, self.n_max = n_local, row, max(self.sizes)
        box = [capi.Gatherer.unique_id() if self.rank == dst else None]
        dist.broadcast_object_list(box, src=dst)
        self._g = capi.Gatherer(engine, box[0], self.rank, self.world, dst)
        is_dst = self.rank == dst
        self.recv_pcm = torch.zeros((self.world, self.n_max, row), dtype=torch.int16, device=device) if is_dst else None
        self.recv_cnt = torch.zeros((self.world, self.n_max), dtype=torch.int32, device=device) if is_dst else None

    def gather(self, pcm_rows, counts):
        """pcm_rows [n_local, row] int16 and counts [n_local] (4-byte) as the engine wrote them, contiguous."""
        self._g.gather(pcm_rows.data_ptr(), [s * self.row * 2 for s in self.sizes],
                       self.recv_pcm.data_ptr() if self.recv_pcm is not None else 0, self.n_max * self.row * 2)
        self._g.gather(counts.data_ptr(), [s * 4 for s in self.sizes],
                       self.recv_cnt.data_ptr() if self.recv_cnt is not None else 0, self.n_max * 4)

    def result(self):
        if self.rank != self.dst:
            return None, None
        return ([self.recv_pcm[r, :n] for r, n in enumerate(self.sizes)], [self.recv_cnt[r, :n] for r, n in enumerate(self.sizes)])

    def info(self):
        """What the communicator reports about itself: RCCL's version code, ncclCommCount, whether the process's own
        copy of the library was reused."""
        return self._g.info()

    def close(self):
        self._g.close()


def gather_pcm(pcm_rows, counts, dst=0):
    """One-off gather (set-up and step in one call); a timed loop keeps a PcmGatherer instead."""
    g = PcmGatherer(pcm_rows.shape[0], pcm_rows.shape[1], pcm_rows.device, dst=dst, count_dtype=counts.dtype)
    g.gather(pcm_rows, counts)
    return g.result()


def max_over_ranks(seconds, device):
    """The benchmark clock: the slowest rank's elapsed time."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
