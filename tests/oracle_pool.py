"""Test infrastructure: the oracle for MANY channels at once, one worker process per host core (the GPU boxes have 256
hardware threads, so every channel of a 4096-channel launch can be held to the oracle in a second or two instead of a
sample of them - VERDICT r3 item 5).

The workers are plain child processes of their own (`python tests/oracle_pool.py <job directory> <k> <n>`), started with
subprocess: nothing of the test process - its HIP runtime, its threads, its engine objects - is inherited (a fork of a
process with a live GPU context is undefined behaviour, and a forked worker that dies in an inherited finaliser leaves a
multiprocessing pool waiting for ever).  Rows go over as one .npy file that the workers map; results come back as one
.npz per worker."""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker_main(job_dir, k, n_workers):
    sys.path.insert(0, ROOT)
    from oracle import bindings as B
    with open(os.path.join(job_dir, "job.json")) as f:
        job = json.load(f)
    rows = np.load(os.path.join(job_dir, "rows.npy"), mmap_mode="r")
    O = B.Oracle()
    out = {}
    for c in range(k, len(rows), n_workers):
        ch = O.chain()
        ch.set_mode(job["modes"][c])
        ch.set_rotation(job["rots"][c])
        if job["threshold"] is not None:
            ch.set_squelch(job["threshold"])
        if job["agc"] is not None:
            ch.agc_set_type(job["agc"])
            ch.agc_enable(True)
        row = np.array(rows[c])
        for call in range(job["calls"]):
            pcm, mag, allowed = ch.accept_stream(row)
            out["p_%d_%d" % (call, c)], out["m_%d_%d" % (call, c)], out["a_%d_%d" % (call, c)] = pcm, mag, allowed
        ch.close()
    np.savez(os.path.join(job_dir, "out_%d.npz" % k), **out)


def oracle_all_channels(rows, modes, rots=None, calls=1, threshold=None, agc=None, workers=None, timeout=900):
    """rows [n_ch][bytes] uint8 (the same bytes go in at every call); modes / rots per channel.  Returns three lists
    indexed [call][channel]: PCM arrays, per-block magnitudes, per-block squelch results."""
    n_ch = len(rows)
    workers = workers or max(1, min(96, (os.cpu_count() or 2) - 2, n_ch))
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 2 * rows.nbytes + (1 << 28) else None
    job_dir = tempfile.mkdtemp(prefix="iqd_oracle_", dir=base)
    try:
        np.save(os.path.join(job_dir, "rows.npy"), np.ascontiguousarray(rows))
        with open(os.path.join(job_dir, "job.json"), "w") as f:
            json.dump({"modes": list(modes), "rots": [int(r) for r in (rots if rots is not None else [1] * n_ch)],
                       "calls": calls, "threshold": threshold, "agc": agc}, f)
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), job_dir, str(k), str(workers)]) for k in range(workers)]
        for p in procs:
            rc = p.wait(timeout=timeout)
            assert rc == 0, "an oracle worker exited with %d" % rc
        pcm = [[None] * n_ch for _ in range(calls)]
        mag = [[None] * n_ch for _ in range(calls)]
        allowed = [[None] * n_ch for _ in range(calls)]
        for k in range(workers):
            with np.load(os.path.join(job_dir, "out_%d.npz" % k)) as z:
                for c in range(k, n_ch, workers):
                    for call in range(calls):
                        pcm[call][c], mag[call][c], allowed[call][c] = z["p_%d_%d" % (call, c)], z["m_%d_%d" % (call, c)], z["a_%d_%d" % (call, c)]
        return pcm, mag, allowed
    finally:
        shutil.rmtree(job_dir, ignore_errors=True)


def bench_rows(base_u8, n_ch, first_global, row_bytes):
    """numpy twin of bench.per_channel_rows: channel c (job-wide index g) = the periodic signal rolled by
    2 * ((g * 37) % 1009) bytes, four bytes of its own (g, little endian) xor-ed in at the front."""
    period = len(base_u8)
    out = np.empty((n_ch, row_bytes), np.uint8)
    reps = (row_bytes + period - 1) // period + 1
    tiled = np.tile(base_u8, reps)
    for c in range(n_ch):
        g = first_global + c
        s = 2 * ((g * 37) % 1009)
        out[c] = tiled[s:s + row_bytes]
        out[c, :4] ^= np.frombuffer(np.uint32(g).tobytes(), np.uint8)
    return out


def gated_rows(loud_u8, quiet_u8, n_ch, first_global, row_bytes, patterns, block_bytes=32768):
    """numpy twin of bench.gated_rows (configs[4]): the channel's own roll of the loud and of the quiet signal, block by
    block after the pattern of its class g % len(patterns)."""
    rows = bench_rows(loud_u8, n_ch, first_global, row_bytes)
    quiet = bench_rows(quiet_u8, n_ch, first_global, row_bytes)
    for c in range(n_ch):
        pat = patterns[(first_global + c) % len(patterns)]
        for b in range(max(1, row_bytes // block_bytes)):
            if not pat[b % 4]:
                rows[c, b * block_bytes:(b + 1) * block_bytes] = quiet[c, b * block_bytes:(b + 1) * block_bytes]
    return rows


if __name__ == "__main__":
    _worker_main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
