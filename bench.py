#!/usr/bin/env python3
"""bench.py — IQ MSamples/s through the WBFM chain on N MI355X GPUs, with the HBM roofline fraction
of the chain kernel and the CPU baseline timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1]): WBFM, 1 channel per GPU,
2^28 synthetic IQ samples (512 MiB of uint8 I/Q) resident in HBM before the timed region.
A "step" is one iqd_accept_iq_device() call over the whole batch: the fused chain kernel, the
tile hand-off verification, the squelch bookkeeping and the state update.  With N > 1 every rank
runs its own channel on its own GPU (independent channels are the shard; no data-path collective),
so the job is weak-scaled and `value` is the sum over ranks divided by the slowest rank's time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

ALGO_BYTES_PER_SAMPLE = 2.0 + 2.0 / 32.0     # int8 I + int8 Q in, int16 PCM out at 1/32 rate
HBM_PEAK_GBS = 8000.0                        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(period_u8, seconds_target=12.0):
    """Times the reference CPU chain (oracle/_ref, the unmodified reference sources) — or, where that
    library is absent, the oracle port — on one host core over a bounded sample of the same signal."""
    from oracle import bindings as B
    if B.have_ref():
        chain, kind = B.Reference().chain(), "reference"
    else:
        chain, kind = B.Oracle().chain(), "port"
    chain.set_mode("wbfm")
    n_period = len(period_u8) // 2
    # calibrate on 2^21 samples, then run ~seconds_target
    t0 = time.perf_counter()
    chain.accept_stream(period_u8[: 2 << 21])
    rate = (1 << 21) / (time.perf_counter() - t0)
    reps = max(1, min(64, int(seconds_target * rate / n_period)))
    t0 = time.perf_counter()
    for _ in range(reps):
        chain.accept_stream(period_u8)
    dt = time.perf_counter() - t0
    return {"value": round(reps * n_period / dt / 1e6, 3), "unit": "MSamples/s", "cores": 1, "kind": kind,
            "sample": "WBFM, 1 channel, %d x 2^24 samples of the bench signal through "
                      "IqDataProcessor::acceptIqData in 32768-byte blocks, 1 thread" % reps}


def host_path(eng, iq_dev, n, n_ch, reps=3):
    """The same workload through the host-pointer entry point (iqd_accept_iq): IQ in page-locked host memory,
    uploaded in slices that overlap the kernels, PCM downloaded.  Reported beside `value`, never as it."""
    iq = eng.host_array((n_ch, 2 * n))
    pcm = eng.host_array((n_ch, n // 32), np.int16)
    iq[:] = iq_dev.view(n_ch, -1).cpu().numpy()
    cnt = np.zeros(n_ch, np.uint32)
    eng.accept_into(iq, pcm, cnt)
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.accept_into(iq, pcm, cnt)
    dt = (time.perf_counter() - t0) / reps
    eng.host_free(iq)
    eng.host_free(pcm)
    return {"value": round(n * n_ch / dt / 1e6, 1), "unit": "MSamples/s", "ms_per_step": round(1e3 * dt, 3),
            "GB_per_s_over_pcie": round(2.0625 * n * n_ch / dt / 1e9, 2),
            "what": "iqd_accept_iq from page-locked host buffers (upload, kernels, PCM download), %d calls" % reps}


def measured_traffic(mode, n_ch, log2_samples):
    """HBM bytes per launch of the chain kernel from the committed rocprofv3 PMC summary of the same
    workload (profiles/), or None: bench.py itself cannot collect PMC counters."""
    path = os.path.join(ROOT, "profiles", "r1_wbfm_2p28_pmc.json")
    if mode == "wbfm" and n_ch == 1 and log2_samples == 28 and os.path.exists(path):
        with open(path) as f:
            return json.load(f).get("traffic_bytes_per_launch")
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2-samples", type=int, default=28, help="IQ samples per channel per step (default 2^28)")
    ap.add_argument("--channels", type=int, default=1, help="channels per GPU")
    ap.add_argument("--mode", default="wbfm")
    ap.add_argument("--signal", default="fm_tone", choices=["fm_tone", "white", "carrier", "quiet", "small", "large"],
                    help="synthetic input: the FM test tone of SURVEY 8(d) (default) or uniform random bytes")
    ap.add_argument("--no-magnitude", action="store_true", help="IQD_F_NO_MAGNITUDE: skip the per-block squelch magnitudes (nobody listens to them at the default threshold)")
    ap.add_argument("--squelch", type=int, default=None, help="squelch threshold in dBFS (default: the reference's -200, never closes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive iqd_accept_iq measurement")
    ap.add_argument("--gather", action="store_true", help="also gather the PCM to rank 0 over RCCL each step")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world

    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the engine has no CPU path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:   # launched by torch.distributed.run (also with 1 rank)
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from rtlsdrdiags_amd import capi, shard, synth

    n = 1 << args.log2_samples
    n_ch = args.channels
    period = min(n, 1 << 24)
    dev = torch.device("cuda", local_rank)
    # the same seeded signal on every rank, phase-continuous over the period
    if args.signal == "fm_tone":
        period_u8 = synth.fm_tone(period, seed=1234)
    elif args.signal == "white":
        period_u8 = synth.white_u8(period, seed=1234)
    elif args.signal == "carrier":      # unmodulated, noiseless: every sample hits the same table cell
        period_u8 = synth.fm_tone(period, seed=1234, deviation=0.0, sigma=0.0)
    elif args.signal == "small":        # weak signal: the table accesses stay within a few cache lines
        period_u8 = synth.fm_tone(period, seed=1234, amplitude=6.0, sigma=1.0)
    elif args.signal == "large":        # strong signal: a wide ring through the table
        period_u8 = synth.fm_tone(period, seed=1234, amplitude=120.0)
    else:                               # speech-like: 3 kHz deviation
        period_u8 = synth.fm_tone(period, seed=1234, deviation=3000.0)
    iq = torch.from_numpy(period_u8).to(dev).repeat(n // period)
    if n_ch > 1:
        iq = iq.unsqueeze(0).repeat(n_ch, 1).contiguous()
    pcm = torch.zeros(n_ch * (n // 32), dtype=torch.int16, device=dev)
    cnt = torch.zeros(n_ch, dtype=torch.int32, device=dev)
    mag = torch.zeros(n_ch * (2 * n // 32768), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    eng = capi.Engine(n_channels=n_ch, device=local_rank, flags=1 if args.no_magnitude else 0)
    if args.mode == "mixed":     # BASELINE configs[3]: ch % 5 -> {AM, FM, WBFM, LSB, USB}
        for c in range(n_ch):
            eng.set_mode(["am", "fm", "wbfm", "lsb", "usb"][c % 5], first=c, n=1)
    else:
        eng.set_mode(args.mode)

    if args.squelch is not None:
        eng.set_squelch(args.squelch)

    def step():
        if args.no_magnitude:
            eng.accept_device(iq.data_ptr(), 2 * n, pcm.data_ptr())
        else:
            eng.accept_device(iq.data_ptr(), 2 * n, pcm.data_ptr(), cnt.data_ptr(), mag.data_ptr())
        if args.gather and dist is not None:
            shard.gather_pcm(pcm.view(n_ch, -1), cnt, dst=0)

    for _ in range(args.warmup):
        step()
    eng.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    eng.set_profiling(True)
    k0 = eng.stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    k1 = eng.stats()
    eng.set_profiling(False)
    if dist is not None:
        elapsed = shard.max_over_ranks(elapsed, dev)

    total_samples = float(n) * n_ch * args.steps * world
    value = total_samples / elapsed / 1e6
    kern_ms = (k1["chain_kernel_ms"] - k0["chain_kernel_ms"]) / max(1, k1["chain_kernel_count"] - k0["chain_kernel_count"])
    achieved = ALGO_BYTES_PER_SAMPLE * n * n_ch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0

    if rank == 0:
        out = {
            "metric": "IQ MSamples/s through WBFM chain at 1/2/4/8 GPUs; % HBM roofline",
            "value": round(value, 1), "unit": "MSamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int8/int16 Q15 + f32", "data": "synthetic" if args.signal == "fm_tone" else "synthetic (%s)" % args.signal,
            "config": {"workload": "%s, %d channel(s) per GPU, 2^%d IQ samples per channel per step, "
                                   "uint8 I/Q resident in HBM (BASELINE configs[1])"
                                   % (args.mode.upper(), n_ch, args.log2_samples),
                       "sharding": "independent channels, one per rank, no data-path collective"
                                   + ("; PCM gathered to rank 0 over RCCL" if args.gather else "")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": measured_traffic(args.mode, n_ch, args.log2_samples),
                         "kernel": "%s_chain_kernel" % ("am" if args.mode in ("am", "lsb", "usb") else ("first family's" if args.mode == "mixed" else args.mode)), "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_SAMPLE * n * n_ch},
            "state_checks": k1["state_checks"] - k0["state_checks"],
            "state_repairs": k1["state_repairs"] - k0["state_repairs"],
            "segment_repairs": k1["segment_repairs"] - k0["segment_repairs"],
        }
        if world == 1 and not args.no_host_path:
            out["host_path"] = host_path(eng, iq, n, n_ch)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(period_u8)
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
