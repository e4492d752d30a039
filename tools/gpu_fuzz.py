"""Diagnostic: randomized engine-vs-oracle runs on an MI355X (modes, gains, rotation, squelch gating, AGC, scanner,
block sizes, call boundaries, signal kinds).  `python tools/gpu_fuzz.py [seconds] [seed]`; exits non-zero on the first
difference and prints the configuration that produced it."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                     # noqa: E402
from oracle import bindings as B                       # noqa: E402
from rtlsdrdiags_amd import capi, synth                # noqa: E402

MODES = ["none", "am", "fm", "wbfm", "lsb", "usb"]
DEMOD = {"am": 1, "fm": 2, "wbfm": 3, "lsb": 4, "usb": 4}


def carrier(n, amp):
    pat = np.array([[amp, 0], [0, -amp], [-amp, 0], [0, amp]], np.int16)
    return (128 + np.tile(pat, (n // 4, 1))).astype(np.uint8).reshape(-1)


def signal(rng, n, block_samples):
    kind = rng.integers(0, 8)
    if kind == 0:
        return synth.white_u8(n, int(rng.integers(1 << 30)))
    if kind == 1:
        return synth.rails_u8(n, int(rng.integers(1 << 30)))
    if kind == 2:
        return carrier(n, int(rng.integers(0, 128)))
    if kind == 3:
        return np.full(2 * n, int(rng.integers(0, 256)), np.uint8)
    if kind == 4:
        nblk = n // block_samples
        amps = [int(a) for a in rng.choice([0, 2, 30, 70, 120], nblk)]
        return synth.stepped_amplitude(amps, block_samples=block_samples, seed=int(rng.integers(1 << 30)))
    if kind == 5:   # bursts between noiseless stretches
        a = carrier(n, 50).reshape(-1, 2)
        k0, k1 = sorted(rng.integers(0, n, 2))
        a[k0:k1] = synth.fm_tone(n, seed=int(rng.integers(1 << 30))).reshape(-1, 2)[k0:k1]
        return a.reshape(-1)
    if kind == 6:
        return synth.am_tone(n, seed=int(rng.integers(1 << 30)), depth=float(rng.uniform(0, 1)),
                             amplitude=float(rng.uniform(5, 100)))
    return synth.fm_tone(n, seed=int(rng.integers(1 << 30)), deviation=float(rng.uniform(0, 9e4)),
                         amplitude=float(rng.uniform(2, 127)), sigma=float(rng.choice([0.0, 1.0, 6.0])))


def one_case(rng, O):
    bb = int(rng.choice([256, 2048, 4096, 32768]))
    n_ch = int(rng.integers(1, 6 if os.environ.get("FUZZ_BIG") else 24))
    budget = (1 << 24) if os.environ.get("FUZZ_BIG") else (1 << 21)      # bytes of IQ per case, all channels
    nblk = int(rng.integers(1, max(2, budget // (bb * n_ch))))
    nblk = max(1, min(nblk, 4096))
    n = nblk * bb // 2
    cfg = []
    eng = capi.Engine(n_ch, block_bytes=bb)
    chains = []
    iq = np.empty((n_ch, 2 * n), np.uint8)
    for c in range(n_ch):
        mode = MODES[int(rng.integers(0, 6))]
        gain = None
        if mode != "none" and rng.random() < 0.5:
            gain = float(np.float32(10.0 ** rng.uniform(-2, 8)) * (1 if rng.random() < 0.9 else -1))
        rot = int(rng.integers(-1, 2))
        thr = int(rng.choice([-200, -200, -60, -45, -30]))
        rxg = int(rng.integers(0, 47))
        agc = rng.random() < 0.4
        agc_type, agc_alpha = int(rng.integers(0, 2)), float(np.float32(rng.choice([0.05, 0.5, 0.8])))
        scan = rng.random() < 0.3
        cfg.append((mode, gain, rot, thr, rxg, agc, agc_type, agc_alpha, scan))
        o = O.chain()
        for tgt in (o,):
            tgt.set_mode(mode)
            if gain is not None:
                tgt.set_gain(DEMOD[mode], gain)
            tgt.set_rotation(rot)
            tgt.set_squelch(thr)
            tgt.set_rx_gain_db(rxg)
            if agc:
                tgt.agc_set_type(agc_type); tgt.agc_set_filter_coefficient(agc_alpha); tgt.agc_enable(True)
            if scan:
                tgt.scanner_set_parameters(1000, 9000, 1000); tgt.scanner_start()
        eng.set_mode(mode, first=c, n=1)
        if gain is not None:
            eng.set_gain(DEMOD[mode], gain, first=c, n=1)
        eng.set_rotation(rot, first=c, n=1)
        eng.set_squelch(thr, first=c, n=1)
        eng.set_rx_gain_db(rxg, first=c, n=1)
        if agc:
            eng.agc_set_type(agc_type, first=c, n=1); eng.agc_set_filter_coefficient(agc_alpha, first=c, n=1)
            eng.agc_enable(True, first=c, n=1)
        if scan:
            eng.scanner_set_parameters(1000, 9000, 1000, first=c, n=1); eng.scanner_start(True, first=c, n=1)
        chains.append(o)
        iq[c] = signal(rng, n, bb // 2)
    cuts = sorted(set([0, nblk] + [int(x) for x in rng.integers(0, nblk + 1, int(rng.integers(0, 6)))]))
    got = [[] for _ in range(n_ch)]
    ops_log = []
    got_allowed, got_mag = [], []
    ref_parts = [[[], [], []] for _ in range(n_ch)]
    for a, b in zip(cuts[:-1], cuts[1:]):
        if a and rng.random() < 0.6:     # between calls: the operator changes something on a few channels
            for c in rng.integers(0, n_ch, int(rng.integers(1, 4))):
                c = int(c)
                what = int(rng.integers(0, 10))
                if os.environ.get("FUZZ_OPS"):
                    what = int(rng.choice([int(x) for x in os.environ["FUZZ_OPS"].split(",")]))
                ops_log.append((a, c, what))
                if what == 0:
                    m = MODES[int(rng.integers(0, 6))]
                    chains[c].set_mode(m); eng.set_mode(m, first=c, n=1)
                elif what == 1:
                    chains[c].reset(); eng.reset(first=c, n=1)
                elif what == 2:
                    # any sequence of gain changes, however close together (the engine keeps every change whose
                    # samples are still inside the 2048-sample tail: GainEpochList)
                    d, g = int(rng.integers(1, 5)), float(np.float32(10.0 ** rng.uniform(0, 6)))
                    chains[c].set_gain(d, g); eng.set_gain(d, g, first=c, n=1)
                    ops_log[-1] += (d, g)
                elif what == 3:
                    t = int(rng.choice([-200, -70, -50, -35]))
                    chains[c].set_squelch(t); eng.set_squelch(t, first=c, n=1)
                elif what == 4:
                    r = int(rng.integers(-1, 2))
                    chains[c].set_rotation(r); eng.set_rotation(r, first=c, n=1)
                    ops_log[-1] += (r,)
                elif what == 5:
                    g = int(rng.integers(0, 47))
                    chains[c].set_rx_gain_db(g); eng.set_rx_gain_db(g, first=c, n=1)
                elif what == 6:
                    on = bool(rng.integers(0, 2))
                    chains[c].agc_enable(on); eng.agc_enable(on, first=c, n=1)
                    p = int(rng.integers(-30, 0))
                    chains[c].agc_set_operating_point(p); eng.agc_set_operating_point(p, first=c, n=1)
                elif what == 7:
                    d = int(rng.integers(1, 5))
                    chains[c].reset_demod(d); eng.reset_demod(d, first=c, n=1)
                elif what == 9:
                    k = int(rng.integers(0, 3))
                    if k == 0:
                        a_, b_, i_ = int(rng.integers(1, 5000)), int(rng.integers(5000, 9000)), int(rng.integers(0, 3000))
                        assert chains[c].scanner_set_parameters(a_, b_, i_) == eng.scanner_set_parameters(a_, b_, i_, first=c, n=1)
                    elif k == 1:
                        assert chains[c].scanner_start() == eng.scanner_start(True, first=c, n=1)
                    else:
                        assert chains[c].scanner_stop() == eng.scanner_start(False, first=c, n=1)
                    cfg[c] = cfg[c][:8] + (True,)      # compare the tuned frequency at the end
                else:
                    t_, db_, bl_ = int(rng.integers(0, 2)), int(rng.integers(0, 11)), int(rng.integers(0, 11))
                    al_ = float(np.float32(rng.choice([0.01, 0.3, 0.9])))
                    for tgt, kw in ((chains[c], {}), (eng, dict(first=c, n=1))):
                        tgt.agc_set_type(t_, **kw); tgt.agc_set_deadband(db_, **kw)
                        tgt.agc_set_blanking_limit(bl_, **kw); tgt.agc_set_filter_coefficient(al_, **kw)
        # the call covers all channels - or, sometimes, two calls cover two sub-ranges in a random order
        ranges = [(0, n_ch)]
        if n_ch > 1 and rng.random() < 0.4:
            m = int(rng.integers(1, n_ch))
            ranges = [(0, m), (m, n_ch)] if rng.random() < 0.5 else [(m, n_ch), (0, m)]
        allowed_call = np.zeros((n_ch, b - a), np.uint8)
        mag_call = np.zeros((n_ch, b - a), np.uint32)
        for c0, c1 in ranges:
            pcm, cnt, mag, allowed = eng.accept(iq[c0:c1, a * bb:b * bb], first=c0, n=c1 - c0)
            allowed_call[c0:c1], mag_call[c0:c1] = allowed, mag
            for c in range(c0, c1):
                got[c].append(pcm[c - c0, :cnt[c - c0]])
        for c in range(n_ch):
            r = chains[c].accept_stream(iq[c, a * bb:b * bb], bb)
            for k in range(3):
                ref_parts[c][k].append(r[k])
        got_allowed.append(allowed_call)
        got_mag.append(mag_call)
    got_allowed, got_mag = np.concatenate(got_allowed, axis=1), np.concatenate(got_mag, axis=1)
    ref_out = [tuple(np.concatenate(ref_parts[c][k]) for k in range(3)) for c in range(n_ch)]
    for c in range(n_ch):
        ref, rmag, rallowed = ref_out[c]
        what = None
        if not np.array_equal(got_allowed[c], rallowed):
            what = "allowed"
        elif not np.array_equal(got_mag[c], rmag):
            what = "magnitude"
        elif not np.array_equal(np.concatenate(got[c]), ref):
            what = "pcm"
        elif eng.rx_gain_db(c) != chains[c].rx_gain_db():
            what = "gain"
        elif cfg[c][8] and chains[c].scanner_tuned()[1] > 0 and eng.scanner_tuned(c) != chains[c].scanner_tuned():
            what = "scanner"
        if what:
            print("MISMATCH in %s: channel %d of %d, block_bytes %d, %d blocks, cuts %s, cfg %s" %
                  (what, c, n_ch, bb, nblk, cuts, cfg[c]))
            if what == "pcm":
                g, r = np.concatenate(got[c]), ref
                bad = np.flatnonzero(g != r)
                print("  first differing PCM sample %d of %d (%d differ, last %d): got %s, oracle %s; operations between calls: %s"
                      % (bad[0], len(r), len(bad), bad[-1], g[bad[0]:bad[0] + 6], r[bad[0]:bad[0] + 6], ops_log))
                print("  the calls' first PCM samples: %s; all differing: %s; stats %s" % (np.cumsum([0] + [len(x) for x in got[c]]).tolist(), bad[:20].tolist(), eng.stats()))
            return False
    eng.close()
    return True


def wide_case(rng, O):
    """Many channels in one launch (hundreds to thousands): the launch geometry the small cases never reach - channel
    lists sorted by rotation group and padded, several rounds of the persistent workgroups, CU shares between families,
    gated calls of many rows.  Channels are drawn from a handful of templates (configuration + signal), scattered over
    the batch, so that a few oracle runs check EVERY row; device-pointer calls, so that a big batch is one launch and
    not slices; one operator command on one template between the two calls."""
    k_tpl = int(rng.integers(3, 9))
    lo, hi = (int(x) for x in os.environ.get("FUZZ_WIDE_RANGE", "600,4000").split(","))   # (24,600: the small calls that take
    n_ch = int(rng.integers(lo, hi + 1))                                                   # the one-launch arrangement since round 4)
    nblk = int(rng.integers(1, 5))
    bb = 32768
    while n_ch * nblk * bb > (256 << 20):
        nblk -= 1
    n = nblk * bb // 2
    flags = int(rng.choice([0, 0, 0, 4, 8, 12]))      # 4: every chain streams where its rules allow; 8: the gated calls' pre-pass one call ahead
    tpl = []
    for t in range(k_tpl):
        mode = MODES[int(rng.integers(1, 6))] if rng.random() < 0.95 else "none"
        gain = float(np.float32(10.0 ** rng.uniform(0, 5))) if (mode != "none" and rng.random() < 0.4) else None
        rot = int(rng.choice([1, 1, 0, -1]))
        thr = int(rng.choice([-200, -200, -200, -45]))
        agc = rng.random() < 0.25
        sig = [signal(rng, n, bb // 2) for _ in range(2)]
        tpl.append(dict(mode=mode, gain=gain, rot=rot, thr=thr, agc=agc, sig=sig))
    which = rng.integers(0, k_tpl, n_ch)
    if rng.random() < 0.5:      # WBFM streams only with one rotation selector per launch: half the cases keep it that way
        r0 = [t["rot"] for t in tpl if t["mode"] == "wbfm"]
        for t in tpl:
            if t["mode"] == "wbfm":
                t["rot"] = r0[0]
    eng = capi.Engine(n_ch, block_bytes=bb, flags=flags)
    chains = []
    for t in tpl:
        o = O.chain()
        o.set_mode(t["mode"]); o.set_rotation(t["rot"]); o.set_squelch(t["thr"])
        if t["gain"] is not None:
            o.set_gain(DEMOD[t["mode"]], t["gain"])
        if t["agc"]:
            o.agc_enable(True)
        chains.append(o)
    for c in range(n_ch):
        t = tpl[which[c]]
        eng.set_mode(t["mode"], first=c, n=1)
        if t["rot"] != 1:
            eng.set_rotation(t["rot"], first=c, n=1)
        if t["thr"] != -200:
            eng.set_squelch(t["thr"], first=c, n=1)
        if t["gain"] is not None:
            eng.set_gain(DEMOD[t["mode"]], t["gain"], first=c, n=1)
        if t["agc"]:
            eng.agc_enable(True, first=c, n=1)
    iq_d, pcm_d = eng.dev_alloc(n_ch * 2 * n), eng.dev_alloc(n_ch * (n // 32) * 2)
    cnt_d, mag_d, al_d = eng.dev_alloc(n_ch * 4), eng.dev_alloc(n_ch * nblk * 4), eng.dev_alloc(n_ch * nblk)
    ok = True
    desc = "wide case: %d channels, %d blocks, flags %d, templates %s" % (
        n_ch, nblk, flags, [(t["mode"], t["gain"], t["rot"], t["thr"], t["agc"]) for t in tpl])
    for call in range(2):
        if call == 1 and rng.random() < 0.7:
            t_i = int(rng.integers(0, k_tpl))
            what = int(rng.integers(0, 4))
            members = np.flatnonzero(which == t_i)
            if what == 0:
                m = MODES[int(rng.integers(1, 6))]
                chains[t_i].set_mode(m)
                for c in members: eng.set_mode(m, first=int(c), n=1)
            elif what == 1:
                chains[t_i].reset()
                for c in members: eng.reset(first=int(c), n=1)
            elif what == 2:
                d, g = int(rng.integers(1, 5)), float(np.float32(10.0 ** rng.uniform(0, 5)))
                chains[t_i].set_gain(d, g)
                for c in members: eng.set_gain(d, g, first=int(c), n=1)
            else:
                r = int(rng.integers(-1, 2))
                chains[t_i].set_rotation(r)
                for c in members: eng.set_rotation(r, first=int(c), n=1)
            desc += "; before call 1: op %d on template %d" % (what, t_i)
        iq = np.stack([t["sig"][call] for t in tpl])[which]
        eng.dev_upload(iq_d, iq)
        eng.accept_device(iq_d, 2 * n, pcm_d, cnt_d, mag_d, al_d)
        eng.synchronize()
        pcm = eng.dev_download(pcm_d, n_ch * (n // 32) * 2, np.int16).reshape(n_ch, -1)
        cnt = eng.dev_download(cnt_d, n_ch * 4, np.uint32)
        mag = eng.dev_download(mag_d, n_ch * nblk * 4, np.uint32).reshape(n_ch, nblk)
        al = eng.dev_download(al_d, n_ch * nblk, np.uint8).reshape(n_ch, nblk)
        for t_i in range(k_tpl):
            ref, rmag, rallowed = chains[t_i].accept_stream(tpl[t_i]["sig"][call], bb)
            rows = np.flatnonzero(which == t_i)
            if not len(rows):
                continue
            bad = None
            if not (al[rows] == rallowed).all():
                bad = "allowed"
            elif not (mag[rows] == rmag).all():
                bad = "magnitude"
            elif not (cnt[rows] == len(ref)).all():
                bad = "pcm count"
            elif not (pcm[rows, :len(ref)] == ref).all():
                bad = "pcm"
            if bad:
                wrong = [int(c) for c in rows if not (np.array_equal(pcm[c, :len(ref)], ref) and np.array_equal(mag[c], rmag)
                                                      and np.array_equal(al[c], rallowed) and cnt[c] == len(ref))]
                print("MISMATCH in %s: call %d, template %d, %d of its %d channels (first %s); %s; stats %s"
                      % (bad, call, t_i, len(wrong), len(rows), wrong[:8], desc, eng.stats()))
                ok = False
                break
        if not ok:
            break
    for p_ in (iq_d, pcm_d, cnt_d, mag_d, al_d):
        eng.dev_free(p_)
    eng.close()
    return ok


def short_case(rng, O):
    """Calls in the reference's own granularity: every call is either a few whole blocks or ONE short block of any multiple
    of 64 bytes (Radio.cc:1895-1906 forwards short reads; the rotation strides 8 bytes, IqDataProcessor.cc:586), WBFM
    included since round 4.  A handful of channels in one engine, every mode, resets / gain / mode / rotation / threshold
    changes between calls; PCM, magnitude and squelch flag of every channel and call against the oracle."""
    bb = int(rng.choice([2048, 4096, 32768]))
    n_ch = int(rng.integers(1, 9))
    eng = capi.Engine(n_ch, block_bytes=bb)
    chains, cfg = [], []
    for c in range(n_ch):
        mode = MODES[int(rng.integers(1, 6))] if rng.random() < 0.5 else "wbfm"
        rot = int(rng.integers(-1, 2))
        thr = int(rng.choice([-200, -200, -200, -45]))
        o = O.chain()
        o.set_mode(mode); o.set_rotation(rot); o.set_squelch(thr)
        eng.set_mode(mode, first=c, n=1); eng.set_rotation(rot, first=c, n=1); eng.set_squelch(thr, first=c, n=1)
        chains.append(o)
        cfg.append((mode, rot, thr))
    log, sizes = [], []
    for call in range(int(rng.integers(4, 14))):
        if rng.random() < 0.7:
            nb = 64 * int(rng.integers(1, bb // 64))                       # one short block
        else:
            nb = bb * int(rng.integers(1, max(2, (1 << 17) // bb)))         # whole blocks
        if rng.random() < 0.5:
            c = int(rng.integers(0, n_ch))
            what = int(rng.integers(0, 5))
            log.append((call, c, what))
            if what == 0:
                chains[c].reset(); eng.reset(first=c, n=1)
            elif what == 1:
                d, g = int(rng.integers(1, 5)), float(np.float32(10.0 ** rng.uniform(0, 6)))
                chains[c].set_gain(d, g); eng.set_gain(d, g, first=c, n=1)
                log[-1] += (d, g)
            elif what == 2:
                m = MODES[int(rng.integers(1, 6))]
                chains[c].set_mode(m); eng.set_mode(m, first=c, n=1)
                log[-1] += (m,)
            elif what == 3:
                r = int(rng.integers(-1, 2))
                chains[c].set_rotation(r); eng.set_rotation(r, first=c, n=1)
                log[-1] += (r,)
            else:
                t = int(rng.choice([-200, -50, -35]))
                chains[c].set_squelch(t); eng.set_squelch(t, first=c, n=1)
                log[-1] += (t,)
        sizes.append(nb)
        rows = np.stack([signal(rng, nb // 2, min(bb, nb) // 2) for _ in range(n_ch)])
        pcm, cnt, mag, allowed = eng.accept(rows)
        for c in range(n_ch):
            ref, rmag, rall = chains[c].accept_stream(rows[c], min(nb, bb))
            if cnt[c] != len(ref) or not np.array_equal(pcm[c, :cnt[c]], ref) or not np.array_equal(mag[c], rmag) \
                    or not np.array_equal(allowed[c], rall):
                what = ("count %d vs %d" % (cnt[c], len(ref)) if cnt[c] != len(ref) else
                        "allowed %s vs %s" % (allowed[c], rall) if not np.array_equal(allowed[c], rall) else
                        "magnitude %s vs %s" % (mag[c], rmag) if not np.array_equal(mag[c], rmag) else
                        "pcm: first difference at %d of %d (all: %s; got %s, oracle %s; stats %s)" % (
                            int(np.flatnonzero(pcm[c, :cnt[c]] != ref)[0]), len(ref), np.flatnonzero(pcm[c, :cnt[c]] != ref)[:12].tolist(),
                            pcm[c, :cnt[c]][np.flatnonzero(pcm[c, :cnt[c]] != ref)[:4]].tolist(), ref[np.flatnonzero(pcm[c, :cnt[c]] != ref)[:4]].tolist(), eng.stats()))
                eng.close()
                return "short-block case: bb=%d cfg=%r call %d of %d bytes (calls so far %r), channel %d differs in %s (ops %r)" % (bb, cfg, call, nb, sizes, c, what, log)
    eng.close()
    return None


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    O = B.Oracle()
    t0, cases, n_wide = time.time(), 0, 0
    import json
    if os.environ.get("FUZZ_REPLAY"):       # the generator state a failed run left behind (gpurun_out/fuzz_fail_state.json): that one case again
        rng.bit_generator.state = json.load(open(os.environ["FUZZ_REPLAY"]))
        seconds = 0.0
        wide = os.environ.get("FUZZ_WIDE") == "1" or (os.environ.get("FUZZ_WIDE") is None and rng.random() < 0.03)   # (as the loop below draws it)
        if os.environ.get("FUZZ_SHORT"):
            bad = short_case(rng, O)
            if bad:
                print("MISMATCH:", bad)
            ok = bad is None
        else:
            ok = wide_case(rng, O) if wide else one_case(rng, O)
        print("replayed case:", "identical" if ok else "MISMATCH")
        sys.exit(0 if ok else 1)
    def keep_state():
        try:
            os.makedirs("gpurun_out", exist_ok=True)
            json.dump(state0, open("gpurun_out/fuzz_fail_state.json", "w"))
        except Exception:
            pass
    while time.time() - t0 < seconds:
        state0 = rng.bit_generator.state
        wide = os.environ.get("FUZZ_WIDE") == "1" or (os.environ.get("FUZZ_WIDE") is None and rng.random() < 0.03)
        if os.environ.get("FUZZ_SHORT"):        # every call a short block / a few whole blocks (short_case)
            bad = short_case(rng, O)
            if bad:
                print("MISMATCH:", bad)
                keep_state()
                sys.exit(1)
        elif not (wide_case(rng, O) if wide else one_case(rng, O)):
            keep_state()
            sys.exit(1)
        cases += 1
        n_wide += 1 if wide else 0
    if os.environ.get("FUZZ_SHORT"):
        print("gpu_fuzz: %d short-block cases (every call one short block of any multiple of 64 bytes, or a few whole blocks) "
              "identical to the oracle in %.0f s (seed %d)" % (cases, time.time() - t0, seed))
        return
    print("gpu_fuzz: %d cases (%d of them wide: hundreds to thousands of channels) identical to the oracle in %.0f s (seed %d)"
          % (cases, n_wide, time.time() - t0, seed))


if __name__ == "__main__":
    main()
