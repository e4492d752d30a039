#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the UNMODIFIED reference (oracle/_ref).

Run in the build container only (needs /root/reference):
    make -C oracle ref && python tests/golden/make_golden.py
Each fixture stores the exact uint8 IQ input next to what the reference's
IqDataProcessor::acceptIqData produced for it (PCM per mode, per-block squelch magnitude and
signalAllowed), so the fixtures stay valid without numpy RNG stream compatibility.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bindings as B          # noqa: E402
from rtlsdrdiags_amd import synth          # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
MODES = ["am", "fm", "wbfm", "lsb", "usb"]


def run_all_modes(R, u8, block_bytes=32768, threshold=None, rx_gain=None, gains=None):
    out = {}
    for mode in ["none"] + MODES:
        c = R.chain()
        c.set_rx_gain_db(24 if rx_gain is None else rx_gain)
        c.set_mode(mode)
        if threshold is not None:
            c.set_squelch(threshold)
        if gains:
            for which, g in gains.items():
                c.set_gain(which, g)
        pcm, mag, allowed = c.accept_stream(u8, block_bytes)
        out["pcm_" + mode] = pcm
        out["magnitude"] = mag
        out["allowed"] = allowed
        c.close()
    R.chain().set_rx_gain_db(24)
    return out


DEMOD_ENTRY_STEPS = [("proc", 32768), ("demod", 32768), ("demod", 8448), ("proc", 32768), ("demod", 256), ("demod", 24576),
                     ("proc", 4096)]


def demod_entry(R):
    """The demodulators' own acceptIqData(int8_t *, uint32_t) (e.g. WbFmDemodulator.h:31; what
    demodulatorResearch/demodulators/demod.cc:262-285 calls) interleaved with the processor's acceptIqData on the same
    objects: "proc" steps are uint8 blocks through IqDataProcessor (-128, +Fs/4, squelch open), "demod" steps signed
    bytes - full range, -128 included - straight into the demodulator; lengths are multiples of 256."""
    rng = np.random.default_rng(29)
    out = {"kinds": np.array([k for k, _ in DEMOD_ENTRY_STEPS]), "lengths": np.array([n for _, n in DEMOD_ENTRY_STEPS])}
    tone = synth.fm_tone(2 * 16384 + 2048, seed=30)
    inputs, at = [], 0
    for k, (kind, n) in enumerate(DEMOD_ENTRY_STEPS):
        if kind == "proc":
            x = tone[at:at + n].copy()
            at += n
        else:
            x = rng.integers(-128, 128, n).astype(np.int8)
            if k == 1:                       # a tone the demodulators make something of, then the rails
                x[:16384] = (synth.fm_tone(8192, seed=31).astype(np.int16) - 128).astype(np.int8)
                x[16384:16384 + 64] = -128
        inputs.append(x)
        out["in%d" % k] = x
    for mode in MODES:
        c = R.chain()
        c.set_mode(mode)
        for k, (kind, n) in enumerate(DEMOD_ENTRY_STEPS):
            if kind == "proc":
                pcm, _, _ = c.accept_stream(inputs[k], n)
            else:
                pcm = c.demod_accept(mode, inputs[k])
            out["pcm_%s_%d" % (mode, k)] = pcm
        c.close()
    np.savez_compressed(os.path.join(OUT, "demod_entry.npz"), **out)


CAPTURE = "/root/reference/demodulatorResearch/yoyo.iq"
CAPTURE_FIRST_BLOCK, CAPTURE_BLOCKS = 56, 4


def capture(R):
    """An off-air signal: blocks 56..59 (32768 bytes each) of the reference's own recording
    demodulatorResearch/yoyo.iq - signed bytes on disk, +128 here to give what the dongle delivers - through
    IqDataProcessor::acceptIqData in every mode.  The excerpt holds a real dropout of the carrier (block 58: the
    magnitude falls from 55 to 9 for a few milliseconds), so the second fixture runs it in 4096-byte blocks with the squelch
    at -33 dBFS: three blocks are rejected in mid-stream and the chains skip them."""
    s8 = np.fromfile(CAPTURE, dtype=np.int8)
    at = CAPTURE_FIRST_BLOCK * 32768
    u8 = (s8[at:at + CAPTURE_BLOCKS * 32768].astype(np.int16) + 128).astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "capture_excerpt.npz"), iq=u8, block_bytes=32768,
                        first_byte=at, **run_all_modes(R, u8))
    g = run_all_modes(R, u8, block_bytes=4096, threshold=-33, rx_gain=24)
    assert 0 < g["allowed"].sum() < len(g["allowed"])
    np.savez_compressed(os.path.join(OUT, "capture_gated.npz"), iq=u8, block_bytes=4096, first_byte=at,
                        threshold=-33, rx_gain_db=24, **g)


def demod_tool(R):
    """The reference's own offline harness as a PROGRAM (oracle/_ref/ref_demod = demodulatorResearch/demodulators/demod.cc,
    unmodified, built by oracle/Makefile): signed bytes on stdin, `-d <1..5>`, PCM on stdout.  Pins the harness level of
    rtlsdrdiags_amd/bin/iqdemod_file `demod` - including what the program really does for -d 4: its switch has no break
    (demod.cc:232-242), so LSB falls through to USB and types 4 and 5 give the same PCM."""
    import subprocess
    tool = os.path.join(ROOT, "oracle", "_ref", "ref_demod")
    x = (synth.fm_tone(5 * 8192 + 1024, seed=41).astype(np.int16) - 128).astype(np.int8)     # 5 reads of 16384 bytes and one of 2048
    x[3 * 16384:3 * 16384 + 96] = -128
    out = {"iq_s8": x}
    for t in range(1, 6):
        r = subprocess.run([tool, "-d", str(t)], input=x.tobytes(), stdout=subprocess.PIPE, check=True)
        out["pcm_%d" % t] = np.frombuffer(r.stdout, dtype=np.int16)
        assert len(out["pcm_%d" % t]) == len(x) // 64
    assert np.array_equal(out["pcm_4"], out["pcm_5"]) and not np.array_equal(out["pcm_1"], out["pcm_5"])
    np.savez_compressed(os.path.join(OUT, "demod_tool.npz"), **out)


def main():
    R = B.Reference()
    if len(sys.argv) > 1:                    # only the named fixtures (python make_golden.py demod_entry capture)
        for name in sys.argv[1:]:
            {"demod_entry": demod_entry, "capture": capture, "demod_tool": demod_tool}[name](R)
        return
    demod_entry(R)
    capture(R)
    demod_tool(R)
    blk = 16384

    # (i) modulated tone at -Fs/4 + noise, moderate amplitude
    u8 = synth.fm_tone(4 * blk, seed=1)
    np.savez_compressed(os.path.join(OUT, "fm_tone.npz"), iq=u8, block_bytes=32768,
                        **run_all_modes(R, u8))
    u8 = synth.am_tone(2 * blk, seed=2)
    np.savez_compressed(os.path.join(OUT, "am_tone.npz"), iq=u8, block_bytes=32768,
                        **run_all_modes(R, u8))
    u8 = synth.ssb_tone(2 * blk, seed=5)
    np.savez_compressed(os.path.join(OUT, "ssb_tone.npz"), iq=u8, block_bytes=32768,
                        **run_all_modes(R, u8))

    # (ii) edge inputs: full-scale white bytes, rail-to-rail, the 255x4/0x4 pattern
    u8 = synth.white_u8(3 * blk, seed=3)
    np.savez_compressed(os.path.join(OUT, "white.npz"), iq=u8, block_bytes=32768,
                        **run_all_modes(R, u8))
    u8 = synth.rails_u8(2 * blk, seed=4)
    np.savez_compressed(os.path.join(OUT, "rails.npz"), iq=u8, block_bytes=32768,
                        **run_all_modes(R, u8))
    # huge demodulator gains: drives the (int16) float casts through wrap-around
    u8 = synth.fm_tone(2 * blk, seed=6, amplitude=100.0)
    gains = {1: 30000.0, 2: 2.0e6, 3: 8.0e6, 4: 30000.0}
    np.savez_compressed(os.path.join(OUT, "cast_overflow.npz"), iq=u8, block_bytes=32768,
                        gain_am=gains[1], gain_fm=gains[2], gain_wbfm=gains[3], gain_ssb=gains[4],
                        **run_all_modes(R, u8, gains=gains))

    # squelch open/close: amplitude-stepped 4096-byte blocks, threshold -40 dBFS, gain 24 dB
    amps = [2, 2, 45, 45, 2, 2, 2, 90, 2, 110, 0, 0, 60, 2, 2, 60]
    u8 = synth.stepped_amplitude(amps, block_samples=2048, seed=7)
    np.savez_compressed(os.path.join(OUT, "squelch_steps.npz"), iq=u8, block_bytes=4096,
                        threshold=-40, rx_gain_db=24,
                        **run_all_modes(R, u8, block_bytes=4096, threshold=-40, rx_gain=24))

    # mode switching mid-stream without reset (Radio never resets)
    u8 = synth.fm_tone(6 * blk, seed=8)
    c = R.chain()
    seq = ["wbfm", "fm", "wbfm", "usb", "am", "lsb"]
    pcm = []
    for k, mode in enumerate(seq):
        c.set_mode(mode)
        p, _, _ = c.accept_stream(u8[k * 32768:(k + 1) * 32768])
        pcm.append(p)
    c.close()
    np.savez_compressed(os.path.join(OUT, "mode_switch.npz"), iq=u8, block_bytes=32768,
                        sequence=np.array(seq), pcm=np.concatenate(pcm))

    # primitives: quantised taps recovered from the reference by a -32768 impulse,
    # clamp behaviour on full-scale int16 noise, float FIR/IIR impulse/step responses
    O = B.Oracle()
    prim = {}
    rng = np.random.default_rng(11)
    x16 = rng.integers(-32768, 32768, 4096).astype(np.int16)
    xsat = np.where(rng.random(4096) < 0.5, -32768, 32767).astype(np.int16)
    prim["x16"], prim["xsat"] = x16, xsat
    for name, factor in [("wbfm_pre", 1), ("wbfm_d1", 4), ("wbfm_d2", 4), ("audio40", 2),
                         ("fm_tuner", 4), ("am_s1", 4), ("am_s2", 4), ("am_s3", 2),
                         ("ssb_delay", 1), ("ssb_hilbert", 1)]:
        h = O.taps_f32(name)
        imp = np.zeros(len(h) + 3, np.int16)
        imp[0] = -32768
        prim["negimp_" + name] = R.fir_q15(h, imp)
        if factor == 1:
            prim["y16_" + name] = R.fir_q15(h, x16)
            prim["ysat_" + name] = R.fir_q15(h, xsat)
        else:
            prim["y16_" + name] = R.decimate_q15(h, factor, x16)
            prim["ysat_" + name] = R.decimate_q15(h, factor, xsat)
    # the reference's own filter demos (Filters/testFirFilter.cc:28-69, testIirFilter.cc:35-110)
    demo_taps = np.array([1, 2, 3, 4, 1, 1, 1, 8], np.float32)
    impulse = np.zeros(16, np.float32); impulse[0] = 1
    step = np.ones(32, np.float32)
    prim["demo_fir_impulse"] = R.fir_f32(demo_taps, impulse)
    prim["demo_iir_half_step"] = R.iir_f32([1.0], [0.5], step)
    prim["demo_dcblock_step"] = R.iir_f32([1.0, -1.0], [-0.95], step)
    xf = rng.normal(0, 5000, 2048).astype(np.float32)
    prim["xf"] = xf
    prim["deemph_xf"] = R.iir_f32([0.0253863, 0.0253863], [-0.9492274], xf)
    prim["dcblock_xf"] = R.iir_f32([1.0, -1.0], [-0.95], xf)
    prim["dbfs_0_299"] = np.array([R.dbfs(m) for m in range(300)], np.int32)
    s8 = rng.integers(-128, 128, 4096).astype(np.int8)
    prim["rot_in"] = s8
    prim["rot_up"] = R.rotate(s8, +1)
    prim["rot_down"] = R.rotate(s8, -1)
    np.savez_compressed(os.path.join(OUT, "primitives.npz"), **prim)

    # stand-alone resamplers: float Decimator / Interpolator, Interpolator_int16 (Filters/)
    rs = {}
    rng = np.random.default_rng(17)
    rs["x"] = rng.normal(0, 2000, 3000).astype(np.float32)
    rs["x16"] = rng.integers(-32768, 32768, 3000).astype(np.int16)
    rs["xsat"] = np.where(rng.random(3000) < 0.5, -32768, 32767).astype(np.int16)
    for name, n_taps, factor in [("a", 33, 4), ("b", 7, 3), ("c", 64, 8), ("d", 5, 1)]:
        h = (rng.normal(0, 0.25, n_taps)).astype(np.float32)
        rs["h_" + name], rs["f_" + name] = h, factor
        rs["dec_" + name] = R.decimate_f32(h, factor, rs["x"])
        rs["int_" + name] = R.interpolate_f32(h, factor, rs["x"])
        rs["i16_" + name] = R.interpolate_q15(h, factor, rs["x16"])
        rs["i16sat_" + name] = R.interpolate_q15(np.clip(4 * h, -1, 0.99997).astype(np.float32), factor, rs["xsat"])
    np.savez_compressed(os.path.join(OUT, "resample.npz"), **rs)

    # AutomaticGainControl (src_diags/AutomaticGainControl.cc compiled unmodified; see oracle/ref_shim.cc):
    # operator commands + magnitudes straight into the AGC, and the AGC inside the acceptIqData flow
    sys.path.insert(0, os.path.dirname(OUT))
    import agc_script as A                      # noqa: E402
    agc = {}
    for seed in (1, 2, 3):
        codes, values = A.random_script(seed, 500)
        flags, gains = A.replay(R.chain(agc=True), codes, values)
        agc.update({"script%d_codes" % seed: codes, "script%d_values" % seed: values,
                    "script%d_flags" % seed: flags, "script%d_gains" % seed: gains})
    # magnitudes 5x8, 100x6, 20x4 with every default
    c = R.chain(agc=True)
    c.agc_enable(True)
    ka = []
    for m in [5] * 8 + [100] * 6 + [20] * 4:
        c.agc_feed(m)
        ka.append(c.rx_gain_db())
    agc["defaults_gains"] = np.array(ka, np.uint32)
    for name, amps, cfg in A.STREAM_CASES:
        u8 = synth.stepped_amplitude(amps, block_samples=2048, seed=21)
        c = R.chain(agc=True)
        A.configure(c, cfg)
        pcm, allowed, gains = A.stream(c, u8, 4096)
        agc.update({name + "_iq": u8, name + "_pcm": pcm, name + "_allowed": allowed, name + "_gains": gains})
    # FrequencyScanner (src_diags/FrequencyScanner.cc compiled unmodified): the tuning commands it issues
    u8 = synth.stepped_amplitude(A.SCAN_AMPS, block_samples=2048, seed=23)
    flags, pcm, freq, count, final = A.scan_scenario(R.chain(scanner=True), u8, 4096, A.feed_blockwise(4096))
    agc.update({"scan_iq": u8, "scan_flags": flags, "scan_pcm": pcm, "scan_freq": freq, "scan_count": count,
                "scan_final": final})
    np.savez_compressed(os.path.join(OUT, "agc.npz"), **agc)

    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if f.endswith(".npz"))
    print("golden fixtures written: %.1f KiB" % (total / 1024.0))


if __name__ == "__main__":
    main()
