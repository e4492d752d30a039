"""Seeded synthetic IQ sources (stand-in for librtlsdr when testing / benchmarking).

The reference receives offset-binary uint8 I/Q from the dongle at 256 kS/s with the wanted
signal at -Fs/4 (Radio.cc:617-618 tunes to f + Fs/4; IqDataProcessor.cc:749 rotates it back).
These generators produce that shape with numpy only; nothing here touches the GPU.
"""
import numpy as np

FS = 256000.0


def _finish(z, sigma, rng):
    i = np.real(z) + rng.normal(0.0, sigma, len(z))
    q = np.imag(z) + rng.normal(0.0, sigma, len(z))
    out = np.empty(2 * len(z), dtype=np.uint8)
    out[0::2] = np.clip(np.rint(i) + 128, 0, 255).astype(np.uint8)
    out[1::2] = np.clip(np.rint(q) + 128, 0, 255).astype(np.uint8)
    return out


def fm_tone(n_samples, seed=1234, deviation=30000.0, tone=1000.0, amplitude=60.0, sigma=3.0):
    """FM carrier at -Fs/4 modulated by a sine tone (SURVEY.md §8(d) config 2 generator)."""
    rng = np.random.default_rng(seed)
    n = np.arange(n_samples, dtype=np.float64)
    phase = -0.5 * np.pi * n + (deviation / tone) * np.sin(2 * np.pi * tone * n / FS)
    return _finish(amplitude * np.exp(1j * phase), sigma, rng)


def am_tone(n_samples, seed=1234, depth=0.5, tone=1000.0, amplitude=50.0, sigma=2.0):
    rng = np.random.default_rng(seed)
    n = np.arange(n_samples, dtype=np.float64)
    env = amplitude * (1.0 + depth * np.sin(2 * np.pi * tone * n / FS))
    return _finish(env * np.exp(-0.5j * np.pi * n), sigma, rng)


def ssb_tone(n_samples, seed=1234, tone=1200.0, amplitude=50.0, sigma=2.0, upper=True):
    rng = np.random.default_rng(seed)
    n = np.arange(n_samples, dtype=np.float64)
    f = tone if upper else -tone
    return _finish(amplitude * np.exp(1j * (-0.5 * np.pi * n + 2 * np.pi * f * n / FS)), sigma, rng)


def white_u8(n_samples, seed=1234):
    """Full-scale white bytes: exercises int8 wrap, -128 negation, (int16) overflow."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, 2 * n_samples, dtype=np.uint8)


def rails_u8(n_samples, seed=1234):
    """Rail-to-rail 0/255 bytes followed by the 255,255,255,255,0,0,0,0 pattern."""
    rng = np.random.default_rng(seed)
    half = n_samples  # bytes
    a = rng.integers(0, 2, half, dtype=np.uint8) * 255
    pat = np.tile(np.array([255, 255, 255, 255, 0, 0, 0, 0], dtype=np.uint8), (2 * n_samples - half + 7) // 8)
    return np.concatenate([a, pat[: 2 * n_samples - half]])


def stepped_amplitude(block_amplitudes, block_samples=16384, seed=1234, sigma=1.0):
    """One FM-tone block per entry with the given carrier amplitude (squelch open/close)."""
    rng = np.random.default_rng(seed)
    out = []
    for k, amp in enumerate(block_amplitudes):
        n = np.arange(block_samples, dtype=np.float64) + k * block_samples
        phase = -0.5 * np.pi * n + 30.0 * np.sin(2 * np.pi * 1000.0 * n / FS)
        out.append(_finish(amp * np.exp(1j * phase), sigma, rng))
    return np.concatenate(out)
