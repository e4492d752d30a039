"""AGC scenarios as data, so that the same sequence of operator commands and block magnitudes can be replayed on
the reference (oracle/_ref), on the oracle restatement and on the GPU engine.

An operation is (code, value):
  0 setType            1 setDeadband         2 setBlankingLimit     3 setAgcFilterCoefficient
  4 setOperatingPoint  5 enable(1)/disable(0) 6 manual IF gain (Radio::setReceiveIfGainInDb)
  7 one block magnitude delivered to the AGC (signalMagnitudeCallback)
`replay` returns (flags, gains): the success flag of every command (1 where the reference returns nothing) and
the receiver's IF gain after every operation.
"""
import numpy as np

SET_TYPE, SET_DEADBAND, SET_BLANKING, SET_ALPHA, SET_OP, ENABLE, SET_GAIN, FEED = range(8)


def replay(chain, codes, values):
    flags = np.ones(len(codes), np.uint8)
    gains = np.zeros(len(codes), np.uint32)
    for k, (code, v) in enumerate(zip(codes, values)):
        code = int(code)
        if code == SET_TYPE:
            flags[k] = chain.agc_set_type(int(v))
        elif code == SET_DEADBAND:
            flags[k] = chain.agc_set_deadband(int(v))
        elif code == SET_BLANKING:
            flags[k] = chain.agc_set_blanking_limit(int(v))
        elif code == SET_ALPHA:
            flags[k] = chain.agc_set_filter_coefficient(float(v))
        elif code == SET_OP:
            chain.agc_set_operating_point(int(v))
        elif code == ENABLE:
            flags[k] = chain.agc_enable(bool(v))
        elif code == SET_GAIN:
            chain.set_rx_gain_db(int(v))
        elif code == FEED:
            chain.agc_feed(int(v))
        gains[k] = chain.rx_gain_db()
    return flags, gains


def random_script(seed, n_ops=600):
    """Mostly magnitudes, drifting in level, sprinkled with every operator command (valid and invalid values)."""
    rng = np.random.default_rng(seed)
    codes, values = [ENABLE], [1.0]
    level = 40.0
    for _ in range(n_ops):
        r = rng.random()
        if r < 0.86:
            level = float(np.clip(level * np.exp(rng.normal(0, 0.35)), 0, 200))
            mag = int(level) if rng.random() < 0.95 else int(rng.integers(0, 300))
            codes.append(FEED); values.append(float(mag))
        elif r < 0.88:
            codes.append(SET_TYPE); values.append(float(rng.integers(0, 3)))
        elif r < 0.90:
            codes.append(SET_DEADBAND); values.append(float(rng.integers(0, 13)))
        elif r < 0.92:
            codes.append(SET_BLANKING); values.append(float(rng.integers(0, 13)))
        elif r < 0.94:
            codes.append(SET_ALPHA); values.append(float(np.float32(rng.choice([0.0005, 0.001, 0.05, 0.3, 0.8, 0.998, 0.999, 1.5]))))
        elif r < 0.96:
            codes.append(SET_OP); values.append(float(rng.integers(-40, 1)))
        elif r < 0.98:
            codes.append(ENABLE); values.append(float(rng.integers(0, 2)))
        else:
            codes.append(SET_GAIN); values.append(float(rng.integers(0, 47)))
    return np.array(codes, np.uint8), np.array(values, np.float32)


def stream(chain, iq_u8, block_bytes):
    """Feeds a chain block by block.  Returns (pcm, allowed per block, IF gain after each block)."""
    pcm, allowed, gains = [], [], []
    for off in range(0, len(iq_u8), block_bytes):
        p, _, a = chain.accept_stream(iq_u8[off:off + block_bytes], block_bytes)
        pcm.append(p)
        allowed.append(int(a[0]))
        gains.append(chain.rx_gain_db())
    return np.concatenate(pcm), np.array(allowed, np.uint8), np.array(gains, np.uint32)


def configure(chain, cfg):
    """cfg: dict with optional type, deadband, blanking, alpha, operating_point, gain, threshold, mode; enables the AGC."""
    if "mode" in cfg: chain.set_mode(cfg["mode"])
    if "threshold" in cfg: chain.set_squelch(int(cfg["threshold"]))
    if "gain" in cfg: chain.set_rx_gain_db(int(cfg["gain"]))
    if "type" in cfg: chain.agc_set_type(int(cfg["type"]))
    if "deadband" in cfg: chain.agc_set_deadband(int(cfg["deadband"]))
    if "blanking" in cfg: chain.agc_set_blanking_limit(int(cfg["blanking"]))
    if "alpha" in cfg: chain.agc_set_filter_coefficient(float(cfg["alpha"]))
    if "operating_point" in cfg: chain.agc_set_operating_point(int(cfg["operating_point"]))
    chain.agc_enable(True)


STREAM_CASES = [   # (name, amplitudes per block, config)
    ("harris_fm", [3, 3, 40, 40, 40, 90, 90, 90, 90, 5, 5, 5, 5, 120, 120, 2, 2, 2, 60, 60, 60, 60, 60, 60],
     dict(mode="fm", threshold=-52, type=1)),
    ("lowpass_usb", [60, 60, 60, 60, 4, 4, 4, 4, 4, 4, 100, 100, 100, 100, 100, 100, 10, 10, 10, 10, 10, 10, 10, 10],
     dict(mode="usb", threshold=-48, type=0, alpha=0.3, deadband=2, blanking=2, operating_point=-9)),
    ("harris_wbfm_open", [20, 20, 20, 80, 80, 80, 80, 80, 80, 6, 6, 6, 6, 6, 6, 6, 6, 127, 127, 127, 127, 30, 30, 30],
     dict(mode="wbfm", type=1, alpha=0.5, deadband=0, blanking=0, operating_point=-20, gain=10)),
]


# ---- FrequencyScanner scenario ------------------------------------------------------------------------
SCAN_AMPS = [2, 2, 60, 60, 2, 2, 2, 90, 2, 2, 2, 2, 70, 2, 2, 2, 2, 2, 2, 2, 2, 50, 2, 2]
SCAN_PARAMS = (100000000, 100300000, 100000)     # start, end, increment in Hz; wraps twice over SCAN_AMPS


def scan_scenario(chain, iq_u8, block_bytes, feed):
    """Commands around a stream: parameters, start (twice), the stream, parameters while scanning (rejected),
    stop (twice), one more block while idle, new parameters, start.  `feed(chain, iq)` pushes bytes through the
    chain and returns (pcm, tuned frequency after each block, tune count after each block).
    Returns (flags, pcm, frequency per block, count per block, final (frequency, count))."""
    chain.set_mode("fm")
    chain.set_squelch(-40)
    flags = [chain.scanner_set_parameters(*SCAN_PARAMS), chain.scanner_start(), chain.scanner_start()]
    pcm, freq, count = feed(chain, iq_u8)
    flags += [chain.scanner_set_parameters(1, 2, 3), chain.scanner_stop(), chain.scanner_stop()]
    p2, f2, c2 = feed(chain, iq_u8[:2 * block_bytes])
    flags += [chain.scanner_set_parameters(5, 50, 7), chain.scanner_start()]
    return (np.array(flags, np.uint8), np.concatenate([pcm, p2]), np.concatenate([freq, f2]).astype(np.uint64),
            np.concatenate([count, c2]).astype(np.uint32), np.array(chain.scanner_tuned(), np.uint64))


def feed_blockwise(block_bytes):
    def feed(chain, iq_u8):
        pcm, freq, count = [], [], []
        for off in range(0, len(iq_u8), block_bytes):
            p, _, _ = chain.accept_stream(iq_u8[off:off + block_bytes], block_bytes)
            pcm.append(p)
            f, n = chain.scanner_tuned()
            freq.append(f)
            count.append(n)
        return np.concatenate(pcm), np.array(freq, np.uint64), np.array(count, np.uint32)
    return feed
