"""ctypes binding of include/iqdemod.h (libiqdemod.so).

Mirrors the C ABI one to one; the names, argument meaning and error behaviour are those of the
header, which in turn cites the reference's IqDataProcessor interface.  No torch types cross
this boundary: device buffers are passed as integer addresses.
"""
import ctypes as C
import os

import numpy as np

from .build import LIB as _DEFAULT_LIB

LIB = os.environ.get("IQD_LIB", _DEFAULT_LIB)   # experiments: alternative builds of the same ABI

MODE = {"none": 0, "am": 1, "fm": 2, "wbfm": 3, "lsb": 4, "usb": 5}
DEMOD = {"am": 1, "fm": 2, "wbfm": 3, "ssb": 4}
F_NO_MAGNITUDE = 0x1
F_WBFM_TILES, F_WBFM_STREAM, F_PREPASS_OVERLAP = 0x2, 0x4, 0x8


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("n_channels", C.c_uint32), ("block_bytes", C.c_uint32),
                ("device", C.c_int32), ("flags", C.c_uint32), ("reserved", C.c_uint32 * 3)]


class Stats(C.Structure):
    _fields_ = [("accepts", C.c_uint64), ("samples", C.c_uint64), ("kernel_launches", C.c_uint64),
                ("state_checks", C.c_uint64), ("state_repairs", C.c_uint64),
                ("chain_kernel_ms", C.c_double), ("chain_kernel_count", C.c_uint64), ("segment_repairs", C.c_uint64),
                ("stream_launches", C.c_uint64), ("device_launches", C.c_uint64), ("device_copies", C.c_uint64),
                ("mixed_launches", C.c_uint64)]


class AgcState(C.Structure):
    _fields_ = [("enabled", C.c_uint32), ("type", C.c_uint32), ("operating_point_dbfs", C.c_int32),
                ("deadband_db", C.c_uint32), ("blanking_limit", C.c_uint32), ("alpha", C.c_float),
                ("rx_gain_db", C.c_uint32), ("if_gain_db", C.c_uint32), ("filtered_if_gain_db", C.c_float),
                ("blanking_counter", C.c_uint32), ("gain_was_adjusted", C.c_uint32),
                ("normalized_level_dbfs", C.c_int32), ("signal_magnitude", C.c_uint32)]


class IqdError(RuntimeError):
    def __init__(self, status, detail):
        super().__init__("libiqdemod: %s (%d): %s" % (_lib().iqd_strerror(status).decode(), status, detail))
        self.status = status


EXPORTS = [
    "iqd_abi_version", "iqd_strerror", "iqd_last_error", "iqd_create", "iqd_destroy", "iqd_set_mode",
    "iqd_set_gain", "iqd_set_squelch", "iqd_set_rx_gain_db", "iqd_set_rotation", "iqd_reset", "iqd_reset_demod",
    "iqd_accept_iq", "iqd_accept_iq_device", "iqd_synchronize", "iqd_get_stats", "iqd_set_profiling",
    "iqd_get_channel_mode", "iqd_get_channel_gain", "iqd_dev_alloc", "iqd_dev_free", "iqd_dev_upload",
    "iqd_dev_download", "iqd_dev_tile", "iqd_stream", "iqd_debug_stamps", "iqd_debug_stamps_ext", "iqd_host_alloc", "iqd_host_free",
    "iqd_get_rx_gain_db", "iqd_agc_set_type", "iqd_agc_set_deadband", "iqd_agc_set_blanking_limit",
    "iqd_agc_set_operating_point", "iqd_agc_set_filter_coefficient", "iqd_agc_enable", "iqd_agc_get_state",
    "iqd_set_gain_trace", "iqd_get_gain_trace", "iqd_scanner_set_parameters", "iqd_scanner_start",
    "iqd_scanner_get", "iqd_get_frequency_trace", "iqd_front_end", "iqd_front_end_device", "iqd_convert_fs_over_4",
    "iqd_resampler_create", "iqd_resampler_destroy", "iqd_resampler_reset", "iqd_resampler_out_count",
    "iqd_resampler_run", "iqd_resampler_run_device", "iqd_gather_unique_id", "iqd_gather_create", "iqd_gather_pcm", "iqd_gather_destroy",
    "iqd_demod_accept", "iqd_demod_set_sideband", "iqd_get_device", "iqd_gather_info", "iqd_device_count",
]

_LIB = None


def _lib():
    """Loads libiqdemod.so (raises if it has not been built — there is no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB):
        raise FileNotFoundError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB)
    L = C.CDLL(LIB)
    vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
    L.iqd_abi_version.restype = u32
    L.iqd_strerror.restype = C.c_char_p
    L.iqd_strerror.argtypes = [C.c_int]
    L.iqd_last_error.restype = C.c_char_p
    L.iqd_last_error.argtypes = [vp]
    L.iqd_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.iqd_destroy.argtypes = [vp]
    L.iqd_destroy.restype = None
    L.iqd_set_mode.argtypes = [vp, u32, u32, C.c_int]
    L.iqd_set_gain.argtypes = [vp, u32, u32, C.c_int, C.c_float]
    L.iqd_set_squelch.argtypes = [vp, u32, u32, C.c_int32]
    L.iqd_set_rx_gain_db.argtypes = [vp, u32, u32, u32]
    L.iqd_set_rotation.argtypes = [vp, u32, u32, C.c_int]
    L.iqd_reset.argtypes = [vp, u32, u32]
    L.iqd_reset_demod.argtypes = [vp, u32, u32, C.c_int]
    L.iqd_accept_iq.argtypes = [vp, u32, u32, vp, sz, vp, vp, vp, vp]
    L.iqd_accept_iq_device.argtypes = [vp, u32, u32, vp, sz, vp, vp, vp, vp]
    L.iqd_synchronize.argtypes = [vp]
    L.iqd_demod_accept.argtypes = [vp, u32, u32, C.c_int, vp, sz, vp]
    L.iqd_demod_set_sideband.argtypes = [vp, u32, u32, C.c_int]
    L.iqd_front_end.argtypes = [vp, u32, u32, vp, sz, vp]
    L.iqd_resampler_create.argtypes = [vp, C.c_int, vp, u32, u32, u32, C.POINTER(vp)]
    L.iqd_resampler_destroy.argtypes = [vp]
    L.iqd_resampler_destroy.restype = None
    L.iqd_resampler_reset.argtypes = [vp]
    L.iqd_resampler_out_count.argtypes = [vp, sz]
    L.iqd_resampler_out_count.restype = sz
    L.iqd_resampler_run.argtypes = [vp, vp, sz, vp]
    L.iqd_resampler_run_device.argtypes = [vp, vp, sz, vp]
    L.iqd_front_end_device.argtypes = [vp, u32, u32, vp, sz, vp]
    L.iqd_convert_fs_over_4.argtypes = [vp, C.c_int, vp, sz]
    L.iqd_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.iqd_set_profiling.argtypes = [vp, C.c_int]
    L.iqd_get_channel_mode.argtypes = [vp, u32, C.POINTER(C.c_int)]
    L.iqd_get_channel_gain.argtypes = [vp, u32, C.c_int, C.POINTER(C.c_float)]
    L.iqd_dev_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.iqd_dev_free.argtypes = [vp, vp]
    L.iqd_dev_upload.argtypes = [vp, vp, vp, sz]
    L.iqd_dev_download.argtypes = [vp, vp, vp, sz]
    L.iqd_dev_tile.argtypes = [vp, vp, sz, sz]
    L.iqd_host_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.iqd_host_free.argtypes = [vp, vp]
    L.iqd_get_rx_gain_db.argtypes = [vp, u32, C.POINTER(u32)]
    L.iqd_agc_set_type.argtypes = [vp, u32, u32, u32]
    L.iqd_agc_set_deadband.argtypes = [vp, u32, u32, u32]
    L.iqd_agc_set_blanking_limit.argtypes = [vp, u32, u32, u32]
    L.iqd_agc_set_operating_point.argtypes = [vp, u32, u32, C.c_int32]
    L.iqd_agc_set_filter_coefficient.argtypes = [vp, u32, u32, C.c_float]
    L.iqd_agc_enable.argtypes = [vp, u32, u32, C.c_int]
    L.iqd_agc_get_state.argtypes = [vp, u32, C.POINTER(AgcState)]
    L.iqd_set_gain_trace.argtypes = [vp, C.c_int]
    L.iqd_get_gain_trace.argtypes = [vp, u32, u32, vp, sz]
    u64 = C.c_uint64
    L.iqd_scanner_set_parameters.argtypes = [vp, u32, u32, u64, u64, u64]
    L.iqd_scanner_start.argtypes = [vp, u32, u32, C.c_int]
    L.iqd_scanner_get.argtypes = [vp, u32, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_int)]
    L.iqd_get_frequency_trace.argtypes = [vp, u32, u32, vp, sz]
    L.iqd_stream.argtypes = [vp]
    L.iqd_stream.restype = vp
    _LIB = L
    return L


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Engine:
    """N independent channels, each one reference IqDataProcessor + its four demodulators."""

    def __init__(self, n_channels=1, block_bytes=0, device=-1, flags=0):
        self._L = _lib()
        self._h = C.c_void_p()
        cfg = Config(self._L.iqd_abi_version(), n_channels, block_bytes, device, flags)
        rc = self._L.iqd_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise IqdError(rc, "iqd_create failed (is a HIP device visible?)")
        self.n_channels = n_channels
        self.block_bytes = block_bytes or 32768

    def close(self):
        if getattr(self, "_h", None):
            self._L.iqd_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _check(self, rc):
        if rc != 0:
            raise IqdError(rc, self._L.iqd_last_error(self._h).decode())

    def _range(self, first, n):
        return first, (self.n_channels - first if n is None else n)

    # ---- the reference's control surface -------------------------------------------------
    def set_mode(self, mode, first=0, n=None):
        f, n = self._range(first, n)
        self._check(self._L.iqd_set_mode(self._h, f, n, int(MODE.get(mode, mode))))

    def set_gain(self, demod, gain, first=0, n=None):
        f, n = self._range(first, n)
        self._check(self._L.iqd_set_gain(self._h, f, n, int(DEMOD.get(demod, demod)), C.c_float(gain)))

    def set_squelch(self, threshold, first=0, n=None):
        f, n = self._range(first, n)
        self._check(self._L.iqd_set_squelch(self._h, f, n, int(threshold)))

    def set_rx_gain_db(self, gain_db, first=0, n=None):
        f, n = self._range(first, n)
        self._check(self._L.iqd_set_rx_gain_db(self._h, f, n, int(gain_db)))

    def set_rotation(self, rotation, first=0, n=None):
        f, n = self._range(first, n)
        self._check(self._L.iqd_set_rotation(self._h, f, n, int(rotation)))

    def reset(self, first=0, n=None):
        f, n = self._range(first, n)
        self._check(self._L.iqd_reset(self._h, f, n))

    def reset_demod(self, demod, first=0, n=None):
        f, n = self._range(first, n)
        self._check(self._L.iqd_reset_demod(self._h, f, n, int(DEMOD.get(demod, demod))))

    def demod_accept(self, demod, iq_s8, first=0, n=None):
        """{Am,Fm,WbFm,Ssb}Demodulator::acceptIqData: SIGNED bytes [n][bytes] (or one row) straight into the
        demodulator, no squelch; 'lsb' / 'usb' select the sideband first, like the reference's harness does."""
        f, n = self._range(first, n)
        if demod in ("lsb", "usb"):
            self._check(self._L.iqd_demod_set_sideband(self._h, f, n, 1 if demod == "lsb" else 0))
            demod = "ssb"
        iq_s8 = np.ascontiguousarray(iq_s8, dtype=np.int8)
        one = iq_s8.ndim == 1
        rows = iq_s8.reshape(n, -1)
        pcm = np.zeros((n, rows.shape[1] // 64), np.int16)
        self._check(self._L.iqd_demod_accept(self._h, f, n, int(DEMOD.get(demod, demod)), _np_ptr(rows), rows.shape[1], _np_ptr(pcm)))
        return pcm[0] if one else pcm

    # ---- AutomaticGainControl: True/False like the reference's setters -----------------------
    def _agc(self, fn, value, first, n):
        f, n = self._range(first, n)
        rc = fn(self._h, f, n, value)
        if rc in (-1, -6):   # IQD_EINVAL / IQD_EALREADY: the reference method returns false
            return False
        self._check(rc)
        return True

    def agc_set_type(self, t, first=0, n=None):
        return self._agc(self._L.iqd_agc_set_type, int(t), first, n)

    def agc_set_deadband(self, db, first=0, n=None):
        return self._agc(self._L.iqd_agc_set_deadband, int(db), first, n)

    def agc_set_blanking_limit(self, limit, first=0, n=None):
        return self._agc(self._L.iqd_agc_set_blanking_limit, int(limit), first, n)

    def agc_set_operating_point(self, dbfs, first=0, n=None):
        return self._agc(self._L.iqd_agc_set_operating_point, int(dbfs), first, n)

    def agc_set_filter_coefficient(self, a, first=0, n=None):
        return self._agc(self._L.iqd_agc_set_filter_coefficient, C.c_float(a), first, n)

    def agc_enable(self, on=True, first=0, n=None):
        return self._agc(self._L.iqd_agc_enable, 1 if on else 0, first, n)

    def agc_state(self, ch):
        st = AgcState()
        self._check(self._L.iqd_agc_get_state(self._h, ch, C.byref(st)))
        return {k: getattr(st, k) for k, _ in AgcState._fields_}

    def rx_gain_db(self, ch=0):
        g = C.c_uint32()
        self._check(self._L.iqd_get_rx_gain_db(self._h, ch, C.byref(g)))
        return g.value

    def set_gain_trace(self, on=True):
        self._check(self._L.iqd_set_gain_trace(self._h, 1 if on else 0))

    def gain_trace(self, n_blocks, first=0, n=None):
        f, n = self._range(first, n)
        out = np.zeros((n, n_blocks), np.uint32)
        self._check(self._L.iqd_get_gain_trace(self._h, f, n, _np_ptr(out), n_blocks))
        return out

    # ---- FrequencyScanner -------------------------------------------------------------------
    def scanner_set_parameters(self, start_hz, end_hz, increment_hz, first=0, n=None):
        f, n = self._range(first, n)
        rc = self._L.iqd_scanner_set_parameters(self._h, f, n, int(start_hz), int(end_hz), int(increment_hz))
        if rc == -6:
            return False
        self._check(rc)
        return True

    def scanner_start(self, start=True, first=0, n=None):
        f, n = self._range(first, n)
        rc = self._L.iqd_scanner_start(self._h, f, n, 1 if start else 0)
        if rc == -6:
            return False
        self._check(rc)
        return True

    def scanner_tuned(self, ch=0):
        """(frequency the radio was last told to tune to, number of tuning commands)"""
        hz, n, sc = C.c_uint64(), C.c_uint64(), C.c_int()
        self._check(self._L.iqd_scanner_get(self._h, ch, C.byref(hz), C.byref(n), C.byref(sc)))
        return hz.value, n.value

    def frequency_trace(self, n_blocks, first=0, n=None):
        f, n = self._range(first, n)
        out = np.zeros((n, n_blocks), np.uint64)
        self._check(self._L.iqd_get_frequency_trace(self._h, f, n, _np_ptr(out), n_blocks))
        return out

    # ---- data path ------------------------------------------------------------------------
    def accept(self, iq_u8, first=0, n=None):
        """iq_u8: [n_ch, bytes_per_ch] uint8 host array.  Returns (pcm rows, counts, magnitude, allowed)."""
        f, n = self._range(first, n)
        iq_u8 = np.ascontiguousarray(iq_u8, dtype=np.uint8).reshape(n, -1)
        bpc = iq_u8.shape[1]
        nblk = bpc // self.block_bytes if bpc % self.block_bytes == 0 else 0
        pcm = np.zeros((n, bpc // 64), dtype=np.int16)
        cnt = np.zeros(n, dtype=np.uint32)
        mag = np.zeros((n, max(nblk, 1)), dtype=np.uint32)
        allowed = np.zeros((n, max(nblk, 1)), dtype=np.uint8)
        self._check(self._L.iqd_accept_iq(self._h, f, n, _np_ptr(iq_u8), bpc, _np_ptr(pcm), _np_ptr(cnt),
                                          _np_ptr(mag), _np_ptr(allowed)))
        return pcm, cnt, mag, allowed

    def front_end(self, iq_u8, first=0, n=None):
        """u8 -> s8 -> rotation only: the bytes the reference leaves in its buffer / dumps over UDP."""
        f, n = self._range(first, n)
        iq_u8 = np.ascontiguousarray(iq_u8, dtype=np.uint8).reshape(n, -1)
        out = np.zeros(iq_u8.shape, np.int8)
        self._check(self._L.iqd_front_end(self._h, f, n, _np_ptr(iq_u8), iq_u8.shape[1], _np_ptr(out)))
        return out

    def convert_fs_over_4(self, direction, s8):
        """IqDataProcessor::upconvertByFsOver4 (+1) / downconvertByFsOver4 (-1) on a copy of signed bytes."""
        buf = np.array(s8, dtype=np.int8, copy=True)
        self._check(self._L.iqd_convert_fs_over_4(self._h, int(direction), _np_ptr(buf), buf.size))
        return buf

    def accept_device(self, iq_dev, bytes_per_ch, pcm_dev, count_dev=0, mag_dev=0, allowed_dev=0, first=0, n=None):
        f, n = self._range(first, n)
        self._check(self._L.iqd_accept_iq_device(self._h, f, n, C.c_void_p(iq_dev), bytes_per_ch,
                                                 C.c_void_p(pcm_dev), C.c_void_p(count_dev or None),
                                                 C.c_void_p(mag_dev or None), C.c_void_p(allowed_dev or None)))

    def synchronize(self):
        self._check(self._L.iqd_synchronize(self._h))

    def stream_handle(self):
        """The hipStream_t the engine launches on (an integer, for torch.cuda.ExternalStream)."""
        return int(self._L.iqd_stream(self._h) or 0)

    # ---- diagnostics / device helpers -------------------------------------------------------
    def stats(self):
        s = Stats()
        self._check(self._L.iqd_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    def set_profiling(self, on):
        self._check(self._L.iqd_set_profiling(self._h, 1 if on else 0))

    def channel_mode(self, ch):
        m = C.c_int()
        self._check(self._L.iqd_get_channel_mode(self._h, ch, C.byref(m)))
        return m.value

    def channel_gain(self, ch, demod):
        g = C.c_float()
        self._check(self._L.iqd_get_channel_gain(self._h, ch, int(DEMOD.get(demod, demod)), C.byref(g)))
        return g.value

    def debug_stamps_ext(self, n=64):
        out = (C.c_ulonglong * n)()
        self._L.iqd_debug_stamps_ext.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        self._check(self._L.iqd_debug_stamps_ext(self._h, out, n))
        return list(out)

    def debug_stamps(self):
        out = (C.c_ulonglong * 16)()
        self._L.iqd_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self._L.iqd_debug_stamps(self._h, out))
        return list(out)

    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._check(self._L.iqd_dev_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def dev_free(self, ptr):
        self._check(self._L.iqd_dev_free(self._h, C.c_void_p(ptr)))

    def dev_upload(self, dst, host_array):
        a = np.ascontiguousarray(host_array)
        self._check(self._L.iqd_dev_upload(self._h, C.c_void_p(dst), _np_ptr(a), a.nbytes))

    def dev_download(self, src, nbytes, dtype=np.uint8):
        out = np.zeros(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        self._check(self._L.iqd_dev_download(self._h, _np_ptr(out), C.c_void_p(src), nbytes))
        return out

    def host_array(self, shape, dtype=np.uint8):
        """A page-locked numpy array (iqd_host_alloc); free it with host_free(array)."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._check(self._L.iqd_host_alloc(self._h, n, C.byref(p)))
        buf = (C.c_uint8 * n).from_address(p.value)
        a = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p.value
        return a

    def host_free(self, a):
        p = getattr(self, "_pinned", {}).pop(a.ctypes.data, None)
        if p is not None:
            self._check(self._L.iqd_host_free(self._h, C.c_void_p(p)))

    def accept_into(self, iq_u8, pcm, cnt=None, mag=None, allowed=None, first=0, n=None):
        """iqd_accept_iq on caller-owned host arrays (e.g. page-locked ones from host_array)."""
        f, n = self._range(first, n)
        bpc = iq_u8.size // n
        self._check(self._L.iqd_accept_iq(self._h, f, n, _np_ptr(iq_u8), bpc, _np_ptr(pcm), _np_ptr(cnt),
                                          _np_ptr(mag), _np_ptr(allowed)))

    def dev_tile(self, dst, period, total):
        self._check(self._L.iqd_dev_tile(self._h, C.c_void_p(dst), period, total))


class Gatherer:
    """iqd_gather_*: this rank's PCM to row `rank` of a buffer on the root, over RCCL, on the engine's stream."""

    def __init__(self, engine, unique_id, rank, world, root=0):
        self._e, self._L = engine, engine._L
        self.rank, self.world, self.root = rank, world, root
        self._L.iqd_gather_create.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        self._L.iqd_gather_pcm.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t]
        self._L.iqd_gather_destroy.argtypes = [C.c_void_p]
        self._L.iqd_gather_destroy.restype = None
        h = C.c_void_p()
        rc = self._L.iqd_gather_create(engine._h, bytes(unique_id), rank, world, root, C.byref(h))
        if rc != 0:
            raise IqdError(rc, "iqd_gather_create failed (is librccl.so there?)")
        self._h = h

    @staticmethod
    def unique_id():
        L = _lib()
        buf = (C.c_uint8 * 128)()
        L.iqd_gather_unique_id.argtypes = [C.c_void_p]
        rc = L.iqd_gather_unique_id(buf)
        if rc != 0:
            raise IqdError(rc, "iqd_gather_unique_id failed (is librccl.so there?)")
        return bytes(buf)

    def gather(self, send_dev, bytes_per_rank, recv_dev, row_stride):
        arr = (C.c_size_t * self.world)(*[int(b) for b in bytes_per_rank])
        rc = self._L.iqd_gather_pcm(self._h, C.c_void_p(send_dev), arr, C.c_void_p(recv_dev or None), int(row_stride))
        if rc != 0:
            raise IqdError(rc, "iqd_gather_pcm failed")

    def info(self):
        """{'version': RCCL's version code, 'ranks': what ncclCommCount says, 'library_reused': the process's own copy}"""
        v, n, r = C.c_int(), C.c_int(), C.c_int()
        self._L.iqd_gather_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        rc = self._L.iqd_gather_info(self._h, C.byref(v), C.byref(n), C.byref(r))
        if rc != 0:
            raise IqdError(rc, "iqd_gather_info failed")
        return {"version": v.value, "ranks": n.value, "library_reused": bool(r.value)}

    def close(self):
        if self._h:
            self._L.iqd_gather_destroy(self._h)
            self._h = None


class Resampler:
    """Block-wise Decimator / Interpolator / Interpolator_int16 for n_channels streams (include/iqdemod.h)."""
    KINDS = {"decimate_f32": 0, "interpolate_f32": 1, "interpolate_i16": 2}

    def __init__(self, engine, kind, taps, factor, n_channels=1):
        self._e, self._L = engine, engine._L
        self.kind, self.n_ch = self.KINDS[kind], n_channels
        taps = np.ascontiguousarray(taps, np.float32)
        h = C.c_void_p()
        engine._check(self._L.iqd_resampler_create(engine._h, self.kind, _np_ptr(taps), len(taps), int(factor),
                                                   n_channels, C.byref(h)))
        self._h = h
        self.dtype = np.int16 if self.kind == 2 else np.float32

    def run(self, x):
        x = np.ascontiguousarray(x, self.dtype).reshape(self.n_ch, -1)
        out = np.zeros((self.n_ch, self._L.iqd_resampler_out_count(self._h, x.shape[1])), self.dtype)
        self._e._check(self._L.iqd_resampler_run(self._h, _np_ptr(x), x.shape[1], _np_ptr(out)))
        return out

    def reset(self):
        self._e._check(self._L.iqd_resampler_reset(self._h))

    def close(self):
        if self._h:
            self._L.iqd_resampler_destroy(self._h)
            self._h = None

