"""oracle/bindings.py — TEST INFRASTRUCTURE ONLY.

ctypes bindings for
  * oracle/libiqd_oracle.so  — our plain-C restatement (iqd_oracle.c), and
  * oracle/_ref/libiqd_ref.so — the unmodified reference sources + ref_shim.cc
    (only where it was built; it cannot be rebuilt on the GPU box).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libiqd_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libiqd_ref.so")            # the reference's hot path (no AGC, no scanner, no Radio test double)
REF_AGC_SO = os.path.join(HERE, "_ref", "libiqd_ref_agc.so")    # ... plus AutomaticGainControl.cc, FrequencyScanner.cc and their owner's double

MODES = {"none": 0, "am": 1, "fm": 2, "wbfm": 3, "lsb": 4, "usb": 5}
TAP_SETS = ["wbfm_pre", "wbfm_d1", "wbfm_d2", "audio40", "fm_tuner", "fm_post",
            "am_s1", "am_s2", "am_s3", "ssb_delay", "ssb_hilbert"]


def build(ref=True):
    """(Re)build the oracle library and, where /root/reference exists, oracle/_ref."""
    subprocess.check_call(["make", "-s", "-C", HERE, "libiqd_oracle.so"])
    if ref and os.path.isdir("/root/reference/radioDiags"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class _Chain:
    """Common shape of the two chain wrappers."""

    def __init__(self, lib, prefix):
        self._lib, self._p = lib, prefix
        self._h = C.c_void_p(getattr(lib, prefix + "create")())

    def close(self):
        if self._h:
            getattr(self._lib, self._p + "destroy")(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_mode(self, mode):
        getattr(self._lib, self._p + "set_mode")(self._h, int(MODES.get(mode, mode)))

    def set_gain(self, which, gain):
        getattr(self._lib, self._p + "set_gain")(self._h, int(which), C.c_float(gain))

    def set_squelch(self, threshold):
        getattr(self._lib, self._p + "set_squelch")(self._h, int(threshold))

    def reset(self):
        getattr(self._lib, self._p + "reset")(self._h)

    def accept_stream(self, iq_u8, block_bytes=32768):
        """Feeds a uint8 IQ array block by block.  Returns (pcm, magnitude, allowed)."""
        iq_u8 = np.ascontiguousarray(iq_u8, dtype=np.uint8)
        nblk = (len(iq_u8) + block_bytes - 1) // block_bytes
        pcm = np.zeros(len(iq_u8) // 64 + 64, dtype=np.int16)
        mag = np.zeros(nblk, dtype=np.uint32)
        allowed = np.zeros(nblk, dtype=np.uint8)
        n = getattr(self._lib, self._p + "accept_stream")(
            self._h, _ptr(iq_u8), len(iq_u8), block_bytes, _ptr(pcm), len(pcm),
            _ptr(mag), _ptr(allowed))
        if n < 0:
            raise ValueError("accept_stream rejected the input (%d)" % n)
        return pcm[:n].copy(), mag, allowed

    def demod_accept(self, mode, iq_s8):
        iq_s8 = np.array(iq_s8, dtype=np.int8, copy=True)
        pcm = np.zeros(len(iq_s8) // 64 + 64, dtype=np.int16)
        n = getattr(self._lib, self._p + "demod_accept")(
            self._h, int(MODES.get(mode, mode)), _ptr(iq_s8), len(iq_s8), _ptr(pcm), len(pcm))
        return pcm[:n].copy()


def _sig(fn, restype, argtypes):
    fn.restype, fn.argtypes = restype, argtypes


class _Scanner(C.Structure):
    _fields_ = [("start_hz", C.c_uint64), ("end_hz", C.c_uint64), ("increment_hz", C.c_uint64),
                ("current_hz", C.c_uint64), ("new_configuration", C.c_int), ("scanning", C.c_int),
                ("tuned_hz", C.c_uint64), ("tune_count", C.c_uint32)]


class Oracle:
    """libiqd_oracle.so"""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        L = self.lib = C.CDLL(ORACLE_SO)
        vp, sz = C.c_void_p, C.c_size_t
        _sig(L.iqo_create, vp, [])
        _sig(L.iqo_destroy, None, [vp])
        _sig(L.iqo_reset, None, [vp])
        _sig(L.iqo_reset_demod, None, [vp, C.c_int])
        _sig(L.iqo_set_mode, None, [vp, C.c_int])
        _sig(L.iqo_set_gain, None, [vp, C.c_int, C.c_float])
        _sig(L.iqo_set_squelch, None, [vp, C.c_int32])
        _sig(L.iqo_set_rx_gain_db, None, [vp, C.c_uint32])
        _sig(L.iqo_set_rotation, None, [vp, C.c_int])
        _sig(L.iqo_agc_set_type, C.c_int, [vp, C.c_uint32])
        _sig(L.iqo_agc_set_deadband, C.c_int, [vp, C.c_uint32])
        _sig(L.iqo_agc_set_blanking_limit, C.c_int, [vp, C.c_uint32])
        _sig(L.iqo_agc_set_operating_point, None, [vp, C.c_int32])
        _sig(L.iqo_agc_set_filter_coefficient, C.c_int, [vp, C.c_float])
        _sig(L.iqo_agc_enable, C.c_int, [vp, C.c_int])
        _sig(L.iqo_agc_feed, None, [vp, C.c_uint32])
        _sig(L.iqo_get_rx_gain_db, C.c_uint32, [vp])
        _sig(L.iqo_scanner_of, C.POINTER(_Scanner), [vp])
        _sig(L.iqo_scanner_set_parameters, C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_uint64])
        _sig(L.iqo_scanner_start, C.c_int, [vp])
        _sig(L.iqo_scanner_stop, C.c_int, [vp])
        _sig(L.iqo_accept_stream, C.c_long, [vp, vp, sz, sz, vp, sz, vp, vp])
        _sig(L.iqo_demod_accept, C.c_long, [vp, C.c_int, vp, sz, vp, sz])
        _sig(L.iqo_quantize_taps, None, [vp, C.c_int, vp])
        _sig(L.iqo_decimate_q15, C.c_long, [vp, C.c_int, C.c_int, vp, sz, vp])
        _sig(L.iqo_fir_q15, None, [vp, C.c_int, vp, sz, vp])
        _sig(L.iqo_fir_f32, None, [vp, C.c_int, vp, sz, vp])
        _sig(L.iqo_iir_f32, None, [vp, C.c_int, vp, C.c_int, vp, sz, vp])
        _sig(L.iqo_rotate, None, [vp, sz, C.c_int])
        _sig(L.iqo_decimate_f32, C.c_long, [vp, C.c_int, C.c_int, vp, sz, vp])
        _sig(L.iqo_interpolate_f32, None, [vp, C.c_int, C.c_int, vp, sz, vp])
        _sig(L.iqo_interpolate_q15, None, [vp, C.c_int, C.c_int, vp, sz, vp])
        _sig(L.iqo_atan2_lut, None, [vp])
        _sig(L.iqo_fm_theta_lut, None, [vp, C.c_int])
        _sig(L.iqo_db_table, None, [vp])
        _sig(L.iqo_dbfs, C.c_int32, [C.c_uint32])
        _sig(L.iqo_cast_i16, C.c_int16, [C.c_float])
        _sig(L.iqo_block_magnitude, C.c_uint32, [vp, sz])
        _sig(L.iqo_get_taps_q15, C.c_int, [C.c_int, vp])
        _sig(L.iqo_get_taps_f32, C.c_int, [C.c_int, vp])
        _sig(L.iqo_wbfm_stages, None, [vp, sz, C.c_float, vp, vp, vp, vp, vp, vp])

    def chain(self):
        L = self.lib
        c = _Chain(L, "iqo_")
        c.set_rx_gain_db = lambda g: L.iqo_set_rx_gain_db(c._h, int(g))
        c.set_rotation = lambda r: L.iqo_set_rotation(c._h, int(r))
        # AutomaticGainControl: the setters return the reference's success flag
        c.agc_set_type = lambda t: bool(L.iqo_agc_set_type(c._h, int(t)))
        c.agc_set_deadband = lambda d: bool(L.iqo_agc_set_deadband(c._h, int(d)))
        c.agc_set_blanking_limit = lambda n: bool(L.iqo_agc_set_blanking_limit(c._h, int(n)))
        c.agc_set_operating_point = lambda p: L.iqo_agc_set_operating_point(c._h, int(p))
        c.agc_set_filter_coefficient = lambda a: bool(L.iqo_agc_set_filter_coefficient(c._h, C.c_float(a)))
        c.agc_enable = lambda on=True: bool(L.iqo_agc_enable(c._h, 1 if on else 0))
        c.agc_feed = lambda m: L.iqo_agc_feed(c._h, int(m))
        c.rx_gain_db = lambda: int(L.iqo_get_rx_gain_db(c._h))
        c.reset_demod = lambda which: L.iqo_reset_demod(c._h, int(which))
        # FrequencyScanner
        c.scanner_set_parameters = lambda a, b, inc: bool(L.iqo_scanner_set_parameters(c._h, int(a), int(b), int(inc)))
        c.scanner_start = lambda: bool(L.iqo_scanner_start(c._h))
        c.scanner_stop = lambda: bool(L.iqo_scanner_stop(c._h))

        def tuned():
            s = L.iqo_scanner_of(c._h).contents
            return int(s.tuned_hz), int(s.tune_count)
        c.scanner_tuned = tuned
        return c

    # ---- primitives ----
    def taps_q15(self, name):
        hq = np.zeros(40, dtype=np.int16)
        n = self.lib.iqo_get_taps_q15(TAP_SETS.index(name), _ptr(hq))
        return hq[:n].copy()

    def taps_f32(self, name):
        h = np.zeros(40, dtype=np.float32)
        n = self.lib.iqo_get_taps_f32(TAP_SETS.index(name), _ptr(h))
        return h[:n].copy()

    def decimate_q15(self, taps, factor, x):
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.int16)
        out = np.zeros(len(x) // factor + 1, dtype=np.int16)
        n = self.lib.iqo_decimate_q15(_ptr(taps), len(taps), factor, _ptr(x), len(x), _ptr(out))
        return out[:n].copy()

    def fir_f32(self, taps, x):
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros(len(x), dtype=np.float32)
        self.lib.iqo_fir_f32(_ptr(taps), len(taps), _ptr(x), len(x), _ptr(out))
        return out

    def iir_f32(self, b, a, x):
        b = np.ascontiguousarray(b, dtype=np.float32)
        a = np.ascontiguousarray(a, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros(len(x), dtype=np.float32)
        self.lib.iqo_iir_f32(_ptr(b), len(b), _ptr(a), len(a), _ptr(x), len(x), _ptr(out))
        return out

    def decimate_f32(self, taps, factor, x):
        taps, x = np.ascontiguousarray(taps, np.float32), np.ascontiguousarray(x, np.float32)
        out = np.zeros(len(x) // factor + 1, np.float32)
        n = self.lib.iqo_decimate_f32(_ptr(taps), len(taps), factor, _ptr(x), len(x), _ptr(out))
        return out[:n].copy()

    def interpolate_f32(self, taps, factor, x):
        taps, x = np.ascontiguousarray(taps, np.float32), np.ascontiguousarray(x, np.float32)
        out = np.zeros(len(x) * factor, np.float32)
        self.lib.iqo_interpolate_f32(_ptr(taps), len(taps), factor, _ptr(x), len(x), _ptr(out))
        return out

    def interpolate_q15(self, taps, factor, x):
        taps, x = np.ascontiguousarray(taps, np.float32), np.ascontiguousarray(x, np.int16)
        out = np.zeros(len(x) * factor, np.int16)
        self.lib.iqo_interpolate_q15(_ptr(taps), len(taps), factor, _ptr(x), len(x), _ptr(out))
        return out

    def rotate(self, s8, rotation):
        s8 = np.array(s8, dtype=np.int8, copy=True)
        self.lib.iqo_rotate(_ptr(s8), len(s8), int(rotation))
        return s8

    def atan2_lut(self):
        lut = np.zeros((256, 256), dtype=np.float32)
        self.lib.iqo_atan2_lut(_ptr(lut))
        return lut

    def fm_theta_lut(self, half_range):
        w = 2 * half_range + 1
        lut = np.zeros((w, w), dtype=np.float32)
        self.lib.iqo_fm_theta_lut(_ptr(lut), int(half_range))
        return lut

    def db_table(self):
        t = np.zeros(257, dtype=np.int32)
        self.lib.iqo_db_table(_ptr(t))
        return t

    def dbfs(self, magnitude):
        return int(self.lib.iqo_dbfs(int(magnitude)))

    def cast_i16(self, f):
        return int(self.lib.iqo_cast_i16(C.c_float(f)))

    def block_magnitude(self, s8):
        s8 = np.ascontiguousarray(s8, dtype=np.int8)
        return int(self.lib.iqo_block_magnitude(_ptr(s8), len(s8)))

    def wbfm_stages(self, rot_s8, gain=None):
        rot_s8 = np.ascontiguousarray(rot_s8, dtype=np.int8)
        n = len(rot_s8) // 2
        if gain is None:
            gain = np.float32(256000 / (2 * np.pi))
        out = dict(ip=np.zeros(n, np.int8), qp=np.zeros(n, np.int8),
                   theta=np.zeros(n, np.float32), dtheta=np.zeros(n, np.float32),
                   deemph=np.zeros(n, np.float32), w=np.zeros(n, np.int16))
        self.lib.iqo_wbfm_stages(_ptr(rot_s8), n, C.c_float(gain), _ptr(out["ip"]),
                                 _ptr(out["qp"]), _ptr(out["theta"]), _ptr(out["dtheta"]),
                                 _ptr(out["deemph"]), _ptr(out["w"]))
        return out


def have_ref():
    return os.path.exists(REF_SO)


class Reference:
    """oracle/_ref/libiqd_ref.so — the reference itself (where it was built)."""

    def __init__(self):
        if not have_ref():
            raise FileNotFoundError(REF_SO)
        L = self.lib = C.CDLL(REF_SO)
        vp, sz = C.c_void_p, C.c_size_t
        _sig(L.ref_create, vp, [])
        _sig(L.ref_destroy, None, [vp])
        _sig(L.ref_reset, None, [vp])
        _sig(L.ref_set_mode, None, [vp, C.c_int])
        _sig(L.ref_set_gain, None, [vp, C.c_int, C.c_float])
        _sig(L.ref_set_squelch, None, [vp, C.c_int32])
        _sig(L.ref_set_rx_gain_db, None, [C.c_uint32])
        _sig(L.ref_accept_stream, C.c_long, [vp, vp, sz, sz, vp, sz, vp, vp])
        _sig(L.ref_demod_accept, C.c_long, [vp, C.c_int, vp, C.c_uint32, vp, sz])
        _sig(L.ref_upconvert, None, [vp, vp, C.c_uint32])
        _sig(L.ref_downconvert, None, [vp, vp, C.c_uint32])
        _sig(L.ref_decimator_int16, C.c_long, [vp, C.c_int, C.c_int, vp, sz, vp])
        _sig(L.ref_fir_int16, None, [vp, C.c_int, vp, sz, vp])
        _sig(L.ref_fir_f32, None, [vp, C.c_int, vp, sz, vp])
        _sig(L.ref_iir_f32, None, [vp, C.c_int, vp, C.c_int, vp, sz, vp])
        _sig(L.ref_decimator_f32, C.c_long, [vp, C.c_int, C.c_int, vp, sz, vp])
        _sig(L.ref_interpolator_f32, None, [vp, C.c_int, C.c_int, vp, sz, vp])
        _sig(L.ref_interpolator_int16, None, [vp, C.c_int, C.c_int, vp, sz, vp])
        _sig(L.ref_squelch_create, vp, [C.c_int32])
        _sig(L.ref_squelch_destroy, None, [vp])
        _sig(L.ref_squelch_run, C.c_int, [vp, C.c_uint32, vp, C.c_uint32, vp])
        _sig(L.ref_dbfs, C.c_int32, [C.c_uint32])
        self._agc_lib = None

    def _with_owner(self):
        """The second library (oracle/Makefile: -DIQD_REF_WITH_OWNER): the same chain with the reference's AGC and scanner and the
        harness' test double of their owner.  Loaded only by the (f)-2 / (f)-3 pins; the hot-path pin never maps it."""
        if self._agc_lib is not None:
            return self._agc_lib
        if not os.path.exists(REF_AGC_SO):
            raise FileNotFoundError(REF_AGC_SO)
        L = self._agc_lib = C.CDLL(REF_AGC_SO)
        vp, sz = C.c_void_p, C.c_size_t
        _sig(L.ref_create, vp, [])
        _sig(L.ref_destroy, None, [vp])
        _sig(L.ref_reset, None, [vp])
        _sig(L.ref_set_mode, None, [vp, C.c_int])
        _sig(L.ref_set_gain, None, [vp, C.c_int, C.c_float])
        _sig(L.ref_set_squelch, None, [vp, C.c_int32])
        _sig(L.ref_set_rx_gain_db, None, [C.c_uint32])
        _sig(L.ref_accept_stream, C.c_long, [vp, vp, sz, sz, vp, sz, vp, vp])
        _sig(L.ref_demod_accept, C.c_long, [vp, C.c_int, vp, C.c_uint32, vp, sz])
        _sig(L.ref_agc_attach, None, [vp, C.c_int32])
        _sig(L.ref_agc_set, C.c_int, [vp, C.c_int, C.c_float])
        _sig(L.ref_agc_run, None, [vp, C.c_uint32])
        _sig(L.ref_agc_if_gain, C.c_uint32, [vp])
        _sig(L.ref_scanner_attach, None, [vp])
        _sig(L.ref_scanner_cmd, C.c_int, [vp, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64])
        _sig(L.ref_scanner_frequency, C.c_uint64, [vp, C.POINTER(C.c_uint32)])
        return L

    def chain(self, agc=False, operating_point=-12, scanner=False):
        """agc=True attaches the reference's AutomaticGainControl the way Radio.cc:184 does; the chain then has
        its own IF gain (the harness' Radio double) instead of the process-global one, and per-block magnitudes
        read 0xffffffff (the AGC owns the single magnitude-callback slot).  scanner=True attaches the reference's
        FrequencyScanner (it owns the signal-state slot: `allowed` then reads 0xff)."""
        L = self._with_owner() if (agc or scanner) else self.lib
        c = _Chain(L, "ref_")
        c.set_rx_gain_db = lambda g: L.ref_set_rx_gain_db(int(g))  # a process global
        if agc:
            L.ref_agc_attach(c._h, int(operating_point))
            c.set_rx_gain_db = lambda g: bool(L.ref_agc_set(c._h, 6, C.c_float(g)))
            c.agc_set_type = lambda t: bool(L.ref_agc_set(c._h, 0, C.c_float(t)))
            c.agc_set_deadband = lambda d: bool(L.ref_agc_set(c._h, 1, C.c_float(d)))
            c.agc_set_blanking_limit = lambda n: bool(L.ref_agc_set(c._h, 2, C.c_float(n)))
            c.agc_set_filter_coefficient = lambda a: bool(L.ref_agc_set(c._h, 3, C.c_float(a)))
            c.agc_set_operating_point = lambda p: L.ref_agc_set(c._h, 4, C.c_float(p)) and None
            c.agc_enable = lambda on=True: bool(L.ref_agc_set(c._h, 5, C.c_float(1.0 if on else 0.0)))
            c.agc_feed = lambda m: L.ref_agc_run(c._h, int(m))
            c.rx_gain_db = lambda: int(L.ref_agc_if_gain(c._h))
        if scanner:
            L.ref_scanner_attach(c._h)
            c.scanner_set_parameters = lambda a, b, inc: bool(L.ref_scanner_cmd(c._h, 0, int(a), int(b), int(inc)))
            c.scanner_start = lambda: bool(L.ref_scanner_cmd(c._h, 1, 0, 0, 0))
            c.scanner_stop = lambda: bool(L.ref_scanner_cmd(c._h, 2, 0, 0, 0))

            def tuned():
                n = C.c_uint32()
                f = L.ref_scanner_frequency(c._h, C.byref(n))
                return int(f), int(n.value)
            c.scanner_tuned = tuned
        return c

    def decimate_q15(self, taps, factor, x):
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.int16)
        out = np.zeros(len(x) // factor + 1, dtype=np.int16)
        n = self.lib.ref_decimator_int16(_ptr(taps), len(taps), factor, _ptr(x), len(x), _ptr(out))
        return out[:n].copy()

    def decimate_f32(self, taps, factor, x):
        taps, x = np.ascontiguousarray(taps, np.float32), np.ascontiguousarray(x, np.float32)
        out = np.zeros(len(x) // factor + 1, np.float32)
        n = self.lib.ref_decimator_f32(_ptr(taps), len(taps), factor, _ptr(x), len(x), _ptr(out))
        return out[:n].copy()

    def interpolate_f32(self, taps, factor, x):
        taps, x = np.ascontiguousarray(taps, np.float32), np.ascontiguousarray(x, np.float32)
        out = np.zeros(len(x) * factor, np.float32)
        self.lib.ref_interpolator_f32(_ptr(taps), len(taps), factor, _ptr(x), len(x), _ptr(out))
        return out

    def interpolate_q15(self, taps, factor, x):
        taps, x = np.ascontiguousarray(taps, np.float32), np.ascontiguousarray(x, np.int16)
        out = np.zeros(len(x) * factor, np.int16)
        self.lib.ref_interpolator_int16(_ptr(taps), len(taps), factor, _ptr(x), len(x), _ptr(out))
        return out

    def fir_q15(self, taps, x):
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.int16)
        out = np.zeros(len(x), dtype=np.int16)
        self.lib.ref_fir_int16(_ptr(taps), len(taps), _ptr(x), len(x), _ptr(out))
        return out

    def fir_f32(self, taps, x):
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros(len(x), dtype=np.float32)
        self.lib.ref_fir_f32(_ptr(taps), len(taps), _ptr(x), len(x), _ptr(out))
        return out

    def iir_f32(self, b, a, x):
        b = np.ascontiguousarray(b, dtype=np.float32)
        a = np.ascontiguousarray(a, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros(len(x), dtype=np.float32)
        self.lib.ref_iir_f32(_ptr(b), len(b), _ptr(a), len(a), _ptr(x), len(x), _ptr(out))
        return out

    def rotate(self, s8, rotation):
        s8 = np.array(s8, dtype=np.int8, copy=True)
        h = C.c_void_p(self.lib.ref_create())
        if rotation > 0:
            self.lib.ref_upconvert(h, _ptr(s8), len(s8))
        elif rotation < 0:
            self.lib.ref_downconvert(h, _ptr(s8), len(s8))
        self.lib.ref_destroy(h)
        return s8

    def squelch_sequence(self, threshold, gain_db, blocks_s8):
        sq = C.c_void_p(self.lib.ref_squelch_create(int(threshold)))
        out = []
        for b in blocks_s8:
            b = np.ascontiguousarray(b, dtype=np.int8)
            mag = C.c_uint32(0)
            allowed = self.lib.ref_squelch_run(sq, int(gain_db), _ptr(b), len(b), C.byref(mag))
            out.append((int(allowed), int(mag.value)))
        self.lib.ref_squelch_destroy(sq)
        return out

    def dbfs(self, magnitude):
        return int(self.lib.ref_dbfs(int(magnitude)))
