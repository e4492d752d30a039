"""ctypes binding of tests/emu/libiqd_emu.so (host twin of the GPU kernels' phase functions)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "emu", "libiqd_emu.so")


class Carry(C.Structure):
    _fields_ = [("y", C.c_float), ("u", C.c_float), ("back", C.c_int32), ("y_end", C.c_float),
                ("u_end", C.c_float), ("pad", C.c_uint32 * 3)]


def lib():
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu")], check=True)
    L = C.CDLL(os.environ.get("IQD_EMU_LIB", SO))   # IQD_EMU_LIB: e.g. the sanitizer build (tests/emu/Makefile)
    L.emu_wbfm_accept.restype = C.c_int
    L.emu_wbfm_accept.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_float,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
    L.emu_wbfm_reset.argtypes = [C.c_void_p, C.c_void_p]
    L.emu_lds_bytes.restype = C.c_uint32
    L.emu_chain_accept.restype = None
    L.emu_chain_accept.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                   C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


class WbfmChannel:
    """One WBFM channel driven through the emulated kernel, call after call."""

    def __init__(self, L, tile_len, block_samples=16384, rotation=1, gain=None, guess_scale=1.0):
        self.L, self.tile_len, self.block_samples = L, tile_len, block_samples
        self.rotation = rotation
        self.gain = np.float32(256000 / (2 * np.pi)) if gain is None else np.float32(gain)
        self.guess_scale = guess_scale
        self.tail = np.full(4096, 128, np.uint8)
        self.carry = Carry()
        self.hand_off_mismatches = 0
        self.segment_repairs = 0

    def accept(self, u8):
        u8 = np.ascontiguousarray(u8, dtype=np.uint8)
        m = len(u8) // 2
        pcm = np.zeros(m // 32, np.int16)
        mag = np.zeros(max(1, (m + self.block_samples - 1) // self.block_samples), np.uint32)
        rep = C.c_uint32(0)
        self.hand_off_mismatches += self.L.emu_wbfm_accept(
            u8.ctypes.data, m, self.tile_len, self.block_samples, self.rotation, self.gain,
            self.tail.ctypes.data, C.byref(self.carry), pcm.ctypes.data, mag.ctypes.data,
            C.byref(rep), self.guess_scale)
        self.segment_repairs += rep.value
        return pcm, mag

    def reset(self):
        self.L.emu_wbfm_reset(self.tail.ctypes.data, C.byref(self.carry))


FAMILY = {"am": 0, "fm": 1, "lsb": 3, "usb": 3}
DEFAULT_GAIN = {"am": 300.0, "fm": 64000 / (2 * np.pi), "lsb": 300.0, "usb": 300.0}


class FirChannel:
    """One FM / AM / SSB channel driven through the emulated tile kernel + DC-removal pass."""

    def __init__(self, L, mode, tile_len, block_samples=16384, rotation=1, gain=None):
        self.L, self.mode, self.tile_len, self.block_samples = L, mode, tile_len, block_samples
        self.rotation = rotation
        self.gain = np.float32(DEFAULT_GAIN[mode] if gain is None else gain)
        self.tail = np.full(4096, 128, np.uint8)
        self.dc = np.zeros(2, np.float32)

    def accept(self, u8):
        u8 = np.ascontiguousarray(u8, dtype=np.uint8)
        m = len(u8) // 2
        pcm = np.zeros(m // 32, np.int16)
        mag = np.zeros(max(1, (m + self.block_samples - 1) // self.block_samples), np.uint32)
        base = np.zeros(m // 32 + 1, np.int32)
        self.L.emu_chain_accept(FAMILY[self.mode], 1 if self.mode == "lsb" else 0, u8.ctypes.data, m,
                                self.tile_len, self.block_samples, self.rotation, self.gain,
                                self.tail.ctypes.data, self.dc.ctypes.data, pcm.ctypes.data,
                                mag.ctypes.data, base.ctypes.data)
        return pcm, mag

    def reset(self):
        self.tail[:] = 128
        self.dc[:] = 0
