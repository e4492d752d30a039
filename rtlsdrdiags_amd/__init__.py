"""rtlsdrdiags_amd — MI355X-native IQ demodulation engine (drop-in for the RtlSdrDiags DSP hot path).

The product is the C-ABI shared library ``libiqdemod.so`` (see include/iqdemod.h) built from the
hand-written gfx950 kernels under ``csrc/``.  This package holds

  * ``capi``   — ctypes binding of that C ABI (what tests and bench.py drive),
  * ``build``  — in-tree build of the library with hipcc (no GPU needed to build),
  * ``synth``  — seeded synthetic IQ sources standing in for librtlsdr,
  * ``shard``  — channel sharding across the GPUs of one node (torch.distributed / RCCL).

There is deliberately no CPU data path here: ``capi.Engine`` raises if the library or a HIP
device is missing.
"""
from . import synth  # noqa: F401

__all__ = ["synth"]
