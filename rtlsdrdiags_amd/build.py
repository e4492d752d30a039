"""In-tree build of libiqdemod.so (hipcc, gfx950).  Cross-compiles without a GPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libiqdemod.so")
CSRC = os.path.join(HERE, "csrc")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    srcs.append(os.path.join(HERE, "..", "include", "iqdemod.h"))
    return any(os.path.getmtime(s) > t for s in srcs if os.path.isfile(s))


def build(force=False, verbose=False):
    """Builds rtlsdrdiags_amd/libiqdemod.so; returns its path."""
    if force or needs_build():
        cmd = ["make", "-C", CSRC] + (["-B"] if force else [])
        subprocess.run(cmd, check=True, stdout=None if verbose else subprocess.DEVNULL)
    return LIB
