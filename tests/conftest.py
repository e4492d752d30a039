import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)     # helper modules next to the tests (emu_bind, agc_script)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import bindings
    return bindings.Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle import bindings
    if not bindings.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return bindings.Reference()


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden")
    return {f[:-4]: np.load(os.path.join(path, f)) for f in sorted(os.listdir(path)) if f.endswith(".npz")}
