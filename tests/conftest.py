import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The loaded libcoslam_hip.so with a visible device.  Fails (not skips) when the extension is
    missing: GPU tests must never pass on a silent fallback."""
    import coslam_amd

    lib = coslam_amd.lib()
    assert lib.cs_device_count() >= 1, "no HIP device visible"
    return lib
