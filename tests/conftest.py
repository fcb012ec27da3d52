import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the native libraries exist (they travel prebuilt to the GPU box)."""
    lib = os.path.join(ROOT, "vechat_amd", "lib")
    if not (os.path.exists(os.path.join(lib, "libvechat_hip.so")) and os.path.exists(os.path.join(lib, "libvechat_host.so"))
            and os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))):
        import __graft_entry__ as g
        g.build()
    return True
