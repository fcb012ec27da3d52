import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _hip_device_visible():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests need a device: on a box without one they are skipped, not failed."""
    if _hip_device_visible():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def built():
    """Make sure the native libraries exist (they travel prebuilt to the GPU box)."""
    lib = os.path.join(ROOT, "vechat_amd", "lib")
    if not (os.path.exists(os.path.join(lib, "libvechat_hip.so")) and os.path.exists(os.path.join(lib, "libvechat_host.so"))
            and os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))):
        import __graft_entry__ as g
        g.build()
    return True
