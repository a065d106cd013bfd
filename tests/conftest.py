import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a machine without an MI355X (or before the HIP library is built) skips the GPU tests
    instead of failing in them; `-m gpu` on the GPU box runs them all (nothing is skipped there)."""
    import torch
    lib = os.path.join(ROOT, "insmos_amd", "libinsmos_hip.so")
    reason = None
    if not torch.cuda.is_available():
        reason = "no HIP GPU in this process (torch.cuda.is_available() is False)"
    elif not os.path.exists(lib):
        reason = "insmos_amd/libinsmos_hip.so is not built (python __graft_entry__.py)"
    if reason is None:
        return
    skip = pytest.mark.skip(reason=reason)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
