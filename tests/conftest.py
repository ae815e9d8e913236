import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _backward_launch_form():
    """DSS_TEST_BACKWARD_FUSED=1|3|4|5: run the whole session with that launch form of dss_render_backward
    (DSS_OPT_BACKWARD_FUSED, include/dss_hip.h) instead of the automatic one -- every form must pass the same tests."""
    v = os.environ.get("DSS_TEST_BACKWARD_FUSED")
    if v:
        from dss_amd import _lib
        _lib.set_option(_lib.OPT_BACKWARD_FUSED, int(v))
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
