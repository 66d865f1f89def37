import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")
    config.addinivalue_line("markers", "slow: minutes on the GPU box (deselect with -m 'gpu and not slow' for a quick pass)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
