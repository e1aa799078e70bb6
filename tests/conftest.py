import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_report_header(config):
    """the library under test, by checksum: a green run is a statement about THIS build"""
    import hashlib
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tubelet_transformer_amd", "lib", "libtuber_hip.so")
    if os.path.exists(path):
        return "libtuber_hip.so md5 %s" % hashlib.md5(open(path, "rb").read()).hexdigest()[:12]
    return "libtuber_hip.so: not built"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
