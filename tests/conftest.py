import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native():
    """The ctypes binding of libbeamformer_hip.so (built in-tree by __graft_entry__.build())."""
    import __graft_entry__ as ge
    ge.build_native()
    from lib import _native
    return _native


@pytest.fixture(scope="session")
def oracle_lib():
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libdas_oracle.so"])
    import das_oracle
    return das_oracle
