import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cpm-r-cnn_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_ops():
    return np.load(os.path.join(ROOT, "tests", "golden", "ops.npz"))


@pytest.fixture(scope="session")
def golden_model():
    return np.load(os.path.join(ROOT, "tests", "golden", "model_r50.npz"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle
