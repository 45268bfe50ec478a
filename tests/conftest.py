import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


class _Golden:
    def __init__(self, path):
        self._z = np.load(path, allow_pickle=False)

    def __getitem__(self, k):
        return self._z[k]

    def t(self, k):
        import torch
        return torch.from_numpy(np.ascontiguousarray(self._z[k]))

    def keys(self):
        return list(self._z.keys())


@pytest.fixture(scope="session")
def geo():
    return _Golden(os.path.join(GOLDEN, "geometry.npz"))


@pytest.fixture(scope="session")
def mod():
    return _Golden(os.path.join(GOLDEN, "models.npz"))


@pytest.fixture(scope="session")
def mono50():
    return _Golden(os.path.join(GOLDEN, "mono50.npz"))


@pytest.fixture(scope="session")
def pack():
    return _Golden(os.path.join(GOLDEN, "packnet.npz"))


@pytest.fixture(scope="session")
def opt():
    return _Golden(os.path.join(GOLDEN, "options.npz"))


@pytest.fixture(scope="session")
def evg():
    return _Golden(os.path.join(GOLDEN, "eval.npz"))
