import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "reference_vectors.npz")
    z = np.load(path)
    index = bytes(z["index"]).decode().split("\n")
    return z, index


@pytest.fixture(scope="session")
def codec():
    """The HIP codec on cuda:0. Fails (not skips) when the extension is missing on a GPU box."""
    # torch first: its wheel bundles its own HIP runtime, which only finds the GPU if it initialises before the
    # system runtime that libfcz_hip.so links (tests that build inputs on the device need both in one process)
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    from foldcomp_amd.codec import Codec
    c = Codec(0)
    yield c
    c.close()
