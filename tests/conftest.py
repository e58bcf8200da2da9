import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("filterwarnings", "ignore:TF32 acceleration on top of oneDNN:UserWarning")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name), allow_pickle=True)


def golden_feature0(g, arch, P):
    """theta0 of a g3 golden: stored, or -- for the large compact cases, which drop it to keep the fixture small -- the
    oracle's G head of the stored z (exactly what make_golden.py fed the reference's class)."""
    import torch
    from oracle import nets_ref as N
    if g["feature0"].size:
        return g["feature0"]
    with torch.no_grad():
        return N.input_to_feature(arch, P, torch.from_numpy(g["z"])).numpy()
