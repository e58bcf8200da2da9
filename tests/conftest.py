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


def bx6_serves(Cin, Cout):
    """Does the split-bf16 implicit GEMM (csrc/igemm_bx6.hip) serve a conv / deconv layer with these channel counts in at least
    one direction?  Forward reduces over Cin into Cout, backward-data over Cout into Cin: a reduction over whole 32-channel chunks
    into whole 64-column wave tiles."""
    return (Cin % 32 == 0 and Cout % 64 == 0) or (Cout % 32 == 0 and Cin % 64 == 0)


@pytest.fixture(params=["f32", "bx6"])
def contraction(request):
    """Runs a kernel-level test once per contraction arithmetic of the library (include/cgs_hip.h, cgs_set_contraction): the exact
    fp32 MFMA default and the opt-in split-bf16 form -- forced for every call whose geometry it can serve ("bx6_all"), so the small
    test shapes reach it; shape-parametrized cases it cannot serve are skipped in that mode (they would repeat the f32 run).
    The SAME assertions at the SAME tolerances hold in both modes at the operator level; the two K-step bars that measure chaotic
    amplification of rounding (the goldens' image-drift sanity bound, 60x instead of 25x the trajectory tolerance; the K = 50 trajectory of
    test_gpu_fullsize) scale with the measured error ratio and say so where they do."""
    from cgs_amd import kernels as K
    mode = request.param
    if mode == "bx6":
        ps = getattr(getattr(request.node, "callspec", None), "params", {})
        if "Cin" in ps and "Cout" in ps and not bx6_serves(ps["Cin"], ps["Cout"]):
            pytest.skip("no direction of this shape is served by the split-bf16 kernel")
    K.set_contraction("bx6_all" if mode == "bx6" else "f32")
    try:
        yield mode
    finally:
        K.set_contraction("f32")


@pytest.fixture(autouse=True)
def _default_contraction(request):
    """Engines put their own contraction mode in force when they run; no test inherits another one's."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        from cgs_amd import kernels as K
        K.set_contraction("f32")
