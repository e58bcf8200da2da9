"""SURVEY 8f-4 / BASELINE config 5: CycleGAN-style ResNet generator tail + PatchGAN discriminator with instance norm.
The reference has no code for this config, so parity is kernel-vs-oracle only (stated as unpinned in DESIGN.md)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nets_ref as N
from oracle import ops_ref as R
from oracle import sampling_ref as S


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).float()


def close(got, want, tol):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item() + 1e-30
    assert err <= tol * ref, f"max|delta|={err:.3e} vs max|ref|={ref:.3e}"


@pytest.mark.parametrize("shape", [(3, 8, 8, 16), (2, 32, 32, 64), (5, 7, 9, 32), (64, 4, 4, 128)])
@pytest.mark.parametrize("leak", [0.2, 0.0, 1.0])
def test_instance_norm_fwd_bwd(shape, leak):
    from cgs_amd import kernels as K
    C = shape[-1]
    x = (rnd(shape, 1) * 1.5 + 0.3).requires_grad_(True)
    sc, of = rnd((C,), 2, 0.1) + 1.0, rnd((C,), 3, 0.1)
    n = R.instance_norm(x, sc, of)
    y = torch.where(n > 0, n, leak * n)
    dy = rnd(shape, 4)
    (y * dy).sum().backward()
    d = dev()
    gy, mean, invstd = K.instnorm_lrelu_fwd(x.detach().to(d), sc.to(d), of.to(d), leak)
    close(mean, x.detach().mean((1, 2)), 1e-5)
    close(gy, y, 1e-5)
    gdx = K.instnorm_lrelu_bwd_data(dy.to(d), x.detach().to(d), sc.to(d), of.to(d), mean, invstd, leak)
    close(gdx, x.grad, 5e-5)


def test_add_kernel():
    from cgs_amd import kernels as K
    a, b = rnd((3, 5, 7, 9), 1), rnd((3, 5, 7, 9), 2)
    assert torch.equal(K.add(a.to(dev()), b.to(dev())).cpu(), a + b)


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "hipgraph"])
def test_cyclegan_patchgan_refinement_vs_oracle(use_graph):
    from cgs_amd.engine import RefineEngine
    from cgs_amd.nets import to_device
    arch, B, Ksteps = "cyclegan_tiny", 4, 3
    P = N.init_params(arch, 2019, True)
    d = dev()
    eng = RefineEngine(arch, to_device(P, d), B, d, use_graph=use_graph)
    src = (torch.rand((B, 32, 32, 3), generator=torch.Generator().manual_seed(5)) * 2 - 1)       # the image G translates
    f0 = eng.input_to_feature(src.to(d)).clone()
    with torch.no_grad():
        f0_ref = N.input_to_feature(arch, P, src)
    close(f0, f0_ref, 1e-4)                                            # encoder + residual trunk (the propose step)
    gt, dd = (lambda f: N.feature_to_data(arch, P, f)), (lambda x: N.discriminator(arch, P, x))
    lm_o, grad_o = S.forward_logits_and_grad(f0_ref, gt, dd)
    lm, grad = eng.compute_forward_logits_and_grad(f0)                 # PatchGAN: mean over the 4x4 logit map
    close(lm, lm_o, 1e-4)
    close(grad, grad_o, 2e-3)
    want = S.collaborative_refine(f0_ref, gt, dd, Ksteps, 0.1)
    for _ in range(2 if use_graph else 1):
        img, dl, ol, os_, of = eng.refine(f0, Ksteps, 0.1)
        close(dl, want[1], 1e-4)
        assert torch.equal(os_.cpu(), want[3])
        close(ol, want[2], 2e-3)
        close(of, want[4], 2e-3)
        close(img, want[0], 5e-2)
        assert torch.equal(img, eng.feature_to_data(of))
