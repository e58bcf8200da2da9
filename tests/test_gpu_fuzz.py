"""Seeded random shapes through the C ABI against the CPU oracle: odd sizes, k in {1,3,4,5,7}, stride 1/2, channel counts on
and off the vector paths (1, 3, 4, 24, 32, 48, 64, 96, 160), batch 1..130 -- forward, backward-data and the fused
epilogues of conv2d / conv2d_transpose.  40 cases per op; every case names its shape in the test id."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ops_ref as R

CH = [1, 3, 4, 24, 32, 48, 64, 96, 160]
N_CASES = int(os.environ.get("CGS_FUZZ_N", "40"))          # CGS_FUZZ_N=400 for a longer hunt
SEED = int(os.environ.get("CGS_FUZZ_SEED", "0"))


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).float()


def ktol(k, c):
    """2e-5 for reductions up to 1000 terms, growing with the square root of the length beyond (fp32 summation order)."""
    return 2e-5 * max(1.0, (k * k * c / 1000.0) ** 0.5)


def close(got, want, tol=2e-5):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item() + 1e-30
    assert err <= tol * ref, f"max|delta|={err:.3e} vs max|ref|={ref:.3e}"


def cases(seed, n):
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        B = int(rs.choice([1, 2, 5, 130]))
        k, s = int(rs.choice([1, 3, 4, 5, 7])), int(rs.choice([1, 2]))
        H, W = int(rs.randint(1, 19)), int(rs.randint(1, 19))
        cin, cout = int(rs.choice(CH)), int(rs.choice(CH))
        if B * H * W * max(cin, cout) > 3_000_000 or B * H * W * cin * cout * k * k > 6e9:
            continue
        out.append((B, H, W, cin, cout, k, s))
    return out


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s", cases(11 + SEED, N_CASES))
def test_conv_random_shapes(B, H, W, Cin, Cout, k, s):
    from cgs_amd import kernels as K, lib
    d = dev()
    x = rnd((B, H, W, Cin), 1).requires_grad_(True)
    w, b = rnd((k, k, Cin, Cout), 2, 0.1), rnd((Cout,), 3, 0.2)
    want = R.lrelu(R.conv2d(x, w, b, s, s))
    got = K.conv2d_fwd(x.detach().to(d), w.to(d), b.to(d), s, s, lib.EPI_LRELU)
    assert tuple(got.shape) == tuple(want.shape)
    close(got, want, ktol(k, Cin))
    y = R.conv2d(x, w, torch.zeros(Cout), s, s)
    dy = rnd(tuple(y.shape), 4)
    (y * dy).sum().backward()
    close(K.conv2d_bwd_data(dy.to(d), w.to(d), (H, W), s, s), x.grad, ktol(k, Cout))


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s", cases(23 + SEED, N_CASES))
def test_deconv_random_shapes(B, H, W, Cin, Cout, k, s):
    """conv2d_transpose from [B,H,W,Cin] to an output whose SAME conv maps back to (H, W): both admissible output sizes
    for stride 2 (2H and 2H-1) are drawn."""
    from cgs_amd import kernels as K, lib
    d = dev()
    rs = np.random.RandomState(B * 1000 + H * 37 + W)
    Ho = H * s - (int(rs.randint(0, s)) if H * s > 1 else 0)
    Wo = W * s - (int(rs.randint(0, s)) if W * s > 1 else 0)
    x = rnd((B, H, W, Cin), 1).requires_grad_(True)
    w, b = rnd((k, k, Cout, Cin), 2, 0.1), rnd((Cout,), 3, 0.2)
    want = torch.tanh(R.deconv2d(x, w, b, (B, Ho, Wo, Cout), s, s))
    got = K.deconv2d_fwd(x.detach().to(d), w.to(d), b.to(d), (Ho, Wo), s, s, lib.EPI_TANH)
    close(got, want, ktol(k, Cin))
    y = R.deconv2d(x, w, torch.zeros(Cout), (B, Ho, Wo, Cout), s, s)
    dy = rnd(tuple(y.shape), 4)
    (y * dy).sum().backward()
    close(K.deconv2d_bwd_data(dy.to(d), w.to(d), (H, W), s, s), x.grad, ktol(k, Cout))
