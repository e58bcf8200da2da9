"""Seeded random network topologies through the fused engine against the CPU oracle: generator tails and discriminators
drawn from the layer vocabulary (deconv / conv with per-layer kernel and stride, batch norm, instance norm, residual
blocks, relu / lrelu / tanh, fc or PatchGAN logit heads).  Exercises the engine's stage compiler and its fusion rules
(forward epilogues, backward activation gradients folded into the next contraction) on graphs nobody hand-picked."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nets_ref as N
from oracle import sampling_ref as S

N_ARCHS = int(os.environ.get("CGS_FUZZ_ARCHS", "8"))
SEED = int(os.environ.get("CGS_FUZZ_SEED", "0"))


@pytest.fixture(autouse=True)
def _plain_cpu_convolutions():
    """The checker runs on torch-CPU; its oneDNN convolution backward corrupts the heap on some degenerate shapes (1x1 kernel,
    stride 2, one input channel: 'double free or corruption' in this image's torch 2.10), which a random-shape hunt does reach.
    The native CPU kernels are slower and fine."""
    with torch.backends.mkldnn.flags(enabled=False):
        yield


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def random_arch(seed):
    rs = np.random.RandomState(seed)
    pick = lambda xs: xs[int(rs.randint(len(xs)))]
    h = pick([4, 6, 8])
    c = pick([16, 32, 64])
    feat = (h, h, c)
    norm_g = pick(["bn", "instnorm"])
    tail, shape, idx = [], feat, 0
    if rs.rand() < 0.4:                                      # a residual block in front (CycleGAN trunk style)
        tail.append(("res", [("conv", "g_r_c1", c, 3, 1), ("instnorm", "g_r_n1"), ("relu",),
                             ("conv", "g_r_c2", c, 3, 1), ("instnorm", "g_r_n2")]))
    n_up = pick([1, 2])
    img_c = pick([1, 3])
    for u in range(n_up):
        last = u == n_up - 1
        co = img_c if last and rs.rand() < 0.6 else pick([16, 32])
        k = pick([3, 4, 5])
        shape = (shape[0] * 2, shape[1] * 2, co)
        tail.append(("deconv", f"g_up{u}", shape, k, 2))
        if not (last and co == img_c):
            tail += [(norm_g, f"g_n{u}"), ("relu",)]
            idx += 1
    if shape[2] != img_c:                                    # an RGB / grey head: stride-1 conv
        k = pick([3, 5, 7])
        tail.append(("conv", "g_head_rgb", img_c, k, 1))
        shape = (shape[0], shape[1], img_c)
    tail.append(("tanh",))
    img = shape
    norm_d = pick(["bn", "instnorm"])
    d, ds = [], img
    n_d = pick([1, 2, 3])
    for i in range(n_d):
        co = pick([16, 32, 64])
        k, s = pick([3, 4, 5]), 2
        d.append(("conv", f"d_c{i}", co, k, s))
        ds = (-(-ds[0] // s), -(-ds[1] // s), co)
        if i > 0:
            d.append((norm_d, f"d_n{i}"))
        d.append(("lrelu",))
    if rs.rand() < 0.5 and ds[0] > 1:
        d.append(("conv", "d_patch", 1, pick([3, 4]), 1))     # PatchGAN logit map
    else:
        d += [("flatten",), ("linear", "d_fc", 1)]
    return dict(z_dim=8, img=img, k=5, stride=2, feature=feat,
                g_head=[("linear", "g_fc0", h * h * c), ("reshape", feat), ("bn", "g_bn0"), ("relu",)],
                g_tail=tail, d=d)


def close(got, want, tol, what, floor=1e-6):
    """Relative to the largest reference value, with an absolute floor for degenerate draws whose outputs are ~1e-5."""
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item() + 1e-30
    assert err <= max(tol * ref, floor), f"{what}: max|delta|={err:.3e} vs max|ref|={ref:.3e}"


def kink_margins(name, P, f0):
    """Per sample: how close the oracle's forward pass (G tail + D) comes to a relu / lrelu kink, as min |pre-activation| over
    the largest |pre-activation| of that tensor.  A sample whose margin is within the forward rounding error can legitimately
    take the other slope in a second arithmetic, and its WHOLE gradient then differs (seed 1002: an instance-norm output of
    3.3e-7 in one of 8 samples)."""
    from oracle import ops_ref as R
    margins = []
    relu0, lrelu0 = torch.relu, R.lrelu

    def tap(fn):
        def f(x, *a, **k):
            if x.dim() >= 2:
                flat = x.detach().reshape(x.shape[0], -1).abs().double()
                margins.append(flat.min(dim=1).values / (flat.max() + 1e-30))
            return fn(x, *a, **k)
        return f
    torch.relu, R.lrelu = tap(relu0), tap(lrelu0)
    try:
        with torch.no_grad():
            N.discriminator(name, P, N.feature_to_data(name, P, f0))
    finally:
        torch.relu, R.lrelu = relu0, lrelu0
    return torch.stack(margins).min(dim=0).values if margins else torch.ones(f0.shape[0], dtype=torch.float64)


@pytest.mark.parametrize("seed", [1000 + SEED + i for i in range(N_ARCHS)])
@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "hipgraph"])
def test_random_topology_matches_oracle(seed, use_graph):
    from cgs_amd import nets
    from cgs_amd.engine import RefineEngine
    A = random_arch(seed)
    name = f"fuzz{seed}"
    N.ARCHS[name] = A
    nets.ARCHS[name] = A
    try:
        B, Ksteps = int(np.random.RandomState(seed).choice([3, 8, 16])), 2
        P = N.init_params(name, 7, True)
        d = dev()
        eng = RefineEngine(name, nets.to_device(P, d), B, d, use_graph=use_graph)
        z = torch.from_numpy(np.random.RandomState(seed + 1).uniform(-1, 1, (B, 8)).astype(np.float32))
        f0 = eng.input_to_feature(z.to(d)).clone()
        with torch.no_grad():
            f0_ref = N.input_to_feature(name, P, z)
        close(f0, f0_ref, 1e-4, "feature0")
        gt, dd = (lambda f: N.feature_to_data(name, P, f)), (lambda x: N.discriminator(name, P, x))
        lm_o, grad_o = S.forward_logits_and_grad(f0_ref, gt, dd)
        lm, grad = eng.compute_forward_logits_and_grad(f0)
        close(lm, lm_o, 2e-4, "mean logit")
        # a pre-activation within rounding of a relu / lrelu kink can take the other slope in the two arithmetics: require the
        # bulk of the gradient entries to agree tightly instead of the maximum
        g, go = grad.cpu().double(), grad_o.double()
        keep = torch.ones(B, dtype=torch.bool)       # samples whose refinement is compared below (all but the kink-excused ones)
        if go.abs().max().item() < 1e-12:          # degenerate draw (e.g. instance norm over a 1x1 map): the gradient is exactly 0
            assert g.abs().max().item() < 1e-6, f"grad should vanish, max {g.abs().max().item():.3e}"
        else:
            rel = (g - go).abs() / go.abs().max()
            if not ((rel < 2e-3).double().mean().item() > 0.95 and rel.max().item() < 0.3):
                # per sample: every sample must meet the criterion unless the oracle itself sits on a kink there (margin below
                # 1e-4 of the tensor's scale -- forward differences of 1e-5 are normal behind an instance norm over a 3x3 map),
                # and at most a quarter of the batch may be excused that way
                per = rel.reshape(B, -1)
                ok = ((per < 2e-3).double().mean(dim=1) > 0.95) & (per.max(dim=1).values < 0.3)
                margin = kink_margins(name, P, f0_ref)
                excused = (~ok) & (margin < 1e-4)
                assert bool((ok | excused).all()) and int(excused.sum()) <= max(1, B // 4), \
                    f"grad: max rel {rel.max().item():.3e}; failing samples {(~ok).nonzero().flatten().tolist()}, kink margins {margin.tolist()}"
                keep = ~excused
        want = S.collaborative_refine(f0_ref, gt, dd, Ksteps, 0.1)
        img, dl, ol, os_, of = eng.refine(f0, Ksteps, 0.1)
        close(dl, want[1], 2e-4, "default logit")
        close(img.cpu()[keep], want[0][keep], 5e-2, "images")   # two steps downstream of the kink effect above; a wrong kernel is off by O(1)
    finally:
        N.ARCHS.pop(name, None)
        nets.ARCHS.pop(name, None)
