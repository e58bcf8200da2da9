"""the discriminator shaping step (weight gradients + Adam) against torch autograd on the CPU oracle."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nets_ref as N
from oracle import ops_ref as R


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


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s", [(3, 16, 16, 64, 128, 5, 2), (4, 28, 28, 1, 64, 4, 2), (2, 32, 32, 3, 64, 5, 2),
                                               (2, 7, 9, 32, 40, 5, 2), (40, 8, 8, 128, 96, 5, 2), (2, 6, 5, 16, 8, 3, 1)])
def test_conv_weight_and_bias_grads(B, H, W, Cin, Cout, k, s):
    from cgs_amd import kernels as K
    x = rnd((B, H, W, Cin), 1)
    w = rnd((k, k, Cin, Cout), 2, 0.05).requires_grad_(True)
    b = torch.zeros(Cout, requires_grad=True)
    y = R.conv2d(x, w, b, s, s)
    dy = rnd(tuple(y.shape), 3)
    (y * dy).sum().backward()
    d = dev()
    gw = K.conv2d_bwd_weight(x.to(d), dy.to(d), k, k, s, s)
    close(gw, w.grad, 3e-5)
    gw2 = K.conv2d_bwd_weight(x.to(d), dy.to(d), k, k, s, s, out=gw.clone(), accumulate=True)
    close(gw2, 2 * w.grad, 3e-5)
    if Cout % 4 == 0:
        close(K.bias_grad(dy.to(d)), b.grad, 1e-5)


@pytest.mark.parametrize("B,Kin,Nout", [(64, 6272, 1024), (33, 1024, 1), (16, 100, 40)])
def test_linear_weight_grad(B, Kin, Nout):
    from cgs_amd import kernels as K
    x, dy = rnd((B, Kin), 1), rnd((B, Nout), 2)
    close(K.linear_bwd_weight(x.to(dev()), dy.to(dev())), x.t() @ dy, 3e-5)


def _oracle_d_with_lrelu_sides(arch, P, x, sides):
    """The oracle D in float64 with every LeakyReLU taking the side (x > 0 ? 1 : leak) it is TOLD (``sides``: one bool tensor per lrelu, in
    layer order) instead of deciding it from its own pre-activation: the exact gradient of the piecewise-linear branch the GPU's forward
    evaluated.  At batch 64 a handful of the ~10^6 pre-activations of a D pass lie within fp32 rounding of the kink, land on the other side
    than in the oracle's arithmetic and multiply ONE gradient entry by 5 each (measured with tools/diag_shaping2.py: 2 of 524,288 at
    dcgan64 / batch 32, every kernel on the way within 2.4e-6 of float64) -- a measure-zero event of the function, not an error of a kernel."""
    sides = list(sides)
    for L in N.ARCHS[arch]["d"]:
        if L[0] == "lrelu":
            m = sides.pop(0)
            x = torch.where(m, x, R.LRELU_LEAK * x)
        else:
            x = N.run_layers([L], x, P, "discriminator", bn_training=True)
    assert not sides
    return x


@pytest.mark.parametrize("arch,B", [("mnist", 16), ("dcgan32", 8), ("mnist", 64), ("dcgan64", 64)])      # (64: nsgan/main.py:32)
def test_d_shaping_step_matches_autograd(arch, B):
    from cgs_amd import lib as L
    from cgs_amd.engine import _BnTrainLrelu, _Conv, _Linear
    from cgs_amd.nets import to_device
    from cgs_amd.shaping import DShaper
    P = N.init_params(arch, 2019, True)
    real = rnd((B,) + tuple(N.ARCHS[arch]["img"]), 1).clamp(-1, 1)
    fake = torch.tanh(rnd((B,) + tuple(N.ARCHS[arch]["img"]), 2))
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    d = dev()
    Pd = to_device(P, d)
    sh = DShaper(arch, Pd, B, d, learning_rate=1e-3)

    def lrelu_sides(x):        # the side every LeakyReLU took in the GPU's forward of this batch
        sh.tape.forward(x.to(d))
        out = []
        for st in sh.tape.stages:
            if (isinstance(st, (_Conv, _Linear)) and st.epi == L.EPI_LRELU) or (isinstance(st, _BnTrainLrelu) and st.leak != 1.0):
                out.append((st.out > 0).cpu())
        return out
    sides_r, sides_f = lrelu_sides(real), lrelu_sides(fake)
    # (a) float64 autograd of the branch the GPU evaluated: the tight bar, at every size
    Pg = {k: (v.clone().double().requires_grad_(True) if k.startswith("discriminator/") and "moving" not in k else v.double()) for k, v in P.items()}
    lr = _oracle_d_with_lrelu_sides(arch, Pg, real.double(), sides_r)
    lf = _oracle_d_with_lrelu_sides(arch, Pg, fake.double(), sides_f)
    loss = bce(lr, torch.ones_like(lr)) + bce(lf, torch.zeros_like(lf))                 # nsgan/GAN.py:126-131
    loss.backward()
    # (b) plain float32 autograd of the oracle D (its own lrelu decisions)
    Pp = {k: (v.clone().requires_grad_(True) if k.startswith("discriminator/") and "moving" not in k else v) for k, v in P.items()}
    lr32, lf32 = N.discriminator(arch, Pp, real), N.discriminator(arch, Pp, fake)
    loss32 = bce(lr32, torch.ones_like(lr32)) + bce(lf32, torch.zeros_like(lf32))
    loss32.backward()
    got_loss = sh.loss_and_grads(real.to(d), fake.to(d))
    assert abs(got_loss.item() - loss.item()) < 1e-5 * max(1.0, abs(loss.item())) and abs(got_loss.item() - loss32.item()) < 1e-5 * max(1.0, abs(loss32.item()))
    names = [k for k in Pg if Pg[k].requires_grad]
    checked = 0
    for st in sh.tape.stages:
        for attr in ("w", "b", "gamma", "beta"):
            if hasattr(st, "g_" + attr):
                p = getattr(st, attr)
                name = [k for k in names if Pd[k] is p][0]
                g_ref, g32, got = Pg[name].grad, Pp[name].grad.double(), getattr(st, "g_" + attr).cpu().double()
                if float(g_ref.abs().max()) < 1e-6:                    # bias in front of a batch norm: exactly-zero gradient
                    assert float(got.abs().max()) < 1e-4
                else:
                    close(got, g_ref, 2e-5)                               # the evaluated branch, float64: every kernel of the step
                    # against the oracle's OWN branch: identical unless a pre-activation sat on the kink (then a few entries move)
                    l2 = float((got - g32).norm() / g32.norm())
                    assert l2 < (2e-4 if B <= 16 else 2e-2), (name, l2)
                    if B <= 16:
                        close(got, g32, 2e-4)
                checked += 1
    assert checked == len(names)
    Pg = Pp                                                                # (the Adam check below: the plain float32 gradients, as before)
    # one Adam step (tf.train.AdamOptimizer formula) from the step's OWN gradients (their parity is pinned above; Adam's first step is
    # lr_t * g / (|g| sqrt(1 - beta2) + eps): where g is within rounding of zero its sign decides the update, so the oracle's gradient is
    # not the yardstick for the optimizer kernel)
    before = {k: Pd[k].clone() for k in names}
    own = {}
    for st in sh.tape.stages:
        for attr in ("w", "b", "gamma", "beta"):
            if hasattr(st, "g_" + attr):
                own[[k for k in names if Pd[k] is getattr(st, attr)][0]] = getattr(st, "g_" + attr).clone()
    sh.step(real.to(d), fake.to(d))
    lr_t = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.5)
    for k in names:
        g = own[k].cpu()
        want = before[k].cpu() - lr_t * (0.5 * g) / (torch.sqrt(0.001 * g * g) + 1e-8)
        assert (Pd[k].cpu() - want).abs().max().item() <= 1e-6 + 2e-3 * lr_t * 16, k          # (the update itself is <= lr_t * 15.9 per element)
    # the refiner sees the shaped weights
    from cgs_amd.engine import RefineEngine
    eng = RefineEngine(arch, Pd, B, d)
    z = rnd((B, N.ARCHS[arch]["z_dim"]), 5).clamp(-1, 1).to(d)
    f0 = eng.input_to_feature(z).clone()
    l0 = eng.compute_forward_logits_and_grad(f0)[0].clone()
    sh.step(real.to(d), fake.to(d))
    eng.refresh_weights()
    l1 = eng.compute_forward_logits_and_grad(f0)[0].clone()
    assert not torch.equal(l0, l1)
    Pnow = {k: v.cpu() for k, v in Pd.items()}
    with torch.no_grad():
        want = N.discriminator(arch, Pnow, N.feature_to_data(arch, Pnow, f0.cpu())).reshape(B)
    close(l1, want, 2e-4)


def test_shape_step_loop_runs_and_lowers_d_loss_on_refined():
    """A few shaping iterations (refine probabilistically -> D Adam step): D learns to score the refined batch lower."""
    from cgs_amd.engine import RefineEngine
    from cgs_amd.nets import to_device
    from cgs_amd.shaping import DShaper, shape_step
    d = dev()
    B, arch = 16, "mnist"
    Pd = to_device(N.init_params(arch, 2019, True), d)
    eng = RefineEngine(arch, Pd, B, d)
    sh = DShaper(arch, Pd, B, d, learning_rate=2e-3)
    real = rnd((B, 28, 28, 1), 1).clamp(-1, 1).to(d)
    z = rnd((B, 62), 2).clamp(-1, 1).to(d)
    np.random.seed(0)
    losses = [float(shape_step(eng, sh, z, real, 3, 0.1)) for _ in range(6)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_graph_replay_sees_the_shaped_discriminator():
    """hipGraphs are warmed up and captured on one explicit stream, so the capture finds the packed weights
    (no pack kernel is recorded) and ``refresh_weights`` re-packs those very buffers: after a D shaping step a graph REPLAY
    must give what a freshly built eager engine gives on the new weights -- and not what it gave before the step."""
    from cgs_amd.engine import RefineEngine
    from cgs_amd.nets import to_device
    from cgs_amd.shaping import DShaper
    d = dev()
    B, arch, K = 16, "mnist", 3
    Pd = to_device(N.init_params(arch, 2019, True), d)
    geng = RefineEngine(arch, Pd, B, d, use_graph=True)
    sh = DShaper(arch, Pd, B, d, learning_rate=2e-3)
    real = rnd((B, 28, 28, 1), 1).clamp(-1, 1).to(d)
    z = rnd((B, 62), 2).clamp(-1, 1).to(d)
    before = [t.clone() for t in geng.refine_from_z(z, K, 0.1)]           # capture
    again = [t.clone() for t in geng.refine_from_z(z, K, 0.1)]            # replay
    assert all(torch.equal(a, b) for a, b in zip(before, again))
    sh.step(real, before[0])
    geng.refresh_weights()
    after = [t.clone() for t in geng.refine_from_z(z, K, 0.1)]            # replay on the shaped D
    fresh = [t.clone() for t in RefineEngine(arch, Pd, B, d).refine_from_z(z, K, 0.1)]
    assert not torch.equal(after[2], before[2])
    for a, b in zip(after, fresh):
        assert torch.equal(a, b)
