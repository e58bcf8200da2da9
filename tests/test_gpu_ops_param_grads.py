"""The operator API (cgs_amd.ops = nsgan/ops.py's surface) is differentiable w.r.t. its VARIABLES too, for the training-side
callers (nsgan/GAN.py:141-146): parameter gradients of a small generator-tail + discriminator built from the operators,
against torch autograd on the CPU oracle's ops with the same variables."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ops_ref as R


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def net_ops(ops, x, B, reuse=False):
    """Built twice like the reference builds D (nsgan/GAN.py:62,75): first call creates the variables, later calls reuse."""
    with ops.variable_scope("generator", reuse=reuse):
        h = ops.relu(ops.bn(ops.deconv2d(x, [B, 8, 8, 16], name="g_dc1"), is_training=True, scope="g_bn1"))
        img = ops.tanh(ops.deconv2d(h, [B, 16, 16, 4], name="g_dc2"))
    with ops.variable_scope("discriminator", reuse=reuse):
        d = ops.lrelu(ops.conv2d(img, 16, name="d_c1"))
        d = ops.lrelu(ops.bn(ops.conv2d(d, 32, name="d_c2"), is_training=True, scope="d_bn2"))
        d = d.reshape(B, -1)
        d = ops.lrelu(ops.bn(ops.linear(d, 64, scope="d_fc3"), is_training=True, scope="d_bn3"))
        return ops.linear(d, 1, scope="d_fc4")


def net_ref(P, x, B):
    g = lambda n: P["generator/" + n]
    d_ = lambda n: P["discriminator/" + n]
    h = torch.relu(R.bn_train(R.deconv2d(x, g("g_dc1/w"), g("g_dc1/biases"), (B, 8, 8, 16), 2, 2), g("g_bn1/gamma"), g("g_bn1/beta")))
    img = torch.tanh(R.deconv2d(h, g("g_dc2/w"), g("g_dc2/biases"), (B, 16, 16, 4), 2, 2))
    d = R.lrelu(R.conv2d(img, d_("d_c1/w"), d_("d_c1/biases"), 2, 2))
    d = R.lrelu(R.bn_train(R.conv2d(d, d_("d_c2/w"), d_("d_c2/biases"), 2, 2), d_("d_bn2/gamma"), d_("d_bn2/beta")))
    d = d.reshape(B, -1)
    d = R.lrelu(R.bn_train(R.linear(d, d_("d_fc3/Matrix"), d_("d_fc3/bias")), d_("d_bn3/gamma"), d_("d_bn3/beta")))
    return R.linear(d, d_("d_fc4/Matrix"), d_("d_fc4/bias"))


def test_parameter_gradients_match_autograd_on_the_oracle():
    from cgs_amd import ops
    B = 12
    ops.reset_variables()
    ops.set_device(dev())
    torch.manual_seed(3)
    x = torch.randn(B, 4, 4, 32)
    xd = x.to(dev())
    with torch.no_grad():
        net_ops(ops, xd, B)                                      # creates the variables with the reference's initialisers
    V = ops.variables()
    rs = np.random.RandomState(5)
    for k, v in V.items():                                       # move gamma / beta / biases off their trivial init values
        leaf = k.rsplit("/", 1)[1]
        if leaf in ("gamma", "beta", "biases", "bias"):
            v.add_(torch.from_numpy(rs.normal(0, 0.2, tuple(v.shape)).astype(np.float32)).to(v.device))
    train = [k for k in V if not k.endswith(("moving_mean", "moving_variance"))]
    for k in train:
        V[k].requires_grad_(True)
    logits = net_ops(ops, xd, B, reuse=True)
    loss = torch.nn.functional.softplus(-logits).mean() + 0.1 * (logits ** 2).mean()
    loss.backward()
    P = {k: v.detach().cpu().clone().requires_grad_(k in train) for k, v in V.items()}
    lr = net_ref(P, x, B)
    loss_ref = torch.nn.functional.softplus(-lr).mean() + 0.1 * (lr ** 2).mean()
    loss_ref.backward()
    assert abs(loss.item() - loss_ref.item()) <= 1e-5 * max(1.0, abs(loss_ref.item()))
    for k in train:
        g, gr = V[k].grad.cpu().double(), P[k].grad.double()
        scale = gr.abs().max().item()
        if scale < 1e-6:                                         # a bias in front of a batch norm: no gradient but rounding noise
            assert g.abs().max().item() < 1e-5, k
            continue
        assert (g - gr).abs().max().item() <= 3e-4 * scale, (k, (g - gr).abs().max().item(), scale)
    for k in train:
        V[k].requires_grad_(False)
    ops.reset_variables()



def test_training_mode_bn_updates_the_moving_averages_with_decay_09():
    """nsgan/ops.py:19-26: contrib batch_norm(decay=0.9, updates_collections=None) moves moving_mean / moving_variance towards the BATCH
    statistics (biased variance) in place at every training-mode call -- also during refinement, where D runs on batch statistics."""
    from cgs_amd import ops
    d = dev()
    ops.reset_variables()
    g = torch.Generator().manual_seed(5)
    x = (torch.randn((6, 5, 5, 8), generator=g) * 2.0 + 0.7).float()
    with ops.variable_scope("discriminator"):
        ops.bn(x.to(d), is_training=True, scope="d_bn")
    with ops.variable_scope("discriminator", reuse=True):
        ops.bn(x.to(d), is_training=True, scope="d_bn", leak=0.2)
    V = ops.variables()
    mean, var = x.double().reshape(-1, 8).mean(0), x.double().reshape(-1, 8).var(0, unbiased=False)
    mm, mv = torch.zeros(8, dtype=torch.float64), torch.ones(8, dtype=torch.float64)
    for _ in range(2):
        mm, mv = 0.9 * mm + 0.1 * mean, 0.9 * mv + 0.1 * var
    assert torch.allclose(V["discriminator/d_bn/moving_mean"].cpu().double(), mm, rtol=1e-5, atol=1e-6)
    assert torch.allclose(V["discriminator/d_bn/moving_variance"].cpu().double(), mv, rtol=1e-5, atol=1e-6)
    ops.reset_variables()
