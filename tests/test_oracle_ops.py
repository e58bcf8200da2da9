"""Cross-check the operator oracle (TF-'SAME' conv / conv_transpose, batch norm)
three independent ways: direct numpy loops, the adjoint identity, closed forms."""
import numpy as np
import pytest
import torch

from oracle import ops_ref as R


def test_same_pads_table():
    # SURVEY Appendix B
    assert R.same_pads(28, 4, 2) == (1, 1) and R.same_pads(14, 4, 2) == (1, 1)
    for H in (64, 32, 16, 8, 4):
        assert R.same_pads(H, 5, 2) == (1, 2)
    assert R.same_pads(7, 5, 2) == (2, 2) and R.same_pads(5, 3, 1) == (1, 1)


@pytest.mark.parametrize("H,W,k,s", [(8, 8, 5, 2), (7, 9, 5, 2), (14, 14, 4, 2), (6, 5, 3, 1), (9, 4, 4, 2)])
def test_conv2d_vs_loops(H, W, k, s):
    rs = np.random.RandomState(0)
    x, w, b = rs.randn(2, H, W, 3), rs.randn(k, k, 3, 4), rs.randn(4)
    y = R.conv2d(torch.tensor(x), torch.tensor(w), torch.tensor(b), s, s).numpy()
    np.testing.assert_allclose(y, R.conv2d_loops(x, w, b, s), rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("Ho,Wo,k,s", [(8, 8, 5, 2), (7, 9, 5, 2), (14, 14, 4, 2), (6, 5, 3, 1), (16, 16, 5, 2)])
def test_deconv2d_vs_loops_and_adjoint(Ho, Wo, k, s):
    rs = np.random.RandomState(1)
    H, W = R.conv_out_size_same(Ho, s), R.conv_out_size_same(Wo, s)
    x, w, b = rs.randn(2, H, W, 4), rs.randn(k, k, 3, 4), rs.randn(3)
    xt, wt, bt = torch.tensor(x), torch.tensor(w), torch.tensor(b)
    y = R.deconv2d(xt, wt, bt, (2, Ho, Wo, 3), s, s)
    np.testing.assert_allclose(y.numpy(), R.deconv2d_loops(x, w, b, Ho, Wo, s), rtol=1e-10, atol=1e-10)
    # adjoint: <conv(u; w), x> == <u, deconv(x; w) - b>   (deconv weights [kh,kw,Cout,Cin] ARE the HWIO
    # weights of the conv it transposes, nsgan/ops.py:51)
    u = torch.tensor(rs.randn(2, Ho, Wo, 3))
    lhs = (R.conv2d(u, wt, torch.zeros(4, dtype=torch.float64), s, s) * xt).sum()
    rhs = (u * (y - bt)).sum()
    assert abs(lhs - rhs) <= 1e-9 * max(1.0, abs(lhs))


def test_bn_closed_forms():
    rs = np.random.RandomState(2)
    x = torch.tensor(rs.randn(6, 3, 3, 5))
    g, b = torch.tensor(rs.rand(5) + 0.5), torch.tensor(rs.randn(5))
    y = R.bn_train(x, g, b)
    flat = ((y - b) / g).reshape(-1, 5)
    np.testing.assert_allclose(flat.mean(0).numpy(), 0, atol=1e-12)
    np.testing.assert_allclose((flat ** 2).mean(0).numpy(), (x.reshape(-1, 5).var(0, unbiased=False) /
                               (x.reshape(-1, 5).var(0, unbiased=False) + R.BN_EPS)).numpy(), rtol=1e-10)
    # backward-data closed form
    xr = x.clone().requires_grad_(True)
    dy = torch.tensor(rs.randn(6, 3, 3, 5))
    (R.bn_train(xr, g, b) * dy).sum().backward()
    mu = x.mean((0, 1, 2)); var = ((x - mu) ** 2).mean((0, 1, 2)); r = 1 / torch.sqrt(var + R.BN_EPS)
    xh = (x - mu) * r
    dx = g * r * (dy - dy.mean((0, 1, 2)) - xh * (dy * xh).mean((0, 1, 2)))
    np.testing.assert_allclose(xr.grad.numpy(), dx.numpy(), rtol=1e-9, atol=1e-11)
    mm, mv = torch.tensor(rs.randn(5)), torch.tensor(rs.rand(5) + 0.1)
    yi = R.bn_infer(x, g, b, mm, mv)
    a = g / torch.sqrt(mv + R.BN_EPS)
    np.testing.assert_allclose(yi.numpy(), (a * x + (b - a * mm)).numpy(), rtol=1e-12, atol=1e-12)


def test_lrelu_and_loss_seed():
    x = torch.linspace(-3, 3, 13)
    np.testing.assert_allclose(R.lrelu(x).numpy(), np.where(x.numpy() > 0, x.numpy(), 0.2 * x.numpy()), rtol=1e-6)
    l = x.clone().double().requires_grad_(True)
    R.sigmoid_xent_ones(l).sum().backward()
    np.testing.assert_allclose(l.grad.numpy(), (torch.sigmoid(x.double()) - 1).numpy(), rtol=1e-10)
