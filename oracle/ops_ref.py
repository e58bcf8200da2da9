"""Operator oracle: nsgan/ops.py semantics on NHWC torch-CPU fp32 tensors.

Every function cites the reference lines it follows (paths under /root/reference).
Weights keep the reference layouts: conv HWIO ``[kh,kw,Cin,Cout]``
(nsgan/ops.py:39), deconv ``[kh,kw,Cout,Cin]`` (nsgan/ops.py:51), linear
``[in,out]`` (nsgan/ops.py:76).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # nsgan/ops.py:23
LRELU_LEAK = 0.2       # nsgan/ops.py:69


def conv_out_size_same(size, stride):
    """nsgan/ops.py:28-29."""
    return int(math.ceil(float(size) / float(stride)))


def same_pads(size, k, s):
    """TF 'SAME' padding: extra pixel goes bottom/right."""
    out = conv_out_size_same(size, s)
    tot = max((out - 1) * s + k - size, 0)
    return tot // 2, tot - tot // 2


def conv2d(x, w, b, d_h=2, d_w=2):
    """tf.nn.conv2d(x, w, [1,d_h,d_w,1], 'SAME') + bias_add  (nsgan/ops.py:41-44)."""
    kh, kw = w.shape[0], w.shape[1]
    pt, pb = same_pads(x.shape[1], kh, d_h)
    pl, pr = same_pads(x.shape[2], kw, d_w)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn, w.permute(3, 2, 0, 1).contiguous(), b, stride=(d_h, d_w))      # (contiguous: torch-CPU's native conv
    # backward refuses a permuted filter gradient for one output channel)
    return y.permute(0, 2, 3, 1).contiguous()


def deconv2d(x, w, b, output_shape, d_h=2, d_w=2):
    """tf.nn.conv2d_transpose(x, w[kh,kw,Cout,Cin], output_shape, strides) + bias
    (nsgan/ops.py:55,61-62): the adjoint of the 'SAME' conv whose *input* has
    ``output_shape``."""
    kh, kw = w.shape[0], w.shape[1]
    Ho, Wo = int(output_shape[1]), int(output_shape[2])
    pt, _ = same_pads(Ho, kh, d_h)
    pl, _ = same_pads(Wo, kw, d_w)
    assert conv_out_size_same(Ho, d_h) == x.shape[1] and conv_out_size_same(Wo, d_w) == x.shape[2]
    full = F.conv_transpose2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), None, stride=(d_h, d_w))
    # full size (H-1)*s+k may be smaller than pt+Ho when k < s; pad with zeros then.
    need_h, need_w = pt + Ho - full.shape[2], pl + Wo - full.shape[3]
    if need_h > 0 or need_w > 0:
        full = F.pad(full, (0, max(need_w, 0), 0, max(need_h, 0)))
    y = full[:, :, pt:pt + Ho, pl:pl + Wo] + b.view(1, -1, 1, 1)
    return y.permute(0, 2, 3, 1).contiguous()


def bn_train(x, gamma, beta):
    """contrib batch_norm(is_training=True): batch statistics, biased variance
    (nsgan/ops.py:19-26).  Works for [B,H,W,C] and [B,C]."""
    red = tuple(range(x.dim() - 1))
    mean = x.mean(dim=red, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=red, keepdim=True)
    return gamma * (x - mean) / torch.sqrt(var + BN_EPS) + beta


def bn_infer(x, gamma, beta, moving_mean, moving_var):
    """contrib batch_norm(is_training=False): moving averages (nsgan/GAN.py:87,94)."""
    return gamma * (x - moving_mean) / torch.sqrt(moving_var + BN_EPS) + beta


def instance_norm(x, scale, offset):
    """Instance norm: statistics over the pixels of every (sample, channel); biased variance, eps 1e-5.
    (No reference call site: used only by the build-defined CycleGAN / PatchGAN nets of BASELINE config 5.)"""
    mean = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    return scale * (x - mean) / torch.sqrt(var + BN_EPS) + offset


def lrelu(x, leak=LRELU_LEAK):
    """tf.maximum(x, leak*x)  (nsgan/ops.py:69-70)."""
    return torch.maximum(x, leak * x)


def linear(x, matrix, bias):
    """tf.matmul(x, Matrix) + bias  (nsgan/ops.py:81-83)."""
    return x @ matrix + bias


def sigmoid_xent_ones(logits):
    """tf.nn.sigmoid_cross_entropy_with_logits(labels=1) = softplus(-logit),
    not reduced (nsgan/GAN.py:176-177)."""
    return F.softplus(-logits)


# ---------------------------------------------------------------------------
# Independent direct-loop evaluation (numpy f64) used to cross-check the two
# torch formulations above on tiny shapes (tests/test_oracle_ops.py).
# ---------------------------------------------------------------------------
def conv2d_loops(x, w, b, s):
    B, H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    Ho, Wo = conv_out_size_same(H, s), conv_out_size_same(W, s)
    pt, _ = same_pads(H, kh, s)
    pl, _ = same_pads(W, kw, s)
    y = np.zeros((B, Ho, Wo, Co))
    for oy in range(Ho):
        for ox in range(Wo):
            for ky in range(kh):
                for kx in range(kw):
                    iy, ix = oy * s + ky - pt, ox * s + kx - pl
                    if 0 <= iy < H and 0 <= ix < W:
                        y[:, oy, ox, :] += x[:, iy, ix, :] @ w[ky, kx]
    return y + b


def deconv2d_loops(x, w, b, Ho, Wo, s):
    """Scatter form of the transposed conv: y[.., oy*s+ky-pt, ..] += x[oy] * w[ky,kx,co,ci]."""
    B, H, W, Ci = x.shape
    kh, kw, Co, _ = w.shape
    pt, _ = same_pads(Ho, kh, s)
    pl, _ = same_pads(Wo, kw, s)
    y = np.zeros((B, Ho, Wo, Co))
    for iy in range(H):
        for ix in range(W):
            for ky in range(kh):
                for kx in range(kw):
                    oy, ox = iy * s + ky - pt, ix * s + kx - pl
                    if 0 <= oy < Ho and 0 <= ox < Wo:
                        y[:, oy, ox, :] += x[:, iy, ix, :] @ w[ky, kx].T
    return y + b
