#!/usr/bin/env python3
"""bn backward of the D tape at [B,8,8,256] vs a float64 formula: stand-alone, in place, with the statistics of each forward path (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgs_amd import kernels as K

d = torch.device("cuda:0")
torch.manual_seed(0)
for shape in [(32, 8, 8, 256), (8, 8, 8, 256), (32, 16, 16, 128), (64, 8, 8, 256), (257, 8, 8, 256)]:
    C = shape[-1]
    x = (torch.randn(shape) * 1.5 + 0.3)
    dy = torch.randn(shape) * 1e-3
    g, b = torch.randn(C) * 0.1 + 1.0, torch.randn(C) * 0.1
    xd = x.double().requires_grad_(True)
    red = (0, 1, 2)
    mu, var = xd.mean(red), xd.var(red, unbiased=False)
    u = (xd - mu) / torch.sqrt(var + 1e-5) * g.double() + b.double()
    y = torch.where(u > 0, u, 0.2 * u)
    (y * dy.double()).sum().backward()
    ref = xd.grad
    X, DY, G_, B_ = x.to(d), dy.to(d), g.to(d), b.to(d)
    _, mean, invstd = K.bn_train_lrelu_fwd(X, G_, B_, 0.2)
    rel = lambda a: float((a.cpu().double() - ref).abs().max() / ref.abs().max())
    out1 = K.bn_train_lrelu_bwd_data(DY, X, G_, B_, mean, invstd, 0.2)
    dy2 = DY.clone()
    out2 = K.bn_train_lrelu_bwd_data(dy2, X, G_, B_, mean, invstd, 0.2, out=dy2)
    print(shape, "stand-alone rel", f"{rel(out1):.2e}", "in place rel", f"{rel(out2):.2e}", "equal", bool(torch.equal(out1, out2)))
