#!/usr/bin/env python3
"""Accuracy of the two contraction arithmetics against fp64 (GPU box): the exact-fp32 MFMA kernel (an fp32 fma chain over K) and
the split-bf16 kernel (csrc/igemm_bx6.hip), same layers, same inputs; plus torch-CPU fp32 (the oracle's arithmetic) for scale.
    python tools/bx6_accuracy.py
Prints, per layer and direction, error / sum|a||b| (max and mean over the outputs) for each arithmetic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from cgs_amd import kernels as K
from oracle import ops_ref as R

d = torch.device("cuda:0")
torch.manual_seed(0)


def stats(got, ref64, scale64):
    e = (got.double().cpu() - ref64).abs() / scale64
    return f"max {e.max().item():.2e} mean {e.mean().item():.2e}"


for name, B, H, Ci, Co, k, s in (("d_h2 16x16 128->256", 8, 16, 128, 256, 5, 2), ("d_h3 8x8 256->512", 16, 8, 256, 512, 5, 2),
                                 ("res 3x3 256->256 s1", 2, 32, 256, 256, 3, 1)):
    x = torch.randn(B, H, H, Ci) * (1 + 3 * torch.rand(B, H, H, Ci))
    w = torch.randn(k, k, Ci, Co) * 0.05
    b = torch.zeros(Co)
    ref = R.conv2d(x.double(), w.double(), b.double(), s, s)
    scale = R.conv2d(x.double().abs(), w.double().abs(), b.double(), s, s)
    cpu32 = R.conv2d(x, w, b, s, s)
    out = {}
    for mode in ("f32", "bx6_all"):
        K.set_contraction(mode)
        out[mode] = K.conv2d_fwd(x.to(d), w.to(d), b.to(d), s, s)
    print(f"{name} fwd (K = {k * k * Ci}):  f32-MFMA {stats(out['f32'], ref, scale)} | bx6 {stats(out['bx6_all'], ref, scale)} | torch-CPU fp32 {stats(cpu32, ref, scale)}")
    print(f"      signed mean error / sum|a||b|:  f32-MFMA {((out['f32'].double().cpu() - ref) / scale).mean().item():+.2e} | bx6 {((out['bx6_all'].double().cpu() - ref) / scale).mean().item():+.2e}")
    # backward-data (transposed direction)
    Ho = ref.shape[1]
    dy = torch.randn(B, Ho, Ho, Co)
    x64 = x.double().requires_grad_(True)
    (R.conv2d(x64, w.double(), b.double(), s, s) * dy.double()).sum().backward()
    gref = x64.grad
    xa = x.double().abs().requires_grad_(True)
    (R.conv2d(xa, w.double().abs(), b.double(), s, s) * dy.double().abs()).sum().backward()
    gscale = xa.grad
    for mode in ("f32", "bx6_all"):
        K.set_contraction(mode)
        out[mode] = K.conv2d_bwd_data(dy.to(d), w.to(d), (H, H), s, s)
    print(f"{name} bwd-data:  f32-MFMA {stats(out['f32'], gref, gscale)} | bx6 {stats(out['bx6_all'], gref, gscale)}")
K.set_contraction("f32")
