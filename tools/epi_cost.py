#!/usr/bin/env python3
"""What the fused epilogues cost on the short-K transposed layers of the MNIST net at 2048 images (development aid, GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgs_amd import kernels as K, lib as L
d = torch.device("cuda:0")
B = 2048


def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


x = torch.randn(B, 7, 7, 128, device=d); w = torch.randn(4, 4, 64, 128, device=d) * 0.05; b = torch.zeros(64, device=d)
a, c = torch.rand(64, device=d) + 0.5, torch.zeros(64, device=d)
y = torch.empty(B, 14, 14, 64, device=d)
signs = torch.empty(y.numel() // 32, dtype=torch.int32, device=d)
print("deconv_fwd 7x7x128->14x14x64: plain %.1f us | affine+relu %.1f us | affine+relu+signs %.1f us" % (
    timeit(lambda: K.deconv2d_fwd(x, w, b, (14, 14), 2, 2, out=y)),
    timeit(lambda: K.deconv2d_fwd(x, w, b, (14, 14), 2, 2, L.EPI_AFFINE_RELU, a, c, out=y)),
    timeit(lambda: K.deconv2d_fwd(x, w, b, (14, 14), 2, 2, L.EPI_AFFINE_RELU, a, c, out=y, signs=signs)) if K.conv_signs_ok(L.DECONV_FWD, B, 7, 7, 128, 14, 14, 64, 4, 4, 2, 2, L.EPI_AFFINE_RELU) else -1))
dy = torch.randn(B, 7, 7, 128, device=d); wc = torch.randn(4, 4, 64, 128, device=d) * 0.05; aux = torch.randn(B, 14, 14, 64, device=d); dx = torch.empty(B, 14, 14, 64, device=d)
print("conv_bwd 14x14x64<-7x7x128: plain %.1f us | + lrelu' (aux) %.1f us" % (
    timeit(lambda: K.conv2d_bwd_data(dy, wc, (14, 14), 2, 2, out=dx)),
    timeit(lambda: K.conv2d_bwd_data(dy, wc, (14, 14), 2, 2, out=dx, epilogue=L.EPI_LRELU_BWD, ep_aux=aux))))
# the same contraction with a longer K: k = 6 (3x3 taps per class, K = 1152)
for k in (4, 6, 8):
    wk = torch.randn(k, k, 64, 128, device=d) * 0.05
    t = timeit(lambda: K.deconv2d_fwd(x, wk, b, (14, 14), 2, 2, out=y))
    fl = float(L.load().cgs_last_executed_flops())
    print(f"deconv_fwd k={k}: {t:.1f} us  {fl / t / 1e6:.1f} TF issued ({fl / t / 1e6 / 157.3:.3f})  {L.last_kernel()}")
# the same layer to 128 channels (128 x 128 tiles): is the N = 64 tile the limit?
y2 = torch.empty(B, 14, 14, 128, device=d); b2 = torch.zeros(128, device=d)
for k in (4, 6):
    wk = torch.randn(k, k, 128, 128, device=d) * 0.05
    t = timeit(lambda: K.deconv2d_fwd(x, wk, b2, (14, 14), 2, 2, out=y2))
    fl = float(L.load().cgs_last_executed_flops())
    print(f"deconv_fwd 128->128 k={k}: {t:.1f} us  {fl / t / 1e6:.1f} TF issued ({fl / t / 1e6 / 157.3:.3f})  {L.last_kernel()}")
# dcgan64's N = 64 transposed layer against a 128-channel twin
B2 = 1024
x3 = torch.randn(B2, 16, 16, 128, device=d)
for Co in (64, 128):
    w3 = torch.randn(5, 5, Co, 128, device=d) * 0.02; b3 = torch.zeros(Co, device=d); y3 = torch.empty(B2, 32, 32, Co, device=d)
    t = timeit(lambda: K.deconv2d_fwd(x3, w3, b3, (32, 32), 2, 2, out=y3), n=10)
    fl = float(L.load().cgs_last_executed_flops())
    print(f"deconv_fwd 16x16x128->32x32x{Co} (B=1024): {t:.1f} us  {fl / t / 1e6:.1f} TF issued ({fl / t / 1e6 / 157.3:.3f})  {L.last_kernel()}")
