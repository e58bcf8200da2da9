#!/usr/bin/env python3
"""Per-layer timing of the split-bf16 implicit GEMM on the headline's layers (development aid, GPU box).
    [CGS_LIB=<variant .so>] python tools/bx6_bench.py [iters]
The ten 107-GFLOP layers of dcgan64 at batch 1024 (forward and backward-data of every mid layer), in `bx6_all` mode so that every
eligible layer takes the kernel; prints us per layer, the kernel that ran, and the sum over the bx6 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgs_amd import kernels as K, lib

K.set_contraction(os.environ.get("CGS_CONTRACTION", "bx6"))
d = torch.device("cuda:0")
B = int(os.environ.get("BX6_B", "1024"))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ZERO = bool(os.environ.get("BX6_ZERO"))      # all-zero activations and weights: same instruction stream, no switching activity (is the kernel clock / power bound?)


def timeit(fn):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


tot, rows = 0.0, []
for kind, H, Ci, Co in (("conv", 32, 64, 128), ("conv", 16, 128, 256), ("conv", 8, 256, 512), ("deconv", 8, 256, 128), ("deconv", 16, 128, 64)):
    torch.manual_seed(0)
    if kind == "conv":
        x = torch.randn(B, H, H, Ci, device=d); w = torch.randn(5, 5, Ci, Co, device=d) * 0.02; b = torch.zeros(Co, device=d)
        if ZERO: x.zero_(); w.zero_()
        y = K.conv2d_fwd(x, w, b); dy = torch.randn_like(y)
        if ZERO: dy.zero_()
        f = lambda: K.conv2d_fwd(x, w, b, out=y)
        g = lambda: K.conv2d_bwd_data(dy, w, (H, H), out=x)
    else:
        x = torch.randn(B, H, H, Ci, device=d); w = torch.randn(5, 5, Co, Ci, device=d) * 0.02; b = torch.zeros(Co, device=d)
        if ZERO: x.zero_(); w.zero_()
        y = K.deconv2d_fwd(x, w, b, (2 * H, 2 * H)); dy = torch.randn_like(y)
        if ZERO: dy.zero_()
        f = lambda: K.deconv2d_fwd(x, w, b, (2 * H, 2 * H), out=y)
        g = lambda: K.deconv2d_bwd_data(dy, w, (H, H), out=x)
    for tag, fn in (("fwd", f), ("bwd", g)):
        t = timeit(fn); fn(); kn = lib.last_kernel()
        rows.append(f"{kind:6s} {H:2d}x{H:<2d} {Ci:3d}->{Co:<3d} {tag} {t:7.1f} us  {kn}")
        if kn.startswith("igemm_bx6"):
            tot += t
print("\n".join(rows))
print(f"SUM of igemm_bx6 launches: {tot:.1f} us   (lib: {os.environ.get('CGS_LIB', 'default')})")
