#!/usr/bin/env python3
"""Timing of the fully connected layers at the MNIST discriminator's shapes (development aid, GPU box).
    python tools/linear_bench.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgs_amd import kernels as K, lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = torch.device("cuda:0")


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[2:]] or [(6272, 1024), (1024, 6272), (1024, 1), (8192, 1)]
for kin, kout in shapes:
    x = torch.randn(B, kin, device=d); w = torch.randn(kin, kout, device=d) * 0.02; b = torch.zeros(kout, device=d)
    y = K.linear_fwd(x, w, b); name_f = lib.last_kernel()
    dy = torch.randn_like(y)
    dx = K.linear_bwd_data(dy, w); name_b = lib.last_kernel()
    tf = timeit(lambda: K.linear_fwd(x, w, b, out=y)); tb = timeit(lambda: K.linear_bwd_data(dy, w, out=dx))
    fl = 2.0 * B * kin * kout
    print(f"linear B={B} {kin:5d}->{kout:<5d} fwd {tf:7.1f} us {fl/tf/1e6:6.2f} TF [{name_f}] | bwd {tb:7.1f} us {fl/tb/1e6:6.2f} TF [{name_b}]"
          f" | weights {kin*kout*4/1e6:.1f} MB = {kin*kout*4/tf/1e6:.2f} / {kin*kout*4/tb/1e6:.2f} TB/s")
