#!/usr/bin/env python3
"""Per-layer timing of the conv-family kernels at a BASELINE config (development aid, GPU box).
    python tools/layer_bench.py [arch] [B]
Prints ms and TFLOP/s per layer and direction, hipEvent-timed on the current stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgs_amd import kernels as K, lib

arch = sys.argv[1] if len(sys.argv) > 1 else "dcgan64"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
d = torch.device("cuda:0")
s = {"dcgan64": 64, "dcgan32": 32}[arch]
convs = [(s, 3, 64), (s // 2, 64, 128), (s // 4, 128, 256), (s // 8, 256, 512)]          # H, Cin, Cout
deconvs = [(s // 8, 256, 128), (s // 4, 128, 64), (s // 2, 64, 3)]                       # H, Cin, Cout


def timeit(fn, n=int(os.environ.get("LB_ITERS", "5"))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot = 0.0
for H, Ci, Co in convs:
    x = torch.randn(B, H, H, Ci, device=d); w = torch.randn(5, 5, Ci, Co, device=d) * 0.02; b = torch.zeros(Co, device=d)
    y = K.conv2d_fwd(x, w, b); dy = torch.randn_like(y)
    fl = 2.0 * B * (H // 2) ** 2 * Co * 25 * Ci
    t1 = timeit(lambda: K.conv2d_fwd(x, w, b, out=y)); t2 = timeit(lambda: K.conv2d_bwd_data(dy, w, (H, H), out=x))
    print(f"conv   {H:3d}x{H:<3d} {Ci:4d}->{Co:<4d} fwd {t1:8.3f} ms {fl/t1/1e9:7.1f} TF | bwd {t2:8.3f} ms {fl/t2/1e9:7.1f} TF")
    tot += t1 + t2
for H, Ci, Co in deconvs:
    x = torch.randn(B, H, H, Ci, device=d); w = torch.randn(5, 5, Co, Ci, device=d) * 0.02; b = torch.zeros(Co, device=d)
    y = K.deconv2d_fwd(x, w, b, (2 * H, 2 * H)); dy = torch.randn_like(y)
    fl = 2.0 * B * H * H * Ci * 25 * Co
    t1 = timeit(lambda: K.deconv2d_fwd(x, w, b, (2 * H, 2 * H), out=y)); t2 = timeit(lambda: K.deconv2d_bwd_data(dy, w, (H, H), out=x))
    print(f"deconv {H:3d}x{H:<3d} {Ci:4d}->{Co:<4d} fwd {t1:8.3f} ms {fl/t1/1e9:7.1f} TF | bwd {t2:8.3f} ms {fl/t2/1e9:7.1f} TF")
    tot += t1 + t2
print(f"sum fwd+bwd of all conv-family layers: {tot:.2f} ms for B={B}")
