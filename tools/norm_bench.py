#!/usr/bin/env python3
"""Norm passes alone at the tensor sizes the benched configurations run them at (development aid, GPU box):
    CGS_LIB=.../libcgs_exp.so LB_AB="CGS_APPLY_GROUPED=0;CGS_APPLY_GROUPED=1" python tools/norm_bench.py
Forward (statistics + apply: instnorm_lrelu_fwd; the engine's fused form skips the statistics pass) and backward-data (sums + finalize +
apply, in place as the engine runs it), HIP events over LB_ITERS calls, algorithmic GB/s (fwd 3 tensor passes, bwd 5) against 8 TB/s;
modes interleaved in one process, best of LB_REPS rounds.  Results of the modes are compared bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgs_amd import kernels as K

iters, reps = int(os.environ.get("LB_ITERS", "20")), int(os.environ.get("LB_REPS", "3"))
modes = [m for m in os.environ.get("LB_AB", "").split(";") if m] or [""]
d = torch.device("cuda:0")
# (groups, rows per group as (n, h, w), C, what)
CASES = [
    (8, (1, 64, 64), 256, "cyclegan256 res blocks"), (8, (1, 128, 128), 128, "cyclegan256 d128 / u128"), (8, (1, 256, 256), 64, "cyclegan256 c7s1-64 / u64"),
    (8, (1, 64, 64), 128, "patchgan 128"), (8, (1, 32, 32), 256, "patchgan 256"), (8, (1, 31, 31), 512, "patchgan 512"),
    (1, (1024, 16, 16), 128, "dcgan64 d_bn1"), (1, (1024, 8, 8), 256, "dcgan64 d_bn2"), (1, (1024, 4, 4), 512, "dcgan64 d_bn3"),
    (8, (256, 8, 8), 128, "dcgan32 8x256 d_bn1"), (8, (256, 4, 4), 256, "dcgan32 d_bn2"), (8, (256, 2, 2), 512, "dcgan32 d_bn3"),
    (32, (64, 7, 7), 128, "mnist 32x64 d_bn2"), (1, (64, 8, 8), 256, "dcgan64 batch 64 d_bn2"),
]


def set_mode(m):
    for kv in m.split(","):
        if kv:
            k, v = kv.split("=")
            os.environ.pop(k, None) if v == "-" else os.environ.__setitem__(k, v)


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print("modes:", modes)
for G, (n, h, w), C, what in CASES:
    torch.manual_seed(0)
    x = torch.randn(G, n * h * w, 1, C, device=d)
    dy0 = torch.randn_like(x)
    gamma, beta = torch.rand(C, device=d) + 0.5, torch.randn(C, device=d) * 0.1
    mb = x.numel() * 4 / 1e6
    bf, bb, outs = [1e30] * len(modes), [1e30] * len(modes), []
    for r in range(reps):
        for i, m in enumerate(modes):
            set_mode(m)
            y, mean, invstd = K.instnorm_lrelu_fwd(x, gamma, beta, 0.2)
            dy = dy0.clone()
            dx = K.instnorm_lrelu_bwd_data(dy, x, gamma, beta, mean, invstd, 0.2, out=dy).clone()
            if r == 0:
                outs.append((y.clone(), dx))
            bf[i] = min(bf[i], timeit(lambda: K.instnorm_lrelu_fwd(x, gamma, beta, 0.2, out=y, stats=(mean, invstd))))
            bb[i] = min(bb[i], timeit(lambda: K.instnorm_lrelu_bwd_data(dy, x, gamma, beta, mean, invstd, 0.2, out=dy)))
    same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    print(f"{what:28s} {G:3d} x {n * h * w:7d} x {C:4d} ({mb:6.1f} MB)  fwd " + " | ".join(f"{t:7.1f} us {3 * mb / t / 8:.3f}" for t in bf)
          + "   bwd " + " | ".join(f"{t:7.1f} us {5 * mb / t / 8:.3f}" for t in bb) + ("" if len(modes) == 1 else f"   bit-equal: {same}"))
