#!/usr/bin/env python3
"""Diagnostic: what does each NON-contraction part of a step cost the two-batches-in-flight headline?  Runs bench.py's dcgan64
loop (two engines, two streams, hipGraph replay) with groups of kernels replaced by no-ops (results are then meaningless --
timing only):   python tools/skip_probe.py          (GPU box)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cgs_amd import kernels as K, nets                  # noqa: E402
from cgs_amd.engine import RefineEngine                 # noqa: E402

dev = torch.device("cuda:0")
B, Ks, steps = 1024, 20, 12
P = nets.init_params("dcgan64", dev, seed=2019)
z = torch.from_numpy(np.random.RandomState(2019).uniform(-1, 1, (steps + 2, B, 100)).astype(np.float32)).to(dev)
real = {n: getattr(K, n) for n in dir(K) if callable(getattr(K, n)) and not n.startswith("_")}


def run(skip, n_streams=2):
    for n, f in real.items():
        setattr(K, n, f)
    for n in skip:
        if n in ("bn_train_lrelu_fwd_from_partials", "bn_train_lrelu_fwd"):
            setattr(K, n, lambda x, *a, out=None, stats=None, **k: (out if out is not None else x, None, None))
        elif n == "bn_train_lrelu_bwd_data":
            setattr(K, n, lambda dy, x, *a, out=None, **k: out if out is not None else dy)
        elif n in ("refine_update", "refine_select", "refine_select_rows"):
            setattr(K, n, lambda *a, **k: None)
        elif n == "conv2d_fwd_3":          # the 3-channel layers: skip by shape
            f = real["conv2d_fwd"]; setattr(K, "conv2d_fwd", lambda x, w, *a, out=None, f=f, **k: out if x.shape[-1] == 3 and out is not None else f(x, w, *a, out=out, **k))
        elif n == "conv2d_bwd_3":
            f = real["conv2d_bwd_data"]; setattr(K, "conv2d_bwd_data", lambda dy, w, hw, *a, out=None, f=f, **k: out if w.shape[2] == 3 and out is not None else f(dy, w, hw, *a, out=out, **k))
        elif n == "deconv2d_fwd_3":
            f = real["deconv2d_fwd"]; setattr(K, "deconv2d_fwd", lambda x, w, b, hw, *a, out=None, f=f, **k: out if w.shape[2] == 3 and out is not None else f(x, w, b, hw, *a, out=out, **k))
        elif n == "deconv2d_bwd_3":
            f = real["deconv2d_bwd_data"]; setattr(K, "deconv2d_bwd_data", lambda dy, w, hw, *a, out=None, f=f, **k: out if w.shape[2] == 3 and out is not None else f(dy, w, hw, *a, out=out, **k))
    engines = [RefineEngine("dcgan64", P, B, dev, use_graph=True) for _ in range(n_streams)]
    streams = [torch.cuda.Stream(dev) for _ in engines]

    def step(i):
        with torch.cuda.stream(streams[i % n_streams]):
            engines[i % n_streams].refine_from_z(z[i], Ks, 0.1)
    for i in range(n_streams):
        step(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(2, steps + 2):
        step(i)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    del engines
    torch.cuda.empty_cache()
    return dt * 1e3


THREE = ["conv2d_fwd_3", "conv2d_bwd_3", "deconv2d_fwd_3", "deconv2d_bwd_3"]
BN = ["bn_train_lrelu_fwd_from_partials", "bn_train_lrelu_fwd", "bn_train_lrelu_bwd_data"]
UPD = ["refine_update", "refine_select", "refine_select_rows"]
for ns in (2, 1):
    base = run([], ns)
    print(f"{ns} in flight: full step {base:.2f} ms")
    for name, sk in (("without batch norm passes", BN), ("without the 3-channel layers", THREE), ("without update / select", UPD),
                     ("without all three", BN + THREE + UPD)):
        t = run(sk, ns)
        print(f"   {name:32s} {t:8.2f} ms  ({base - t:+.2f} ms, {100 * (base - t) / base:.1f} %)")
