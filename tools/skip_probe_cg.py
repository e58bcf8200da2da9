#!/usr/bin/env python3
"""tools/skip_probe.py for BASELINE config 5's per-GPU share (cyclegan256, batch 8, K = 20): what do the instance-norm passes and
the RGB head / stem layers cost with two batches in flight?  (kernels replaced by no-ops: timing only)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cgs_amd import kernels as K, nets                  # noqa: E402
from cgs_amd.engine import RefineEngine                 # noqa: E402

dev = torch.device("cuda:0")
arch, B, Ks, steps = "cyclegan256", 8, 20, 12
A = nets.ARCHS[arch]
P = nets.init_params(arch, dev, seed=2019)
z = torch.from_numpy(np.random.RandomState(2019).uniform(-1, 1, (steps + 2, B) + nets.g_input_shape(A)).astype(np.float32)).to(dev)
real = {n: getattr(K, n) for n in dir(K) if callable(getattr(K, n)) and not n.startswith("_")}


def run(skip, n_streams=2):
    for n, f in real.items():
        setattr(K, n, f)
    if "norm" in skip:
        K.instnorm_lrelu_fwd = lambda x, *a, out=None, stats=None, **k: (out if out is not None else x, None, None)
        K.instnorm_lrelu_bwd_data = lambda dy, x, *a, out=None, **k: out if out is not None else dy
    if "rgb" in skip:
        f1 = real["conv2d_fwd"]; K.conv2d_fwd = lambda x, w, *a, out=None, f=f1, **k: out if (w.shape[3] == 3 or w.shape[2] == 3) and out is not None else f(x, w, *a, out=out, **k)
        f2 = real["conv2d_bwd_data"]; K.conv2d_bwd_data = lambda dy, w, hw, *a, out=None, f=f2, **k: out if (w.shape[3] == 3 or w.shape[2] == 3) and out is not None else f(dy, w, hw, *a, out=out, **k)
    engines = [RefineEngine(arch, P, B, dev, use_graph=True) for _ in range(n_streams)]
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


for ns in (2, 1):
    base = run((), ns)
    print(f"{ns} in flight: full step {base:.2f} ms = {B / base * 1e3:.1f} samples/s")
    for name, sk in (("without instance-norm passes", ("norm",)), ("without the RGB head / stem layers", ("rgb",)), ("without both", ("norm", "rgb"))):
        t = run(sk, ns)
        print(f"   {name:36s} {t:8.2f} ms  ({base - t:+.2f} ms, {100 * (base - t) / base:.1f} %)")
