#!/usr/bin/env python3
"""tools/skip_probe.py for the small configurations at bench.py's defaults (four engine calls in flight, hipGraph replay): what are the
short DEPENDENT launches of a refinement step worth there?  (kernels replaced by no-ops: timing only)   python tools/skip_probe_small.py [arch ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cgs_amd import kernels as K, nets                  # noqa: E402
from cgs_amd.engine import RefineEngine                 # noqa: E402

dev = torch.device("cuda:0")
CFG = {"mnist": (64, 50, 32), "dcgan32": (256, 20, 8), "cyclegan256": (8, 20, 1)}
real = {n: getattr(K, n) for n in dir(K) if callable(getattr(K, n)) and not n.startswith("_")}


def run(arch, skip, nf=4, steps=8):
    B, Ks, G = CFG[arch]
    A = nets.ARCHS[arch]
    P = nets.init_params(arch, dev, seed=2019)
    z = torch.from_numpy(np.random.RandomState(2019).uniform(-1, 1, (steps + nf, B * G) + nets.g_input_shape(A)).astype(np.float32)).to(dev)
    for n, f in real.items():
        setattr(K, n, f)
    if "select_rows" in skip:
        K.refine_select_rows = lambda *a, **k: None
    if "select" in skip:
        K.refine_select = lambda *a, **k: None
    if "update" in skip:
        K.refine_update = lambda *a, **k: None
    if "norm_bwd" in skip:
        K.instnorm_lrelu_bwd_data = lambda dy, x, *a, out=None, **k: out if out is not None else dy
        K.bn_train_lrelu_bwd_data = lambda dy, x, *a, out=None, **k: out if out is not None else dy
    engines = [RefineEngine(arch, P, B * G, dev, use_graph=True, bn_groups=G) for _ in range(nf)]
    streams = [torch.cuda.Stream(dev) for _ in engines]

    def step(i):
        with torch.cuda.stream(streams[i % nf]):
            engines[i % nf].refine_from_z(z[i], Ks, 0.1)
    for i in range(nf):
        step(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(nf, steps + nf):
        step(i)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    del engines
    torch.cuda.empty_cache()
    return dt * 1e3, B * G


for arch in (sys.argv[1:] or list(CFG)):
    base, n = run(arch, ())
    print(f"{arch}: full step {base:.2f} ms = {n / base * 1e3:.1f} samples/s")
    for name, sk in (("without the image row select", ("select_rows",)), ("without both selects", ("select_rows", "select")),
                     ("without selects and momentum update", ("select_rows", "select", "update")), ("without the backward norm passes", ("norm_bwd",))):
        t, _ = run(arch, sk)
        print(f"   {name:40s} {t:8.2f} ms  ({base - t:+.2f} ms, {100 * (base - t) / base:.1f} %)")
