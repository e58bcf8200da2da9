#!/usr/bin/env python3
"""Whole-step A/B of an experiment build's switches in ONE process (development aid, GPU box):
    CGS_LIB=.../libcgs_exp.so LB_AB="CGS_TAIL=0;CGS_TAIL=1" python tools/step_ab.py arch [batch] [fuse]
One hipGraph engine per mode (the switch is read while the graph is captured), replayed alternately, LB_REPS rounds of LB_ITERS steps,
median ms per step per mode.  (Set the switches that size workspaces -- CGS_TAIL -- to their larger setting in the environment of the call.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cgs_amd import nets
from cgs_amd.engine import RefineEngine

arch = sys.argv[1] if len(sys.argv) > 1 else "mnist"
B = int(sys.argv[2]) if len(sys.argv) > 2 else {"dcgan64": 1024, "dcgan32": 256, "mnist": 64, "cyclegan256": 8}[arch]
G = int(sys.argv[3]) if len(sys.argv) > 3 else {"dcgan32": 8, "mnist": 32}.get(arch, 1)
K = 50 if arch == "mnist" else 20
modes = [m for m in os.environ.get("LB_AB", "").split(";") if m] or [""]
reps, iters = int(os.environ.get("LB_REPS", "5")), int(os.environ.get("LB_ITERS", "4"))
d = torch.device("cuda:0")
P = nets.init_params(arch, d, seed=2019)
z = torch.from_numpy(np.random.RandomState(0).uniform(-1, 1, (B * G,) + nets.g_input_shape(nets.ARCHS[arch])).astype(np.float32)).to(d)


def set_mode(m):
    for kv in m.split(","):
        if kv:
            k, v = kv.split("=")
            if v == "-":                      # "-" unsets the variable
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# (workspace sizes are asked of the library once per call signature and cached: ask them under the environment of the call, before any mode
# narrows it)
RefineEngine(arch, P, B * G, d, bn_groups=G, contraction=os.environ.get("CGS_CONTRACTION", "f32")).refine_from_z(z, 1, 0.1)
torch.cuda.synchronize()
engines = []
for m in modes:
    set_mode(m)
    e = RefineEngine(arch, P, B * G, d, use_graph=True, bn_groups=G, contraction=os.environ.get("CGS_CONTRACTION", "f32"))
    e.refine_from_z(z, K, 0.1); e.refine_from_z(z, K, 0.1)
    torch.cuda.synchronize()
    engines.append(e)
times = [[] for _ in modes]
for _ in range(reps):
    for i, e in enumerate(engines):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            e.refine_from_z(z, K, 0.1)
        e1.record(); torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) / iters)
for m, t in zip(modes, times):
    t = sorted(t)
    print(f"{arch} {G} x {B} K={K}  {m or '(default)':28s} median {t[len(t) // 2]:8.3f} ms/step  (min {t[0]:.3f}, max {t[-1]:.3f})  {B * G / t[len(t) // 2] * 1e3:9.1f} samples/s one call in flight")
