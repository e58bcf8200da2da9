#!/usr/bin/env python3
"""Do the engine calls in flight overlap better as parallel BRANCHES of one hipGraph than as separate graphs on separate streams?
(A kernel trace of the default mnist run shows the four graphs executed one after the other: runs of ~1460 dispatches of one queue.)
    python tools/multi_graph_probe.py [arch ...]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cgs_amd import nets
from cgs_amd.engine import RefineEngine

dev = torch.device("cuda:0")
CFG = {"mnist": (64, 50, 32, 4), "dcgan32": (256, 20, 8, 4), "cyclegan256": (8, 20, 1, 4), "dcgan64": (1024, 20, 1, 2)}
ROUNDS = int(os.environ.get("ROUNDS", "4"))


def separate(arch, nf):
    B, Ks, G, _ = CFG[arch]
    A = nets.ARCHS[arch]; P = nets.init_params(arch, dev, seed=2019)
    n = ROUNDS * nf
    z = torch.from_numpy(np.random.RandomState(2019).uniform(-1, 1, (n + nf, B * G) + nets.g_input_shape(A)).astype(np.float32)).to(dev)
    engines = [RefineEngine(arch, P, B * G, dev, use_graph=True, bn_groups=G) for _ in range(nf)]
    streams = [torch.cuda.Stream(dev) for _ in engines]
    def step(i):
        with torch.cuda.stream(streams[i % nf]):
            engines[i % nf].refine_from_z(z[i], Ks, 0.1)
    for i in range(nf): step(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(nf, n + nf): step(i)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return B * G * n / dt


def branches(arch, nf):
    B, Ks, G, _ = CFG[arch]
    A = nets.ARCHS[arch]; P = nets.init_params(arch, dev, seed=2019)
    n = ROUNDS * nf
    z = torch.from_numpy(np.random.RandomState(2019).uniform(-1, 1, (n + nf, B * G) + nets.g_input_shape(A)).astype(np.float32)).to(dev)
    engines = [RefineEngine(arch, P, B * G, dev, use_graph=False, bn_groups=G) for _ in range(nf)]
    main = torch.cuda.Stream(dev)
    side = [torch.cuda.Stream(dev) for _ in engines]
    zin = [torch.empty_like(z[0]) for _ in engines]
    for j, (e, s) in enumerate(zip(engines, side)):          # eager warm-up on the branch's own stream: packs its workspaces
        with torch.cuda.stream(s):
            zin[j].copy_(z[j]); e.refine_from_z(zin[j], Ks, 0.1)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main, capture_error_mode="thread_local"):
        for j, (e, s) in enumerate(zip(engines, side)):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                e.refine_from_z(zin[j], Ks, 0.1)
        for s in side:
            main.wait_stream(s)
    def launch(r):
        with torch.cuda.stream(main):
            for j in range(nf):
                zin[j].copy_(z[nf + r * nf + j])
            g.replay()
    launch(0)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for r in range(ROUNDS): launch(r)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return B * G * n / dt


for arch in (sys.argv[1:] or list(CFG)):
    nf = CFG[arch][3]
    a = separate(arch, nf); torch.cuda.empty_cache()
    b = branches(arch, nf); torch.cuda.empty_cache()
    b2 = branches(arch, 2 * nf) if arch != "dcgan64" else branches(arch, 3); torch.cuda.empty_cache()
    print(f"{arch}: {nf} separate graphs on {nf} streams {a:.1f} samples/s | one graph with {nf} branches {b:.1f} | with {2 * nf if arch != 'dcgan64' else 3} branches {b2:.1f}", flush=True)
