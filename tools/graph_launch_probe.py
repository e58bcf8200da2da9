#!/usr/bin/env python3
"""How long does the HOST spend in one hipGraph replay of a K-step program, and is it asynchronous?  (mnist 32 x 64, K = 50: ~1460 kernel nodes)
    python tools/graph_launch_probe.py [arch]"""
import os, sys, time, threading
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cgs_amd import nets
from cgs_amd.engine import RefineEngine

dev = torch.device("cuda:0")
arch = sys.argv[1] if len(sys.argv) > 1 else "mnist"
B, Ks, G = {"mnist": (64, 50, 32), "dcgan32": (256, 20, 8), "cyclegan256": (8, 20, 1), "dcgan64": (1024, 20, 1)}[arch]
nf = 4
A = nets.ARCHS[arch]; P = nets.init_params(arch, dev, seed=2019)
z = torch.from_numpy(np.random.RandomState(2019).uniform(-1, 1, (nf, B * G) + nets.g_input_shape(A)).astype(np.float32)).to(dev)
engines = [RefineEngine(arch, P, B * G, dev, use_graph=True, bn_groups=G) for _ in range(nf)]
streams = [torch.cuda.Stream(dev) for _ in engines]
for j in range(nf):
    with torch.cuda.stream(streams[j]):
        engines[j].refine_from_z(z[j], Ks, 0.1)
torch.cuda.synchronize(dev)
# one call alone: host time of the launch vs the GPU time of the program
t0 = time.perf_counter()
with torch.cuda.stream(streams[0]):
    engines[0].refine_from_z(z[0], Ks, 0.1)
t1 = time.perf_counter()
torch.cuda.synchronize(dev)
t2 = time.perf_counter()
print(f"{arch}: one replay: host returns after {1e3 * (t1 - t0):.2f} ms, GPU done after {1e3 * (t2 - t0):.2f} ms")
# four calls back to back from one host thread
t0 = time.perf_counter(); marks = []
for j in range(nf):
    with torch.cuda.stream(streams[j]):
        engines[j].refine_from_z(z[j], Ks, 0.1)
    marks.append(time.perf_counter() - t0)
torch.cuda.synchronize(dev)
t2 = time.perf_counter() - t0
print(f"   four replays, one host thread: launches returned at {[round(1e3 * m, 2) for m in marks]} ms, all done after {1e3 * t2:.2f} ms")
# four calls from four host threads
def work(j):
    with torch.cuda.stream(streams[j]):
        engines[j].refine_from_z(z[j], Ks, 0.1)
for rep in range(2):
    th = [threading.Thread(target=work, args=(j,)) for j in range(nf)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    t1 = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter() - t0
print(f"   four replays, four host threads: all launches returned after {1e3 * t1:.2f} ms, all done after {1e3 * t2:.2f} ms")
