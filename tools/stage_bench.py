#!/usr/bin/env python3
"""Per-stage timing of an engine's contraction stages with their real epilogues, statistics and sign masks (development aid, GPU box).
    python tools/stage_bench.py arch [batch] [fuse]        (defaults: the bench.py defaults of the arch)
Every conv / deconv / linear stage of G tail + D: forward and backward-data, HIP events over LB_ITERS launches on one stream, the kernel the
library picked, issued TFLOP/s (cgs_last_executed_flops) against the fp32 matrix peak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cgs_amd import kernels as K, lib as L, nets
from cgs_amd.engine import RefineEngine, _Conv, _Deconv, _Linear, _Residual, _View

arch = sys.argv[1] if len(sys.argv) > 1 else "mnist"
B = int(sys.argv[2]) if len(sys.argv) > 2 else {"dcgan64": 1024, "dcgan32": 256, "mnist": 64, "cyclegan256": 8}[arch]
G = int(sys.argv[3]) if len(sys.argv) > 3 else {"dcgan32": 8, "mnist": 32}.get(arch, 1)
iters = int(os.environ.get("LB_ITERS", "10"))
d = torch.device("cuda:0")
K.set_contraction(os.environ.get("CGS_CONTRACTION", "f32"))
P = nets.init_params(arch, d, seed=2019)
eng = RefineEngine(arch, P, B * G, d, bn_groups=G, contraction=os.environ.get("CGS_CONTRACTION", "f32"))
z = torch.from_numpy(np.random.RandomState(0).uniform(-1, 1, (B * G,) + nets.g_input_shape(nets.ARCHS[arch])).astype(np.float32)).to(d)
eng.refine_from_z(z, 1, 0.1)          # fills every stage's buffers, packs the weights
K.set_contraction(eng.contraction)


def timeit(fn):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def flat(stages):
    for st in stages:
        if isinstance(st, _Residual):
            yield from flat(st.inner)
        else:
            yield st


# LB_AB="CGS_TAIL=0;CGS_TAIL=1": time every stage under each environment setting of an experiment build (CGS_LIB=.../libcgs_exp.so reads its
# switches at every launch), interleaved in ONE process (clocks and placement differ by up to 10 % between processes), best of LB_REPS rounds
modes = [m for m in os.environ.get("LB_AB", "").split(";") if m] or [""]
reps = int(os.environ.get("LB_REPS", "3"))


def set_mode(m):
    for kv in m.split(","):
        if kv:
            k, v = kv.split("=")
            if v == "-":                      # "-" unsets the variable
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def timed(fn):
    best, info = [1e30] * len(modes), [None] * len(modes)
    for _ in range(reps):
        for i, m in enumerate(modes):
            set_mode(m)
            t = timeit(fn)
            if t < best[i]:
                best[i], info[i] = t, (L.last_kernel(), float(L.load().cgs_last_executed_flops()), int(L.load().cgs_last_tail_tiles()), int(L.load().cgs_last_tail_split()))
    return best, info


def show(best, info):
    out = []
    for t, (k, f, tn, ts) in zip(best, info):
        out.append(f"{t:8.1f} us {f / t / 1e6 / 157.3:.3f}" + (f" tail {tn}x{ts}" if ts else ""))
    return " | ".join(out) + "  " + info[0][0].replace("igemm_ns_kernel", "ig-ns").replace("igemm_kernel", "ig")


tot = [0.0] * len(modes)
print("modes:", modes)
for tape, tname in ((eng.g_tail, "G"), (eng.d, "D")):
    prev = eng.theta if tname == "G" else eng.g_tail.stages[-1].out
    for st in flat(tape.stages):
        if isinstance(st, _View):
            prev = st.fwd(prev)
            continue
        if isinstance(st, (_Conv, _Deconv, _Linear)):
            xin = prev
            dy = torch.randn_like(st.out)
            bf, inf = timed(lambda: st.fwd(xin))
            bb, inb = timed(lambda: st.bwd(dy))
            print(f"{tname} {type(st).__name__[1:]:7s} {str(tuple(xin.shape)):22s} -> {str(tuple(st.out.shape)):22s} fwd {show(bf, inf)}")
            print(f"{'':57s} bwd {show(bb, inb)}")
            tot = [a + b + c for a, b, c in zip(tot, bf, bb)]
        prev = st.out if getattr(st, "out", None) is not None else prev
print("sum of the contraction stages, forward + backward-data (ms): " + " | ".join(f"{t / 1e3:.3f}" for t in tot) + f"   ({arch}, {G} x {B})")
