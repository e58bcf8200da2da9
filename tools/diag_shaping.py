#!/usr/bin/env python3
"""Per-parameter error of DShaper.loss_and_grads against float64 autograd on the oracle D (development aid, GPU box).
    python tools/diag_shaping.py [arch] [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import nets_ref as N
from oracle import ops_ref as R
from cgs_amd import kernels as K
from cgs_amd.nets import to_device
from cgs_amd.shaping import DShaper

arch = sys.argv[1] if len(sys.argv) > 1 else "dcgan64"
Bs = [int(v) for v in sys.argv[2:]] or [8, 64]
d = torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).float()


bce = torch.nn.functional.binary_cross_entropy_with_logits
for B in Bs:
    P = N.init_params(arch, 2019, True)
    real = rnd((B,) + tuple(N.ARCHS[arch]["img"]), 1).clamp(-1, 1)
    fake = torch.tanh(rnd((B,) + tuple(N.ARCHS[arch]["img"]), 2))
    Pg = {k: (v.clone().double().requires_grad_(True) if k.startswith("discriminator/") and "moving" not in k else v.double()) for k, v in P.items()}
    lr = N.discriminator(arch, Pg, real.double()); lf = N.discriminator(arch, Pg, fake.double())
    (bce(lr, torch.ones_like(lr)) + bce(lf, torch.zeros_like(lf))).backward()
    Pd = to_device(P, d)
    sh = DShaper(arch, Pd, B, d, learning_rate=1e-3)
    sh.loss_and_grads(real.to(d), fake.to(d))
    names = [k for k in Pg if Pg[k].requires_grad]
    print(f"== {arch} B={B}")
    for st in sh.tape.stages:
        for attr in ("w", "b", "gamma", "beta"):
            if hasattr(st, "g_" + attr):
                p = getattr(st, attr)
                name = [k for k in names if Pd[k] is p][0]
                g, ref = getattr(st, "g_" + attr).cpu().double(), Pg[name].grad
                print(f"  {name:36s} err {(g - ref).abs().max().item():.3e}  max|ref| {ref.abs().max().item():.3e}  rel {(g - ref).abs().max().item() / (ref.abs().max().item() + 1e-30):.2e}")
    # the first conv's weight gradient stand-alone, on the oracle's own dy
    H, Ci = N.ARCHS[arch]["img"][0], N.ARCHS[arch]["img"][2]
    k = N.ARCHS[arch]["k"]
    x = rnd((B, H, H, Ci), 1)
    w = rnd((k, k, Ci, 64), 2, 0.05).double().requires_grad_(True)
    y = R.conv2d(x.double(), w, torch.zeros(64, dtype=torch.float64), 2, 2)
    dy = rnd(tuple(y.shape), 3)
    (y * dy.double()).sum().backward()
    gw = K.conv2d_bwd_weight(x.to(d), dy.to(d), k, k, 2, 2).cpu().double()
    print(f"  stand-alone conv2d_bwd_weight {H}x{H} {Ci}->64: rel {(gw - w.grad).abs().max().item() / w.grad.abs().max().item():.2e}")
