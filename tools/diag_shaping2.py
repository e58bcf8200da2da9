#!/usr/bin/env python3
"""Stage-by-stage input gradients of the D pass of DShaper against float64 autograd on the oracle (development aid, GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import nets_ref as N
from oracle import ops_ref as R
from cgs_amd import kernels as K, lib as L
from cgs_amd.nets import to_device
from cgs_amd.shaping import DShaper
from cgs_amd.engine import _BnTrainLrelu, _Conv, _Linear, _View

arch = sys.argv[1] if len(sys.argv) > 1 else "dcgan64"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
d = torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).float()


P = N.init_params(arch, 2019, True)
real = rnd((B,) + tuple(N.ARCHS[arch]["img"]), 1).clamp(-1, 1)
Pd64 = {k: v.double() for k, v in P.items()}
# oracle forward in double, keeping every layer boundary
layers = N.ARCHS[arch]["d"]
acts = [real.double().requires_grad_(True)]
x = acts[0]
bounds = []        # index into layers after which an activation is recorded
for i, Lr in enumerate(layers):
    x = N.run_layers([Lr], x, Pd64, "discriminator", bn_training=True)
    x.retain_grad()
    acts.append(x)
logits = acts[-1]
n = logits.numel()
loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, torch.ones_like(logits))
loss.backward()
names = ["input"] + [f"{i}:{Lr[0]}{(':' + Lr[1]) if len(Lr) > 1 and isinstance(Lr[1], str) else ''}" for i, Lr in enumerate(layers)]

sh = DShaper(arch, to_device(P, d), B, d)
lg = sh.tape.forward(real.to(d))
K.bce_logits_grad(lg, 1.0, 1.0 / n, sh.dlogits, sh.loss_buf[0:1])
print("loss", float(sh.loss_buf[0]), float(loss))
dy = sh.dlogits
stages = sh.tape.stages
# map stages to oracle layer indices: walk the layer list the way compile_layers fuses
li = len(layers)
def rel(a, b):
    a, b = a.detach().cpu().double().reshape(-1), b.detach().double().reshape(-1)
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))
# stage -> index of the oracle activation that is the stage's INPUT
idx_in = []
i = 0
for st in stages:
    idx_in.append(i)
    if isinstance(st, _Conv):
        i += 2 if st.epi == L.EPI_LRELU else 1
    elif isinstance(st, _BnTrainLrelu):
        i += 2 if st.leak != 1.0 else 1
    elif isinstance(st, _Linear):
        i += 2 if st.epi == L.EPI_LRELU else 1
    else:
        i += 1
for k in range(len(stages) - 1, -1, -1):
    st = stages[k]
    if isinstance(st, _Conv) and st.epi == L.EPI_LRELU and not st.pre_folded:
        dy = K.lrelu_bwd(dy, st.out, out=dy)
    if isinstance(st, _Linear) and st.epi == L.EPI_LRELU:
        dy = K.lrelu_bwd(dy, st.out, out=dy)
    if isinstance(st, _View):
        dy = st.bwd(dy)
    elif isinstance(st, _BnTrainLrelu):
        dy = st.bwd(dy)
    elif isinstance(st, _Linear):
        dy = K.linear_bwd_data(dy, st.w, out=st.dx)
    elif isinstance(st, _Conv):
        if k == 0:
            break
        e, a, aux = st.bwd_epi
        dy = K.conv2d_bwd_data(dy, st.w, st.in_hw, st.s, st.s, out=st.dx, epilogue=e, ep_a=a, ep_aux=aux)
        print("   kernel:", L.last_kernel(), "family epi", e)
    ref = acts[idx_in[k]].grad
    # when the stage above folded this input's activation gradient in, dy is the gradient wrt the PRE-activation of the stage below
    below = stages[k - 1] if k > 0 else None
    if below is not None and getattr(below, "pre_folded", False):
        ref = acts[idx_in[k] - 1].grad
    print(f"stage {k:2d} {type(st).__name__:14s} d/d(input {names[idx_in[k]]:18s}) rel err {rel(dy, ref):.3e}   shape {tuple(dy.shape)}")

# ---- closer look at the first norm stage whose input gradient is off: saved statistics, saved input, stand-alone recomputation
print("---- norm stages: saved x / mean / invstd against their own definitions, and the backward recomputed from clean inputs")
lg = sh.tape.forward(real.to(d))
K.bce_logits_grad(lg, 1.0, 1.0 / n, sh.dlogits, sh.loss_buf[0:1])
g_in = {}
dy = sh.dlogits
for k in range(len(stages) - 1, 0, -1):
    st = stages[k]
    if isinstance(st, _BnTrainLrelu):
        g_in[k] = dy.clone()
    if isinstance(st, _Linear) and st.epi == L.EPI_LRELU:
        dy = K.lrelu_bwd(dy, st.out, out=dy)
    if isinstance(st, _Conv):
        e, a, aux = st.bwd_epi
        dy = K.conv2d_bwd_data(dy, st.w, st.in_hw, st.s, st.s, out=st.dx, epilogue=e, ep_a=a, ep_aux=aux)
    elif isinstance(st, _Linear):
        dy = K.linear_bwd_data(dy, st.w, out=st.dx)
    else:
        dy = st.bwd(dy)
for k, st in enumerate(stages):
    if not isinstance(st, _BnTrainLrelu):
        continue
    x = st.x
    C = x.shape[-1]
    xf = x.double().reshape(-1, C)
    mu, var = xf.mean(0), xf.var(0, unbiased=False)
    print(f"stage {k}: x vs oracle {rel(x, acts[idx_in[k]]):.2e}; mean err {float((st.mean.double() - mu).abs().max()):.2e}; invstd rel err "
          f"{float(((st.invstd.double() - 1 / torch.sqrt(var + 1e-5)).abs() * torch.sqrt(var + 1e-5)).max()):.2e}; x.data_ptr {x.data_ptr():x} dx(above) {g_in[k].data_ptr():x}")
    clean = K.bn_train_lrelu_bwd_data(g_in[k], x, st.gamma, st.beta, st.mean, st.invstd, st.leak)
    print(f"         backward recomputed stand-alone from the saved inputs: rel err vs oracle {rel(clean, acts[idx_in[k]].grad):.2e}")
    # (a) the float64 formula on the SAVED inputs; (b) the kernel on the ORACLE's incoming gradient
    C_ = x.shape[-1]
    X = x.double().reshape(-1, C_); Gi = g_in[k].double().reshape(-1, C_)
    xh = (X - mu) / torch.sqrt(var + 1e-5)
    u = xh * st.gamma.double() + st.beta.double()
    dp = Gi * torch.where(u > 0, 1.0, float(st.leak))
    want = (st.gamma.double() / torch.sqrt(var + 1e-5)) * (dp - dp.mean(0) - xh * (dp * xh).mean(0))
    print(f"         float64 formula on the saved inputs vs oracle {rel(want.reshape(x.shape), acts[idx_in[k]].grad):.2e}; kernel vs that formula {rel(clean, want.reshape(x.shape).cpu()):.2e}")
    # which oracle activation is the norm's OUTPUT (after its lrelu)?
    j_out = idx_in[k] + (2 if st.leak != 1.0 else 1)
    g_or = acts[j_out].grad.float().to(d).contiguous()
    print(f"         incoming gradient vs oracle's {rel(g_in[k], acts[j_out].grad):.2e}; max|g| {float(acts[j_out].grad.abs().max()):.3e}, max|dx| {float(acts[idx_in[k]].grad.abs().max()):.3e}")
    k2 = K.bn_train_lrelu_bwd_data(g_or, x, st.gamma, st.beta, st.mean, st.invstd, st.leak)
    print(f"         kernel on the oracle's incoming gradient vs oracle {rel(k2, acts[idx_in[k]].grad):.2e}")
    nflip = int(((u > 0).cpu() != (acts[j_out - 1].detach().reshape(-1, C_) > 0)).sum()) if st.leak != 1.0 else -1
    print(f"         lrelu sides that differ from the oracle's: {nflip} of {u.numel()}")
