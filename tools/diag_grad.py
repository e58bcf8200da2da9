import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden
from oracle import nets_ref as N, sampling_ref as S, ops_ref as R
from cgs_amd.engine import RefineEngine
from cgs_amd.nets import to_device
from cgs_amd import kernels as K
name = "g3_collab_mnist_K5_probabilistic.npz"
g = load_golden(name); arch = "mnist"
P = {k: v.double() for k, v in N.init_params(arch, 2019, True).items()}
f0 = torch.from_numpy(g["feature0"]).double().requires_grad_(True)
A = N.ARCHS[arch]
gt_groups = [A["g_tail"][0:3], A["g_tail"][3:5]]
d_groups = [A["d"][0:2], A["d"][2:3], A["d"][3:5], A["d"][5:6], A["d"][6:7], A["d"][7:9], A["d"][9:10]]
acts = [f0]
x = f0
for grp in gt_groups:
    x = N.run_layers(grp, x, P, "generator", False); x.retain_grad(); acts.append(x)
for grp in d_groups:
    x = N.run_layers(grp, x, P, "discriminator", True); x.retain_grad(); acts.append(x)
R.sigmoid_xent_ones(x).sum().backward()
d = torch.device("cuda:0")
P32 = to_device({k: v.float() for k, v in P.items()}, d)
eng = RefineEngine(arch, P32, len(g["feature0"]), d)
th = torch.from_numpy(g["feature0"]).to(d)
stages = eng.g_tail.stages + eng.d.stages
xx = th
outs = []
for st in stages:
    xx = st.fwd(xx); outs.append(xx.clone())
rel = lambda a, b: ((a.double().cpu().reshape(-1) - b.reshape(-1)).abs().max() / (b.abs().max() + 1e-300)).item()
for i, (o, a) in enumerate(zip(outs, acts[1:])):
    print("fwd stage %d %-14s err %.2e" % (i, type(stages[i]).__name__, rel(o, a.detach())))
K.bce_ones_grad_rowmean(xx, eng.dlogits, eng.logit)
dy = eng.dlogits
print("seed err %.2e" % rel(dy, acts[-1].grad))
for i in reversed(range(len(stages))):
    dy = stages[i].bwd(dy)
    print("bwd stage %d %-14s dx err %.2e   (max|ref| %.3e)" % (i, type(stages[i]).__name__, rel(dy, acts[i].grad), acts[i].grad.abs().max()))
    if rel(dy, acts[i].grad) > 1e-3:
        diff = (dy.double().cpu().reshape(acts[i].grad.shape) - acts[i].grad).abs()
        idx = torch.nonzero(diff > 1e-3 * acts[i].grad.abs().max())
        print("   bad elements:", idx.shape[0], "first:", idx[:6].tolist(), "per-sample max:", diff.reshape(diff.shape[0], -1).max(1).values.tolist())
        break
