"""Diagnostic: per-tile phase timing inside conv_patch2_kernel (needs a -DCGS_PATCH_STAMPS build of conv_patch.hip as CGS_LIB)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cgs_amd import kernels as K, lib as L
d = torch.device("cuda:0")
B, H = 1024, 64
x = torch.randn(B, H, H, 3, device=d); w = torch.randn(5, 5, 3, 64, device=d) * 0.02; b = torch.zeros(64, device=d)
dbg = torch.zeros(64 * 4 * 16 * 8, dtype=torch.int64, device=d)
y = torch.empty(B, 32, 32, 64, device=d)
for _ in range(5): K.conv2d_fwd(x, w, b, 2, 2, L.EPI_LRELU, ep_b=dbg.view(torch.float32), out=y)
torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(64, 4, 16, 8).astype(np.float64)
ok = t[..., 5] > 0
print("tiles recorded", ok.sum())
names = ["barrier wait", "decode+issue patch loads", "(gap)", "MFMA phase", "epilogue", "wait+store next patch"]
for i, nm in enumerate(["barrier wait", "decode + patch loads issue", "MFMA phase", "wait + store patch", "epilogue"]):
    dcy = (t[..., i + 1] - t[..., i])[ok]
    print(f"{nm:28s} median {np.median(dcy):8.0f}  mean {dcy.mean():8.0f}  p90 {np.percentile(dcy, 90):8.0f}  (s_memtime ticks)")
tot = (t[..., 5] - t[..., 0])[ok]
print("tile total median", np.median(tot), "mean", tot.mean())
per = t[:, 0, 1:11, 0] - t[:, 0, 0:10, 0]
print("tile-to-tile period (wave 0) median", np.median(per[per > 0]))
