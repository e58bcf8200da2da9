"""Diagnostic: per-block timeline of one igemm launch (prologue / K loop / epilogue, rounds, tail) from in-kernel
s_memrealtime stamps.  Needs a DIAGNOSTIC build of the library (never the shipped one):
    make -C collaborative-gan-sampling_amd/csrc -B CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -DCGS_DIAG_STAMPS"
    python tools/clock_probe.py ; then rebuild normally (make -B)."""
import os, sys, torch, numpy as np
os.environ["CGS_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cgs_amd import kernels as K, lib as L
d = torch.device("cuda:0")
B, H, Ci, Co = 1024, 32, 64, 128
x = torch.randn(B, H, H, Ci, device=d); w = torch.randn(5, 5, Ci, Co, device=d) * 0.02; b = torch.zeros(Co, device=d)
nb = L.conv_ws_bytes(L.CONV_FWD, 5, 5, 2, 2, Ci, Co)
NBLK = 2048
ws = torch.zeros(nb // 4 + NBLK * 8 + 64, device=d)
y = torch.empty(B, H // 2, H // 2, Co, device=d)
def run(pre):
    L.call("cgs_conv2d_nhwc_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, H, Ci, Co, 5, 5, 2, 2, 0, None, None,
           ws.data_ptr(), ws.numel() * 4, pre, torch.cuda.current_stream().cuda_stream)
run(0)
for _ in range(300): run(1)
torch.cuda.synchronize()
run(1); torch.cuda.synchronize()
t = ws[nb // 4: nb // 4 + NBLK * 8].view(torch.int64).cpu().numpy().reshape(NBLK, 4).astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0) / 100.0        # us (100 MHz)
print("kernel span %.1f us; block: prologue %.1f  loop %.1f  epilogue %.1f  total %.1f us (medians)" % (
    t[:, 3].max(), np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1]), np.median(t[:, 3] - t[:, 2]), np.median(t[:, 3] - t[:, 0])))
starts = np.sort(t[:, 0]); ends = np.sort(t[:, 3])
for q in (0, 511, 512, 1023, 1024, 1535, 1536, 2047):
    print("start[%d] = %.1f us   end[%d] = %.1f us" % (q, starts[q], q, ends[q]))
# concurrency profile
grid = np.linspace(0, t[:, 3].max(), 40)
print("resident blocks over time:", [int(((t[:, 0] <= g) & (t[:, 3] > g)).sum()) for g in grid])
