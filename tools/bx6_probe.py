"""Diagnostic: where a block of the split-bf16 implicit GEMM spends its K loop (needs a -DBX6_STAMPS build: tools/build_variant.sh
stamps igemm_bx6.hip "-DBX6_STAMPS"; CGS_LIB=.../libcgs_stamps.so python tools/bx6_probe.py).  Per stage: issue (addresses + global loads),
compute (fragment reads + 48 MFMAs), store (split + LDS stores of the next stage), barrier; cycles of wave 0, summed over the block's stages."""
import os, sys, torch, numpy as np
os.environ["CGS_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cgs_amd import kernels as K, lib as L
L.set_contraction("bx6")
d = torch.device("cuda:0")
B = 1024
NBLK = 16384          # the library's diagnostic builds stamp into the last MiB of the workspace: 16384 blocks x 64 bytes


def probe(kind, H, Ci, Co):
    op = {"conv_fwd": L.CONV_FWD, "conv_bwd": L.CONV_BWD_DATA, "deconv_fwd": L.DECONV_FWD, "deconv_bwd": L.DECONV_BWD_DATA}[kind]
    deconv = kind.startswith("deconv")
    w = torch.randn((5, 5, Co, Ci) if deconv else (5, 5, Ci, Co), device=d) * 0.02
    nb = L.conv_ws_bytes(op, 5, 5, 2, 2, Ci, Co)
    ws = torch.zeros(nb // 4 + NBLK * 16 + 64, device=d)
    tail = (ws.numel() * 4 - (1 << 20) - ((ws.data_ptr() + ws.numel() * 4) & 15)) // 4      # float offset of the stamp area: the last MiB of the workspace
    s = torch.cuda.current_stream().cuda_stream
    if kind == "conv_fwd":
        x = torch.randn(B, H, H, Ci, device=d); y = torch.empty(B, H // 2, H // 2, Co, device=d); b = torch.zeros(Co, device=d)
        run = lambda pre: L.call("cgs_conv2d_nhwc_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, H, Ci, Co, 5, 5, 2, 2, 0, None, None, ws.data_ptr(), ws.numel() * 4, pre, s)
    elif kind == "conv_bwd":
        dy = torch.randn(B, H // 2, H // 2, Co, device=d); dx = torch.empty(B, H, H, Ci, device=d)
        run = lambda pre: L.call("cgs_conv2d_nhwc_bwd_data", dy.data_ptr(), w.data_ptr(), dx.data_ptr(), B, H, H, Ci, Co, 5, 5, 2, 2, 0, None, None, ws.data_ptr(), ws.numel() * 4, pre, s)
    elif kind == "deconv_fwd":
        x = torch.randn(B, H, H, Ci, device=d); y = torch.empty(B, 2 * H, 2 * H, Co, device=d); b = torch.zeros(Co, device=d)
        run = lambda pre: L.call("cgs_deconv2d_nhwc_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, H, Ci, 2 * H, 2 * H, Co, 5, 5, 2, 2, 0, None, None, ws.data_ptr(), ws.numel() * 4, pre, s)
    else:
        dy = torch.randn(B, 2 * H, 2 * H, Co, device=d); dx = torch.empty(B, H, H, Ci, device=d)
        run = lambda pre: L.call("cgs_deconv2d_nhwc_bwd_data", dy.data_ptr(), w.data_ptr(), dx.data_ptr(), B, H, H, Ci, 2 * H, 2 * H, Co, 5, 5, 2, 2, 0, None, None, ws.data_ptr(), ws.numel() * 4, pre, s)
    run(0)
    for _ in range(20): run(1)
    torch.cuda.synchronize()
    ws[nb // 4:].zero_()
    run(1); torch.cuda.synchronize()
    raw = ws[tail: tail + NBLK * 16].view(torch.int64).cpu().numpy().reshape(NBLK, 8)
    raw = raw[raw[:, 1] != 0]
    loop_us = (raw[:, 1] - raw[:, 0]) / 100.0
    ph = raw[:, 2:6].astype(np.float64)
    tot = ph.sum(1)
    # executed stages are not stamped separately: estimate from the phase sums / median stage time is circular; report per block totals
    print(f"== {kind} {H}x{H} {Ci}->{Co}: {L.last_kernel()}  blocks {len(raw)}; K loop us: min {loop_us.min():.1f} median {np.median(loop_us):.1f} max {loop_us.max():.1f}; "
          f"clock {np.median(tot / loop_us) / 1e3:.2f} GHz")
    sh = ph / tot[:, None]
    print("   share of the K loop (median over blocks): issue %.3f  compute %.3f  store %.3f  barrier %.3f" % tuple(np.median(sh, 0)))
    heavy = loop_us >= np.percentile(loop_us, 90)
    print("   ... of the 10 %% longest blocks:             issue %.3f  compute %.3f  store %.3f  barrier %.3f" % tuple(np.median(sh[heavy], 0)))


for a in (("conv_fwd", 16, 128, 256), ("conv_bwd", 16, 128, 256), ("conv_fwd", 8, 256, 512), ("deconv_fwd", 16, 128, 64)):
    probe(*a)
