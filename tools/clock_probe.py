"""Diagnostic: per-block timeline of one igemm launch (prologue / K loop / epilogue, which CU ran it, how many K tiles it
executed) from in-kernel s_memrealtime stamps.  Needs a DIAGNOSTIC build of the library (never the shipped one), e.g.
    hipcc ... -DCGS_DIAG_STAMPS -c igemm.hip ; link as libcgs_hip_diag.so ; CGS_LIB=.../libcgs_hip_diag.so python tools/clock_probe.py
Layers: kind:H:Cin:Cout[:k] with kind in conv_fwd | conv_bwd | deconv_fwd | deconv_bwd   (stride 2, k = 5 by default; PROBE_B images, default 1024)."""
import os, sys, torch, numpy as np
os.environ["CGS_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cgs_amd import kernels as K, lib as L
d = torch.device("cuda:0")
B = int(os.environ.get("PROBE_B", "1024"))
NBLK = 16384          # the library's diagnostic builds stamp into the last MiB of the workspace: 16384 blocks x 64 bytes


def probe(kind, H, Ci, Co, KS=5):
    op = {"conv_fwd": L.CONV_FWD, "conv_bwd": L.CONV_BWD_DATA, "deconv_fwd": L.DECONV_FWD, "deconv_bwd": L.DECONV_BWD_DATA}[kind]
    deconv = kind.startswith("deconv")
    wshape = (KS, KS, Co, Ci) if deconv else (KS, KS, Ci, Co)
    w = torch.randn(wshape, device=d) * 0.02
    nb = L.conv_ws_bytes(op, KS, KS, 2, 2, Ci, Co)
    ws = torch.zeros(nb // 4 + NBLK * 16 + 64, device=d)
    tail = (ws.numel() * 4 - (1 << 20) - ((ws.data_ptr() + ws.numel() * 4) & 15)) // 4      # float offset of the stamp area: the last MiB of the workspace
    s = torch.cuda.current_stream().cuda_stream
    if kind == "conv_fwd":
        x = torch.randn(B, H, H, Ci, device=d); y = torch.empty(B, H // 2, H // 2, Co, device=d); b = torch.zeros(Co, device=d)
        run = lambda pre: L.call("cgs_conv2d_nhwc_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, H, Ci, Co, KS, KS, 2, 2, 0, None, None, ws.data_ptr(), ws.numel() * 4, pre, s)
    elif kind == "conv_bwd":
        dy = torch.randn(B, H // 2, H // 2, Co, device=d); dx = torch.empty(B, H, H, Ci, device=d)
        run = lambda pre: L.call("cgs_conv2d_nhwc_bwd_data", dy.data_ptr(), w.data_ptr(), dx.data_ptr(), B, H, H, Ci, Co, KS, KS, 2, 2, 0, None, None, ws.data_ptr(), ws.numel() * 4, pre, s)
    elif kind == "deconv_fwd":
        x = torch.randn(B, H, H, Ci, device=d); y = torch.empty(B, 2 * H, 2 * H, Co, device=d); b = torch.zeros(Co, device=d)
        run = lambda pre: L.call("cgs_deconv2d_nhwc_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, H, Ci, 2 * H, 2 * H, Co, KS, KS, 2, 2, 0, None, None, ws.data_ptr(), ws.numel() * 4, pre, s)
    else:
        dy = torch.randn(B, 2 * H, 2 * H, Co, device=d); dx = torch.empty(B, H, H, Ci, device=d)
        run = lambda pre: L.call("cgs_deconv2d_nhwc_bwd_data", dy.data_ptr(), w.data_ptr(), dx.data_ptr(), B, H, H, Ci, 2 * H, 2 * H, Co, KS, KS, 2, 2, 0, None, None, ws.data_ptr(), ws.numel() * 4, pre, s)
    run(0)
    for _ in range(30): run(1)
    torch.cuda.synchronize()
    ws[nb // 4:].zero_()
    run(1); torch.cuda.synchronize()
    raw = ws[tail: tail + NBLK * 16].view(torch.int64).cpu().numpy().reshape(NBLK, 8)
    raw = raw[raw[:, 3] != 0]
    n = len(raw)
    t = raw[:, :4].astype(np.float64)
    t = (t - t[:, 0].min()) / 100.0        # us (100 MHz)
    hw = raw[:, 4] & 0xffffffff; xcc = (raw[:, 4] >> 32) & 0xf
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    tiles = raw[:, 5].astype(np.float64)
    span = t[:, 3].max()
    print(f"== {kind} {H}x{H} {Ci}->{Co} B={B}: {L.last_kernel()}  blocks {n}  span {span:.1f} us")
    print("   block medians: prologue %.1f  loop %.1f  epilogue %.1f  total %.1f us; K tiles/block min %d median %d max %d" % (
        np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1]), np.median(t[:, 3] - t[:, 2]), np.median(t[:, 3] - t[:, 0]),
        tiles.min(), np.median(tiles), tiles.max()))
    ucu = np.unique(cuid)
    per_cu_tiles = np.array([tiles[cuid == c].sum() for c in ucu]); per_cu_end = np.array([t[cuid == c, 3].max() for c in ucu])
    per_cu_n = np.array([(cuid == c).sum() for c in ucu])
    print(f"   distinct CUs {len(ucu)}; blocks per CU min {per_cu_n.min()} max {per_cu_n.max()}; K tiles per CU min {per_cu_tiles.min():.0f} "
          f"mean {per_cu_tiles.mean():.0f} max {per_cu_tiles.max():.0f} (max/mean {per_cu_tiles.max() / per_cu_tiles.mean():.3f}); "
          f"CU finish time min {per_cu_end.min():.1f} median {np.median(per_cu_end):.1f} max {per_cu_end.max():.1f} us")
    # loop speed: us per K tile as a function of how many blocks share the CU at that time is not observable directly; report
    # the per-block loop time per tile
    ptt = (t[:, 2] - t[:, 1]) / np.maximum(tiles, 1)
    print("   loop us per K tile: p10 %.3f median %.3f p90 %.3f" % (np.percentile(ptt, 10), np.median(ptt), np.percentile(ptt, 90)))
    grid = np.linspace(0, span, 25)
    print("   resident blocks over time:", [int(((t[:, 0] <= g) & (t[:, 3] > g)).sum()) for g in grid])
    first = np.argsort(t[:, 0])[:12]
    print("   first dispatched blocks -> CU ids:", [int(cuid[i]) for i in first], "start us", [round(float(t[i, 0]), 1) for i in first])


for spec in (sys.argv[1:] or ["conv_fwd:16:128:256", "conv_fwd:32:64:128", "deconv_fwd:16:128:64", "conv_fwd:8:256:512"]):
    f = spec.split(":")            # kind:H:Cin:Cout[:kernel size]   (stride 2)
    probe(f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4]) if len(f) > 4 else 5)
