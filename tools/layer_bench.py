#!/usr/bin/env python3
"""Per-layer timing of the conv-family kernels at a BASELINE config (development aid, GPU box).
    python tools/layer_bench.py [arch] [B]
Prints ms and TFLOP/s per layer and direction, hipEvent-timed on the current stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgs_amd import kernels as K, lib

K.set_contraction(os.environ.get("CGS_CONTRACTION", "f32"))      # f32 | bx6 | bx6_all (include/cgs_hip.h, cgs_set_contraction)

arch = sys.argv[1] if len(sys.argv) > 1 else "dcgan64"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
d = torch.device("cuda:0")
s = {"dcgan64": 64, "dcgan32": 32}[arch]
convs = [(s, 3, 64), (s // 2, 64, 128), (s // 4, 128, 256), (s // 8, 256, 512)]          # H, Cin, Cout
deconvs = [(s // 8, 256, 128), (s // 4, 128, 64), (s // 2, 64, 3)]                       # H, Cin, Cout


def timeit(fn, n=int(os.environ.get("LB_ITERS", "5"))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


# LB_SLAB=1: carve every activation tensor from ONE big allocation at 2 MiB-aligned offsets (placement / page-size experiment)
_slab = torch.empty(6 << 30, dtype=torch.uint8, device=d) if os.environ.get("LB_SLAB") else None
_off = [0]


def alloc(*shape):
    if _slab is None:
        return torch.empty(shape, device=d)
    n = 4
    for v in shape: n *= v
    o = _off[0]; _off[0] = (o + n + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    return _slab[o:o + n].view(torch.float32).view(shape)


tot = 0.0
for H, Ci, Co in convs:
    x = alloc(B, H, H, Ci).normal_(); w = torch.randn(5, 5, Ci, Co, device=d) * 0.02; b = torch.zeros(Co, device=d)
    y = K.conv2d_fwd(x, w, b, out=alloc(B, H // 2, H // 2, Co)); dy = alloc(*y.shape).normal_()
    fl = 2.0 * B * (H // 2) ** 2 * Co * 25 * Ci
    t1 = timeit(lambda: K.conv2d_fwd(x, w, b, out=y)); t2 = timeit(lambda: K.conv2d_bwd_data(dy, w, (H, H), out=x))
    K.conv2d_fwd(x, w, b, out=y); kf = lib.last_kernel(); K.conv2d_bwd_data(dy, w, (H, H), out=x); kb = lib.last_kernel()
    print(f"conv   {H:3d}x{H:<3d} {Ci:4d}->{Co:<4d} fwd {t1:8.3f} ms {fl/t1/1e9:7.1f} TF | bwd {t2:8.3f} ms {fl/t2/1e9:7.1f} TF   {kf.split('<')[0]} | {kb.split('<')[0]}")
    tot += t1 + t2
for H, Ci, Co in deconvs:
    x = alloc(B, H, H, Ci).normal_(); w = torch.randn(5, 5, Co, Ci, device=d) * 0.02; b = torch.zeros(Co, device=d)
    y = K.deconv2d_fwd(x, w, b, (2 * H, 2 * H), out=alloc(B, 2 * H, 2 * H, Co)); dy = alloc(*y.shape).normal_()
    fl = 2.0 * B * H * H * Ci * 25 * Co
    t1 = timeit(lambda: K.deconv2d_fwd(x, w, b, (2 * H, 2 * H), out=y)); t2 = timeit(lambda: K.deconv2d_bwd_data(dy, w, (H, H), out=x))
    K.deconv2d_fwd(x, w, b, (2 * H, 2 * H), out=y); kf = lib.last_kernel(); K.deconv2d_bwd_data(dy, w, (H, H), out=x); kb = lib.last_kernel()
    print(f"deconv {H:3d}x{H:<3d} {Ci:4d}->{Co:<4d} fwd {t1:8.3f} ms {fl/t1/1e9:7.1f} TF | bwd {t2:8.3f} ms {fl/t2/1e9:7.1f} TF   {kf.split('<')[0]} | {kb.split('<')[0]}")
    tot += t1 + t2
print(f"sum fwd+bwd of all conv-family layers: {tot:.2f} ms for B={B}")
if os.environ.get("LB_EPI"):
    # the 3-channel layers with the epilogues the refinement engine really fuses into them
    H = s
    x = torch.randn(B, H, H, 3, device=d).tanh(); w = torch.randn(5, 5, 3, 64, device=d) * 0.02; b = torch.zeros(64, device=d)
    y = K.conv2d_fwd(x, w, b, 2, 2, lib.EPI_LRELU); dy = torch.randn_like(y); dxs = torch.empty_like(x)
    t1 = timeit(lambda: K.conv2d_fwd(x, w, b, 2, 2, lib.EPI_LRELU, out=y))
    t2 = timeit(lambda: K.conv2d_bwd_data(dy, w, (H, H), out=dxs, epilogue=lib.EPI_TANH_BWD, ep_aux=x))
    print(f"d_h0 conv {H}x{H} 3->64: fwd+lrelu {t1*1e3:7.1f} us ({lib.last_kernel()}) | bwd-data+tanh' {t2*1e3:7.1f} us")
    H2 = s // 2
    xg = torch.randn(B, H2, H2, 64, device=d).relu(); wg = torch.randn(5, 5, 3, 64, device=d) * 0.02; bg = torch.zeros(3, device=d)
    a = torch.rand(64, device=d) + 0.5
    img = K.deconv2d_fwd(xg, wg, bg, (H, H), 2, 2, lib.EPI_TANH); dimg = torch.randn_like(img); dxg = torch.empty_like(xg)
    t3 = timeit(lambda: K.deconv2d_fwd(xg, wg, bg, (H, H), 2, 2, lib.EPI_TANH, out=img))
    t4 = timeit(lambda: K.deconv2d_bwd_data(dimg, wg, (H2, H2), 2, 2, out=dxg, epilogue=lib.EPI_RELU_BWD_AFFINE, ep_a=a, ep_aux=xg))
    print(f"g_h4 deconv {H2}x{H2} 64->3: fwd+tanh {t3*1e3:7.1f} us | bwd-data+relu'*a {t4*1e3:7.1f} us ({lib.last_kernel()})")
if os.environ.get("LB_AUX"):
    # the two implicit-GEMM layers whose backward-data carries an aux epilogue in the engine: with and without it
    H = s // 2
    w = torch.randn(5, 5, 64, 128, device=d) * 0.02
    dy = torch.randn(B, H // 2, H // 2, 128, device=d); dx = torch.empty(B, H, H, 64, device=d); aux = torch.randn(B, H, H, 64, device=d)
    t0 = timeit(lambda: K.conv2d_bwd_data(dy, w, (H, H), out=dx))
    t1 = timeit(lambda: K.conv2d_bwd_data(dy, w, (H, H), out=dx, epilogue=lib.EPI_LRELU_BWD, ep_aux=aux))
    print(f"d_h1 bwd-data 32x32 64<-128: plain {t0*1e3:7.1f} us | + lrelu' (aux) {t1*1e3:7.1f} us ({lib.last_kernel()})")
    wg = torch.randn(5, 5, 64, 128, device=d) * 0.02          # g_h3: deconv 16x16x128 -> 32x32x64
    dyg = torch.randn(B, H, H, 64, device=d); dxg = torch.empty(B, H // 2, H // 2, 128, device=d); auxg = torch.randn(B, H // 2, H // 2, 128, device=d)
    a = torch.rand(128, device=d) + 0.5
    t2 = timeit(lambda: K.deconv2d_bwd_data(dyg, wg, (H // 2, H // 2), 2, 2, out=dxg))
    t3 = timeit(lambda: K.deconv2d_bwd_data(dyg, wg, (H // 2, H // 2), 2, 2, out=dxg, epilogue=lib.EPI_RELU_BWD_AFFINE, ep_a=a, ep_aux=auxg))
    print(f"g_h3 bwd-data 16x16 128<-64: plain {t2*1e3:7.1f} us | + relu'*a (aux) {t3*1e3:7.1f} us ({lib.last_kernel()})")
    xg = torch.randn(B, H // 2, H // 2, 128, device=d); bg = torch.zeros(64, device=d); yg = torch.empty(B, H, H, 64, device=d); cc = torch.zeros(64, device=d); a64 = torch.rand(64, device=d) + 0.5
    t4 = timeit(lambda: K.deconv2d_fwd(xg, wg, bg, (H, H), 2, 2, out=yg))
    t5 = timeit(lambda: K.deconv2d_fwd(xg, wg, bg, (H, H), 2, 2, lib.EPI_AFFINE_RELU, a64, cc, out=yg))
    print(f"g_h3 fwd 16x16 128->64: plain {t4*1e3:7.1f} us | + bn-affine + relu {t5*1e3:7.1f} us")
