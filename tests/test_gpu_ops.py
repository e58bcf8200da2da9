"""Parity of every C-ABI kernel (through cgs_amd.kernels -> libcgs_hip.so) against the CPU oracle.
fp32 tolerance: the MFMA contraction is an exact-fp32 k-ordered fma chain, the oracle a torch-CPU
(MKL) contraction in another order -> |delta| <= 2e-5 * sqrt(K) * max|term| is ample; stated per test."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ops_ref as R


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).float()


def close(got, want, tol):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item() + 1e-30
    assert err <= tol * ref, f"max|delta|={err:.3e} vs max|ref|={ref:.3e} (tol {tol})"


CONV_CASES = [  # B,H,W,Cin,Cout,k,s
    (3, 16, 16, 64, 128, 5, 2),      # vec path, BN=128
    (2, 8, 8, 128, 64, 5, 2),        # vec path, BN=64
    (5, 28, 28, 1, 64, 4, 2),        # mnist d_conv1: scalar path K=16
    (2, 14, 14, 64, 128, 4, 2),      # mnist d_conv2
    (3, 32, 32, 3, 64, 5, 2),        # dcgan d_h0: scalar path K=75
    (2, 7, 9, 32, 40, 5, 2),         # odd sizes, N not a multiple of 64
    (1, 6, 5, 32, 64, 3, 1),         # stride 1
    (70, 4, 4, 256, 512, 5, 2),      # many images, M not a multiple of 128
    (130, 8, 8, 64, 128, 5, 2),      # B >= 128, 4x4 grid: pixel-major rows + zero-tap skipping, tiles straddling 2 pixels
    (256, 16, 16, 32, 64, 5, 2),     # pixel-major, 8x8 grid, whole tiles per pixel
    (128, 6, 6, 32, 64, 4, 2),       # pixel-major with 4x4 kernels
    (3, 64, 64, 3, 64, 5, 2),        # dcgan64 d_h0: bwd-data = LDS-patch quad kernel (32x32 quads)
    (8, 128, 128, 64, 3, 7, 1),      # c7s1-3 RGB head: stride-1 small-N VALU kernel (>= 512 tiles)
    (2, 256, 256, 16, 1, 3, 1),      # same kernel, N=1, 3x3, one channel chunk
    (32, 64, 64, 32, 4, 5, 1),       # same kernel, N=4
    (2, 32, 32, 64, 3, 7, 1),        # too few tiles for the VALU head kernel; its backward-data = LDS-patch kernel, transposed stride-1 form
    (2, 16, 32, 3, 64, 7, 1),        # c7s1-64 from RGB: LDS-patch kernel at stride 1 (K = 147)
    (3, 8, 16, 4, 32, 3, 1),         # LDS-patch, stride 1, 4 input channels
    (2, 64, 32, 1, 32, 4, 2),        # 1 channel, 4x4 kernel, 32x16 quads -> global-load quad kernel (Ws % 32 != 0)
    (2, 16, 64, 2, 16, 5, 2),        # 2 channels, 8x32 quads: LDS-patch kernel with a single 16-channel chunk
    (9, 16, 16, 128, 256, 5, 2),     # N = 256 (split-bf16: 128 x 256 tiles, ragged last tile), both directions eligible
    (256, 8, 8, 128, 256, 5, 2),     # the same pixel-major: whole one-pixel tiles, zero-tap skipping, heaviest-first order
    (3, 9, 7, 96, 384, 3, 1),        # N = 384 (split-bf16: 256 x 128 tiles), 3 chunks of 32 channels, stride 1, odd sizes
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s", CONV_CASES)
@pytest.mark.parametrize("epi", ["none", "lrelu"])
def test_conv2d_fwd(B, H, W, Cin, Cout, k, s, epi, contraction):
    from cgs_amd import kernels as K, lib
    x, w, b = rnd((B, H, W, Cin), 1), rnd((k, k, Cin, Cout), 2, 0.05), rnd((Cout,), 3, 0.1)
    want = R.conv2d(x, w, b, s, s)
    if epi == "lrelu":
        want = R.lrelu(want)
    got = K.conv2d_fwd(x.to(dev()), w.to(dev()), b.to(dev()), s, s, lib.EPI_LRELU if epi == "lrelu" else lib.EPI_NONE)
    assert got.shape == want.shape
    close(got, want, 2e-5)
    if contraction == "bx6":
        assert lib.last_kernel().startswith("igemm_bx6_kernel") == (Cin % 32 == 0 and Cout % 64 == 0 and k <= 16), lib.last_kernel()


def test_stride1_rgb_layers_use_the_patch_kernel():
    """c7s1-64 (3 -> 64, stride 1) forward and the backward-data of c7s1-3 (64 <- 3): both are stride-1 correlations over a
    3-channel tensor and run on conv_patch_kernel (the latter with flipped taps / transposed weights)."""
    from cgs_amd import kernels as K, lib
    d = dev()
    x, w, b = rnd((2, 16, 32, 3), 1), rnd((7, 7, 3, 64), 2, 0.05), rnd((64,), 3, 0.1)
    got = K.conv2d_fwd(x.to(d), w.to(d), b.to(d), 1, 1)
    assert lib.last_kernel() == "conv_patch_kernel<147, 21, 67>"        # the fixed-geometry form of the 7x7x3 stride-1 reduction
    close(got, R.conv2d(x, w, b, 1, 1), 2e-5)
    xs = rnd((2, 16, 32, 64), 4).requires_grad_(True)
    w2 = rnd((7, 7, 64, 3), 5, 0.05)
    y = R.conv2d(xs, w2, torch.zeros(3), 1, 1)
    dy = rnd(tuple(y.shape), 6)
    (y * dy).sum().backward()
    got = K.conv2d_bwd_data(dy.to(d), w2.to(d), (16, 32), 1, 1)
    assert lib.last_kernel() == "conv_patch_kernel<147, 21, 67>"
    close(got, xs.grad, 2e-5)
    # the general form (geometry at run time) on a 5x5x3 stride-1 layer, and the PatchGAN stem's 4x4x3 stride-2 form, all fused epilogues
    for k_, s_, name in ((5, 1, "conv_patch_kernel"), (4, 2, "conv_patch_kernel<48, 12, 103>")):
        x, w, b = rnd((3, 16, 32, 3), 7), rnd((k_, k_, 3, 64), 8, 0.05), rnd((64,), 9, 0.1)
        for epi, fn in ((lib.EPI_NONE, lambda t: t), (lib.EPI_LRELU, R.lrelu), (lib.EPI_TANH, torch.tanh)):
            got = K.conv2d_fwd(x.to(d), w.to(d), b.to(d), s_, s_, epi)
            assert lib.last_kernel() == name, lib.last_kernel()
            close(got, fn(R.conv2d(x, w, b, s_, s_)), 2e-5)


def test_conv2d_smalln_head_tanh():
    """c7s1-3 + tanh of the CycleGAN generator tail goes to conv_smalln_f_kernel with the tanh fused."""
    from cgs_amd import kernels as K, lib
    x, w, b = rnd((8, 128, 128, 64), 1), rnd((7, 7, 64, 3), 2, 0.05), rnd((3,), 3, 0.1)
    want = torch.tanh(R.conv2d(x, w, b, 1, 1))
    got = K.conv2d_fwd(x.to(dev()), w.to(dev()), b.to(dev()), 1, 1, lib.EPI_TANH)
    assert lib.last_kernel().startswith("conv_smalln_f_kernel")
    close(got, want, 2e-5)


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s", CONV_CASES)
def test_conv2d_bwd_data(B, H, W, Cin, Cout, k, s, contraction):
    from cgs_amd import kernels as K, lib
    x = rnd((B, H, W, Cin), 1).requires_grad_(True)
    w = rnd((k, k, Cin, Cout), 2, 0.05)
    y = R.conv2d(x, w, torch.zeros(Cout), s, s)
    dy = rnd(tuple(y.shape), 4)
    (y * dy).sum().backward()
    got = K.conv2d_bwd_data(dy.to(dev()), w.to(dev()), (H, W), s, s)
    close(got, x.grad, 2e-5)
    if contraction == "bx6":
        assert lib.last_kernel().startswith("igemm_bx6_kernel") == (Cout % 32 == 0 and Cin % 64 == 0 and s <= 2), lib.last_kernel()


DECONV_CASES = [  # B,H,W,Cin,Ho,Wo,Cout,k,s
    (3, 8, 8, 256, 16, 16, 128, 5, 2),
    (2, 16, 16, 128, 32, 32, 64, 5, 2),
    (2, 16, 16, 64, 32, 32, 3, 5, 2),     # small-N VALU kernel
    (4, 14, 14, 64, 28, 28, 1, 4, 2),     # mnist g_dc4
    (4, 7, 7, 128, 14, 14, 64, 4, 2),     # mnist g_dc3
    (2, 4, 5, 32, 7, 9, 48, 5, 2),        # odd output
    (2, 4, 5, 32, 7, 9, 3, 5, 2),         # odd output, N=3 -> MFMA fallback
    (2, 6, 5, 32, 6, 5, 64, 3, 1),        # stride 1
    (37, 4, 4, 512, 8, 8, 256, 5, 2),
    (128, 4, 4, 64, 8, 8, 128, 5, 2),     # pixel-major (4x4 class grids), both directions
    (160, 8, 8, 32, 16, 16, 64, 5, 2),    # pixel-major, tiles straddling pixels
    (128, 7, 7, 32, 14, 14, 64, 4, 2),    # pixel-major 7x7 grid, 4x4 kernels
    (3, 32, 32, 64, 64, 64, 3, 5, 2),     # dcgan64 g_h4: LDS-patch quad kernel
    (2, 8, 32, 32, 16, 64, 4, 4, 2),      # N = 4, 4x4 kernel, LDS-patch
    (3, 20, 12, 32, 40, 24, 3, 5, 2),     # LDS-patch, 16x16 tiles hanging over both edges
    (2, 40, 40, 16, 80, 80, 2, 5, 2),     # LDS-patch, 16x16 tiles, 2.5 tiles per side
    (130, 4, 4, 128, 8, 8, 256, 5, 2),    # N = 256 forward / 128 backward, pixel-major 4x4 class grids straddling tiles (split-bf16 both ways)
    (3, 6, 5, 128, 11, 10, 128, 4, 2),    # odd output: parity classes of different size, 4x4 kernel, both directions eligible
]


@pytest.mark.parametrize("B,H,W,Cin,Ho,Wo,Cout,k,s", DECONV_CASES)
@pytest.mark.parametrize("epi", ["none", "affine_relu", "tanh"])
def test_deconv2d_fwd(B, H, W, Cin, Ho, Wo, Cout, k, s, epi, contraction):
    from cgs_amd import kernels as K, lib
    x, w, b = rnd((B, H, W, Cin), 1), rnd((k, k, Cout, Cin), 2, 0.05), rnd((Cout,), 3, 0.1)
    a, c = rnd((Cout,), 5).abs() + 0.5, rnd((Cout,), 6, 0.3)
    want = R.deconv2d(x, w, b, (B, Ho, Wo, Cout), s, s)
    d = dev()
    if epi == "affine_relu":
        want = torch.relu(a * want + c)
        got = K.deconv2d_fwd(x.to(d), w.to(d), b.to(d), (Ho, Wo), s, s, lib.EPI_AFFINE_RELU, a.to(d), c.to(d))
    elif epi == "tanh":
        want = torch.tanh(want)
        got = K.deconv2d_fwd(x.to(d), w.to(d), b.to(d), (Ho, Wo), s, s, lib.EPI_TANH)
    else:
        got = K.deconv2d_fwd(x.to(d), w.to(d), b.to(d), (Ho, Wo), s, s)
    close(got, want, 2e-5)
    if contraction == "bx6":
        assert lib.last_kernel().startswith("igemm_bx6_kernel") == (Cin % 32 == 0 and Cout % 64 == 0), lib.last_kernel()


@pytest.mark.parametrize("B,H,W,Cin,Ho,Wo,Cout,k,s", DECONV_CASES)
def test_deconv2d_bwd_data(B, H, W, Cin, Ho, Wo, Cout, k, s, contraction):
    from cgs_amd import kernels as K, lib
    x = rnd((B, H, W, Cin), 1).requires_grad_(True)
    w = rnd((k, k, Cout, Cin), 2, 0.05)
    y = R.deconv2d(x, w, torch.zeros(Cout), (B, Ho, Wo, Cout), s, s)
    dy = rnd(tuple(y.shape), 4)
    (y * dy).sum().backward()
    got = K.deconv2d_bwd_data(dy.to(dev()), w.to(dev()), (H, W), s, s)
    close(got, x.grad, 2e-5)
    if contraction == "bx6":
        assert lib.last_kernel().startswith("igemm_bx6_kernel") == (Cout % 32 == 0 and Cin % 64 == 0), lib.last_kernel()


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,name", [
    # (kernel template arguments: N, 16-pixel tiles per row, straight-line 5x5x64 form, row pairs per task, aux epilogue)
    (37, 32, 32, 64, 3, 5, "convt_rows_kernel<3, 2, true, 4|2, %s>"),      # dcgan64 g_h4 (tall form) / d_h0 backward-data; row pairs not a multiple of the 8 waves
    (9, 16, 16, 64, 3, 5, "convt_rows_kernel<3, 1, true, 4|2, %s>"),       # dcgan32 g_h4: the straight-line 5x5x64 form, one pixel tile, odd image count
    (5, 10, 16, 64, 3, 5, "convt_rows_kernel<3, 1, true, 4|2, %s>"),       # the tall form with a ragged last task (10 row pairs = 4 + 4 + 2)
    (3, 6, 32, 64, 3, 5, "convt_rows_kernel<3, 2, true, 2|2, %s>"),        # fewer than 8 input rows: two row pairs per task
    (300, 16, 16, 32, 3, 5, "convt_rows_kernel<3, 1, false, 2|2, %s>"),    # dcgan32; 4800 row pairs over 512 persistent blocks
    (5, 14, 14, 64, 1, 4, "convt_taps_kernel<4>"),                         # mnist g_dc4 / d_conv1 backward-data (4x4 kernel, one channel): all 16 taps as MFMA columns
    (700, 14, 14, 64, 1, 4, "convt_taps_kernel<4>"),                       # enough images for whole-image tasks; several tasks per wave
    (3, 7, 16, 32, 1, 4, "convt_taps_kernel<2>"),                          # 16-pixel rows, a ragged last row block
    (2, 5, 9, 16, 1, 4, "convt_taps_kernel<1>"),                           # one 16-channel chunk, odd sizes
    (2, 6, 12, 128, 1, 4, "convt_taps_kernel<8>"),                         # eight chunks
    (5, 14, 14, 48, 1, 4, "convt_rows_kernel<1, 1, false, 2|2, %s>"),      # a channel count the taps form does not take: rows form (14 of 16 pixels per tile)
    (3, 20, 12, 32, 3, 5, "convt_rows_kernel<3, 1, false, 2|2, %s>"),      # ragged width, non-square
    (2, 9, 24, 16, 3, 3, "convt_rows_kernel<3, 2, false, 2|2, %s>"),       # 3x3 kernel, 24 of 32 pixels
    (2, 7, 24, 16, 1, 5, "convt_rows_kernel<1, 2, false, 2|2, %s>"),       # one channel, odd number of row pairs
    (2, 6, 48, 16, 1, 5, "convt_quad_lds_kernel<1, 16, 3>"),  # wider than 32 pixels -> quad form
    (2, 8, 40, 16, 3, 5, "convt_quad_lds_kernel<3, 16, 3>"),  # wider than the rows form takes with 3 channels -> quad form
])
def test_rows_form_of_small_channel_transposed_conv(B, H, W, Cin, Cout, k, name):
    """kw * N <= 16: vertical taps folded into K, horizontal taps as GEMM columns + a <= 3-term gather (convt_rows_kernel), forward
    with bias + tanh and as a conv's backward-data with the tanh gradient folded in -- against the oracle operators."""
    from cgs_amd import kernels as K, lib
    d = dev()
    x, w, b = rnd((B, H, W, Cin), 1), rnd((k, k, Cout, Cin), 2, 0.05), rnd((Cout,), 3, 0.1)
    def kname(aux):      # "...<.., fwd pairs|bwd pairs, %s>" -> the instantiation of the forward (no aux) / backward (aux) call
        if "|" not in name:
            return name
        head, rest = name.split("|")
        pairs_f = head.rsplit(", ", 1)[1]
        return (head if not aux else head[:-len(pairs_f)] + rest.split(",")[0]) + ", " + ("true>" if aux else "false>")
    got = K.deconv2d_fwd(x.to(d), w.to(d), b.to(d), (2 * H, 2 * W), 2, 2, lib.EPI_TANH)
    assert lib.last_kernel() == kname(False), lib.last_kernel()
    close(got, torch.tanh(R.deconv2d(x, w, b, (B, 2 * H, 2 * W, Cout), 2, 2)), 2e-5)
    got = K.deconv2d_fwd(x.to(d), w.to(d), None, (2 * H, 2 * W), 2, 2)
    close(got, R.deconv2d(x, w, torch.zeros(Cout), (B, 2 * H, 2 * W, Cout), 2, 2), 2e-5)
    # conv [k,k,Cout(big side),Cin] backward-data: dy [B,H,W,Cin] -> dx [B,2H,2W,Cout], times tanh'(aux)
    wc = rnd((k, k, Cout, Cin), 4, 0.05)
    xb = rnd((B, 2 * H, 2 * W, Cout), 5).requires_grad_(True)
    dy = rnd((B, H, W, Cin), 6)
    (R.conv2d(xb, wc, torch.zeros(Cin), 2, 2) * dy).sum().backward()
    aux = rnd((B, 2 * H, 2 * W, Cout), 7, 0.5)
    got = K.conv2d_bwd_data(dy.to(d), wc.to(d), (2 * H, 2 * W), 2, 2, epilogue=lib.EPI_TANH_BWD, ep_aux=aux.to(d))
    assert lib.last_kernel() == kname(True), lib.last_kernel()
    close(got, xb.grad * (1 - aux * aux), 2e-5)


@pytest.mark.parametrize("mode", ["relu_affine", "lrelu", "tanh"])
@pytest.mark.parametrize("case", [(3, 16, 16, 64, 128, 5, 2), (3, 32, 32, 3, 64, 5, 2), (5, 28, 28, 1, 64, 4, 2), (2, 7, 9, 32, 40, 5, 2),
                                  (2, 64, 64, 3, 64, 5, 2), (5, 12, 12, 128, 128, 5, 2)])
def test_bwd_data_epilogues(case, mode, contraction):
    """The folded activation-gradient epilogues of both backward-data directions (and of the quad small-N kernel)."""
    from cgs_amd import kernels as K, lib
    from conftest import bx6_serves
    B, H, W, Cin, Cout, k, s = case
    if contraction == "bx6" and not bx6_serves(Cin, Cout):
        pytest.skip("no direction of this shape is served by the split-bf16 kernel")
    d = dev()
    w = rnd((k, k, Cin, Cout), 2, 0.05)
    Ho, Wo = -(-H // s), -(-W // s)
    dy = rnd((B, Ho, Wo, Cout), 4)
    aux = rnd((B, H, W, Cin), 5)
    a = rnd((Cin,), 6).abs() + 0.5
    x = rnd((B, H, W, Cin), 1).requires_grad_(True)
    (R.conv2d(x, w, torch.zeros(Cout), s, s) * dy).sum().backward()
    base = x.grad
    if mode == "relu_affine":
        want, args = torch.where(aux > 0, base * a, torch.zeros_like(base)), (lib.EPI_RELU_BWD_AFFINE, a.to(d), aux.to(d))
    elif mode == "lrelu":
        want, args = base * torch.where(aux > 0, 1.0, 0.2), (lib.EPI_LRELU_BWD, None, aux.to(d))
    else:
        want, args = base * (1 - aux * aux), (lib.EPI_TANH_BWD, None, aux.to(d))
    got = K.conv2d_bwd_data(dy.to(d), w.to(d), (H, W), s, s, epilogue=args[0], ep_a=args[1], ep_aux=args[2])
    close(got, want, 2e-5)
    # the F direction: deconv bwd-data of a deconv whose weights are w viewed as [k,k,Cout_d=Cin,Cin_d=Cout]
    xs = rnd((B, Ho, Wo, Cout), 7).requires_grad_(True)
    yb = R.deconv2d(xs, w, torch.zeros(Cin), (B, H, W, Cin), s, s)
    dyb = rnd(tuple(yb.shape), 8)
    (yb * dyb).sum().backward()
    auxs, a_s = rnd((B, Ho, Wo, Cout), 9), rnd((Cout,), 10).abs() + 0.5
    if mode == "relu_affine":
        want, args = torch.where(auxs > 0, xs.grad * a_s, torch.zeros_like(xs.grad)), (lib.EPI_RELU_BWD_AFFINE, a_s.to(d), auxs.to(d))
    elif mode == "lrelu":
        want, args = xs.grad * torch.where(auxs > 0, 1.0, 0.2), (lib.EPI_LRELU_BWD, None, auxs.to(d))
    else:
        want, args = xs.grad * (1 - auxs * auxs), (lib.EPI_TANH_BWD, None, auxs.to(d))
    got = K.deconv2d_bwd_data(dyb.to(d), w.to(d), (Ho, Wo), s, s, epilogue=args[0], ep_a=args[1], ep_aux=args[2])
    close(got, want, 2e-5)


@pytest.mark.parametrize("B,K_,N", [(64, 6272, 1024), (64, 62, 1024), (33, 1024, 6272), (1024, 8192, 1), (7, 100, 8192), (5, 1023, 1)])
def test_linear_fwd_bwd(B, K_, N):
    from cgs_amd import kernels as K
    x = rnd((B, K_), 1).requires_grad_(True)
    w, b = rnd((K_, N), 2, 0.02), rnd((N,), 3, 0.1)
    y = R.linear(x, w, b)
    dy = rnd((B, N), 4)
    (y * dy).sum().backward()
    d = dev()
    from cgs_amd import lib
    close(K.linear_fwd(x.detach().to(d), w.to(d), b.to(d)), y, 2e-5)
    fwd_kernel = lib.last_kernel()
    close(K.linear_bwd_data(dy.to(d), w.to(d)), x.grad, 2e-5)
    if (B, K_, N) in ((64, 6272, 1024), (33, 1024, 6272)):        # <= 64 (or > 64) rows, split over K: 64- (128-) row tiles
        assert fwd_kernel == ("igemm_kernel<64, 128, 4, true, 32, false>" if B <= 64 else "igemm_kernel<128, 128, 4, true, 32, false>"), fwd_kernel
        assert lib.last_kernel() == "igemm_kernel<64, 128, 4, true, 32, false>"


@pytest.mark.parametrize("shape", [(64, 7, 7, 128), (64, 1024), (10, 16, 16, 128), (3, 4, 4, 512), (257, 8, 8, 256),
                                   (128, 256), (129, 256), (50, 40), (1, 5, 5, 8), (16, 1024)])      # (<= 128 rows: the one-launch small-group kernel)
@pytest.mark.parametrize("leak", [0.2, 1.0])
def test_bn_train_lrelu_fwd_bwd(shape, leak):
    from cgs_amd import kernels as K
    C = shape[-1]
    x = (rnd(shape, 1) * 1.5 + 0.3).requires_grad_(True)
    g, b = rnd((C,), 2, 0.1) + 1.0, rnd((C,), 3, 0.1)
    bn = R.bn_train(x, g, b)
    y = torch.where(bn > 0, bn, leak * bn)
    dy = rnd(shape, 4)
    (y * dy).sum().backward()
    d = dev()
    gy, mean, invstd = K.bn_train_lrelu_fwd(x.detach().to(d), g.to(d), b.to(d), leak)
    red = tuple(range(len(shape) - 1))
    close(mean, x.detach().mean(red), 1e-5)
    close(invstd, 1.0 / torch.sqrt(x.detach().var(red, unbiased=False) + R.BN_EPS), 1e-5)
    close(gy, y, 1e-5)
    gdx = K.bn_train_lrelu_bwd_data(dy.to(d), x.detach().to(d), g.to(d), b.to(d), mean, invstd, leak)
    close(gdx, x.grad, 5e-5)
    # determinism: same inputs twice -> bit-identical statistics
    gy2, mean2, invstd2 = K.bn_train_lrelu_fwd(x.detach().to(d), g.to(d), b.to(d), leak)
    assert torch.equal(mean, mean2) and torch.equal(invstd, invstd2) and torch.equal(gy, gy2)


def test_elementwise_and_fold():
    from cgs_amd import kernels as K
    d = dev()
    x = rnd((5, 6, 6, 12), 1)
    dy = rnd((5, 6, 6, 12), 2)
    C = 12
    g, b, mm, mv = rnd((C,), 3).abs() + 0.5, rnd((C,), 4), rnd((C,), 5), rnd((C,), 6).abs() + 0.1
    a, c = K.bn_fold(g.to(d), b.to(d), mm.to(d), mv.to(d))
    aw = g / torch.sqrt(mv + R.BN_EPS)
    close(a, aw, 1e-6); close(c, b - aw * mm, 1e-6)
    y = K.affine_relu_fwd(x.to(d), a, c)
    close(y, torch.relu(R.bn_infer(x, g, b, mm, mv)), 1e-5)
    close(K.affine_relu_bwd(dy.to(d), y, a), dy * (y.cpu() > 0) * aw, 1e-5)
    yl = K.lrelu_fwd(x.to(d))
    close(yl, R.lrelu(x), 1e-6)
    close(K.lrelu_bwd(dy.to(d), yl), dy * torch.where(x > 0, 1.0, 0.2), 1e-6)
    yt = K.tanh_fwd(x.to(d))
    close(yt, torch.tanh(x), 1e-5)
    close(K.tanh_bwd(dy.to(d), yt), dy * (1 - torch.tanh(x) ** 2), 1e-5)
    logits = rnd((9, 4), 7, 5.0)
    dl, lm = K.bce_ones_grad_rowmean(logits.to(d))
    close(dl, torch.sigmoid(logits) - 1, 1e-5); close(lm, logits.mean(1), 1e-6)
    for shape in ((8, 32, 32, 1), (3, 1000), (5, 64), (2, 77)):      # PatchGAN maps: the wave-per-sample form (P >= 64)
        lg = rnd(shape, 8, 3.0)
        dl, lm = K.bce_ones_grad_rowmean(lg.to(d))
        close(dl, torch.sigmoid(lg) - 1, 1e-5)
        close(lm, lg.reshape(shape[0], -1).mean(1), 1e-5)


def test_refine_update_and_select_match_policy():
    from cgs_amd import kernels as K
    from oracle import sampling_ref as S
    d = dev()
    B, F = 6, 3 * 3 * 8
    th0 = rnd((B, 3, 3, 8), 1)
    pol = S.Policy(0.1, "momentum")
    th_ref = th0.clone()
    th, m = th0.to(d).clone(), torch.zeros_like(th0).to(d)
    best_t, best_l, best_s = th.clone(), torch.full((B,), -1.0, device=d), torch.ones(B, device=d)
    bl_ref, bs_ref, bt_ref = torch.full((B,), -1.0), torch.ones(B), th0.clone()
    for i in range(4):
        g = rnd((B, 3, 3, 8), 10 + i)
        th_ref = pol.step(th_ref, g)
        K.refine_update(th, m, g.to(d), 0.1, 0.9, first=(i == 0))
        assert torch.equal(th.cpu(), th_ref), "momentum update must be bit-exact (same op order, policy.py:33-36)"
        logit = rnd((B,), 20 + i)
        upd = logit > bl_ref
        bl_ref = torch.where(upd, logit, bl_ref); bs_ref = torch.where(upd, torch.full_like(bs_ref, i + 1), bs_ref)
        bt_ref = torch.where(upd.view(-1, 1, 1, 1), th_ref, bt_ref)
        K.refine_select(th, logit.to(d), None, i, best_t, best_l, best_s)
        assert torch.equal(best_l.cpu(), bl_ref) and torch.equal(best_s.cpu(), bs_ref) and torch.equal(best_t.cpu(), bt_ref)
    # probabilistic mode: forced indices
    forced = torch.tensor([0, 1, 2, 3, 4, 1], dtype=torch.int32, device=d)
    bt2, bl2, bs2 = th.clone().zero_(), torch.zeros(B, device=d), torch.ones(B, device=d)
    K.refine_select(th, best_l, forced, 1, bt2, bl2, bs2)
    sel = (forced == 1).cpu()
    assert torch.equal(bs2.cpu(), torch.where(sel, 2.0, 1.0)) and torch.equal(bt2.cpu()[sel], th.cpu()[sel]) and bt2.cpu()[~sel].abs().sum() == 0
    # clip (collaborator.py:69-70)
    th3, m3 = th0.to(d).clone(), torch.zeros_like(th0).to(d)
    g = rnd((B, 3, 3, 8), 99)
    K.refine_update(th3, m3, g.to(d), 0.1, 0.9, True, 0.05, 1.5)
    assert torch.equal(th3.cpu(), torch.clamp(th0 - 0.1 * g, 0.05, 1.5))


def test_errors_are_loud():
    from cgs_amd import kernels as K, lib
    d = dev()
    with pytest.raises(lib.CgsError):
        K.conv2d_fwd(torch.zeros(1, 4, 4, 8), torch.zeros(5, 5, 8, 8, device=d), None)      # CPU tensor
    with pytest.raises(lib.CgsError):
        K.conv2d_fwd(torch.zeros(1, 4, 4, 8, device=d), torch.zeros(5, 5, 4, 8, device=d), None)   # Cin mismatch
    with pytest.raises(lib.CgsError):
        K.deconv2d_fwd(torch.zeros(1, 4, 4, 8, device=d), torch.zeros(5, 5, 8, 8, device=d), None, (9, 9))  # bad SAME geometry


def test_edge_shapes_and_argument_errors():
    """Ragged / degenerate cases through the C ABI: batch 1, 1x1 spatial, kernel smaller than stride (parity classes
    with no tap get bias only), misaligned pointers, too-small workspace."""
    from cgs_amd import kernels as K, lib
    d = dev()
    # B = 1, 1x1 input
    x, w, b = rnd((1, 1, 1, 32), 1), rnd((5, 5, 32, 64), 2, 0.05), rnd((64,), 3)
    close(K.conv2d_fwd(x.to(d), w.to(d), b.to(d)), R.conv2d(x, w, b, 2, 2), 2e-5)
    # 1x1 kernel with stride 2 in the transposed direction: three of the four parity classes have no tap -> bias only
    xs, wt, bt = rnd((2, 4, 4, 32), 4), rnd((1, 1, 64, 32), 5, 0.1), rnd((64,), 6)
    close(K.deconv2d_fwd(xs.to(d), wt.to(d), bt.to(d), (8, 8)), R.deconv2d(xs, wt, bt, (2, 8, 8, 64), 2, 2), 2e-5)
    # single output channel linear with odd K, batch 1
    xl, wl, bl = rnd((1, 77), 7), rnd((77, 1), 8), rnd((1,), 9)
    close(K.linear_fwd(xl.to(d), wl.to(d), bl.to(d)), R.linear(xl, wl, bl), 2e-5)
    # misaligned activation pointer is rejected, not mis-read
    buf = torch.zeros(1 * 8 * 8 * 32 + 1, device=d)
    with pytest.raises(lib.CgsError):
        K.conv2d_fwd(buf[1:].view(1, 8, 8, 32), w.to(d), b.to(d))
    # workspace too small is an error code, not a crash
    xd, wd = torch.zeros(1, 8, 8, 32, device=d), w.to(d)
    yd = torch.empty(1, 4, 4, 64, device=d)
    small = torch.empty(16, device=d)
    rc = lib.load().cgs_conv2d_nhwc_fwd(xd.data_ptr(), wd.data_ptr(), None, yd.data_ptr(), 1, 8, 8, 32, 64, 5, 5, 2, 2, 0, None, None,
                                       small.data_ptr(), small.numel() * 4, 0, None)
    assert rc == lib.EWORKSPACE and b"workspace" in lib.load().cgs_last_error()
    rc = lib.load().cgs_conv2d_nhwc_fwd(xd.data_ptr(), wd.data_ptr(), None, yd.data_ptr(), 0, 8, 8, 32, 64, 5, 5, 2, 2, 0, None, None,
                                       small.data_ptr(), small.numel() * 4, 0, None)
    assert rc == lib.EINVAL


@pytest.mark.parametrize("case", ["conv_pixmajor", "conv_parity", "conv_bwd_T", "deconv_fwd_T_tanh", "deconv_bwd_F",
                                  "dominant_F", "dominant_T", "dominant_T_bwd"])
def test_many_block_grids_use_the_16_deep_variant(case):
    """Grids with >= 512 blocks per parity class run the 16-deep K-tile instantiation (four blocks per CU, half-staged
    epilogue; short-K launches to 64 channels its tall 256 x 64 form); every direction / tap order of it against the oracle, and that
    it really is the kernel that ran."""
    from cgs_amd import kernels as K, lib
    d = dev()
    if case in ("conv_pixmajor", "conv_parity"):
        B, H, Cin, Cout = (1024, 16, 32, 64) if case == "conv_pixmajor" else (256, 32, 32, 64)      # 8x8 grid: pixel-major; 16x16: parity taps
        x, w, b = rnd((B, H, H, Cin), 1), rnd((5, 5, Cin, Cout), 2, 0.05), rnd((Cout,), 3, 0.1)
        got = K.conv2d_fwd(x.to(d), w.to(d), b.to(d), 2, 2, lib.EPI_LRELU)
        name = lib.last_kernel()
        close(got, R.lrelu(R.conv2d(x, w, b, 2, 2)), 2e-5)
        assert name == f"igemm_kernel<128, 64, 4, true, 16, {'true' if case == 'conv_parity' else 'false'}>", name
    elif case == "dominant_F":
        # the headline's dominant instantiation (33 % of the dcgan64 step: d_h2 fwd, g_h2 bwd), forward direction:
        # 16x16 -> 8x8 pixel-major grid, N = 256 (two 128-wide n-tiles), 1024 blocks
        B, H, Cin, Cout = 1024, 16, 32, 256
        x, w, b = rnd((B, H, H, Cin), 1), rnd((5, 5, Cin, Cout), 2, 0.05), rnd((Cout,), 3, 0.1)
        got = K.conv2d_fwd(x.to(d), w.to(d), b.to(d), 2, 2, lib.EPI_LRELU)
        name = lib.last_kernel()
        close(got, R.lrelu(R.conv2d(x, w, b, 2, 2)), 2e-5)
        assert name == "igemm_kernel<128, 128, 4, true, 16, false>", name
    elif case == "dominant_T":
        # ... and transposed direction (g_h2 fwd): 8x8 -> 16x16, four parity classes of 512 blocks each, N = 128
        B, H, Cin, Cout = 1024, 8, 32, 128
        x, w, b = rnd((B, H, H, Cin), 1), rnd((5, 5, Cout, Cin), 2, 0.05), rnd((Cout,), 3, 0.1)
        got = K.deconv2d_fwd(x.to(d), w.to(d), b.to(d), (2 * H, 2 * H), 2, 2, lib.EPI_NONE)
        name = lib.last_kernel()
        close(got, R.deconv2d(x, w, b, (B, 2 * H, 2 * H, Cout), 2, 2), 2e-5)
        assert name == "igemm_kernel<128, 128, 4, true, 16, false>", name
    elif case == "dominant_T_bwd":
        # ... and as a conv's backward-data with the lrelu gradient of the layer below folded in (d_h2 bwd)
        B, H, Cin, Cout = 1024, 16, 128, 32
        w = rnd((5, 5, Cin, Cout), 2, 0.05)
        x = rnd((B, H, H, Cin), 1).requires_grad_(True)
        y = R.conv2d(x, w, torch.zeros(Cout), 2, 2)
        dy = rnd(tuple(y.shape), 4)
        (y * dy).sum().backward()
        below = rnd((B, H, H, Cin), 6)                       # saved lrelu output of the layer below
        got = K.conv2d_bwd_data(dy.to(d), w.to(d), (H, H), 2, 2, epilogue=lib.EPI_LRELU_BWD, ep_aux=below.to(d))
        name = lib.last_kernel()
        close(got, x.grad * torch.where(below > 0, 1.0, 0.2), 2e-5)
        assert name == "igemm_kernel<128, 128, 4, true, 16, false>", name
    elif case == "conv_bwd_T":
        B, H, Cin, Cout = 1024, 32, 32, 64
        w = rnd((5, 5, Cin, Cout), 2, 0.05)
        x = rnd((B, H, H, Cin), 1).requires_grad_(True)
        y = R.conv2d(x, w, torch.zeros(Cout), 2, 2)
        dy = rnd(tuple(y.shape), 4)
        (y * dy).sum().backward()
        got = K.conv2d_bwd_data(dy.to(d), w.to(d), (H, H), 2, 2)
        name = lib.last_kernel()
        close(got, x.grad, 2e-5)
        assert name == "igemm_kernel<128, 64, 4, true, 16, false>", name
    elif case == "deconv_fwd_T_tanh":
        B, H, Cin, Cout = 1024, 16, 32, 64
        x, w, b = rnd((B, H, H, Cin), 1), rnd((5, 5, Cout, Cin), 2, 0.05), rnd((Cout,), 3, 0.1)
        got = K.deconv2d_fwd(x.to(d), w.to(d), b.to(d), (2 * H, 2 * H), 2, 2, lib.EPI_TANH)
        name = lib.last_kernel()
        close(got, torch.tanh(R.deconv2d(x, w, b, (B, 2 * H, 2 * H, Cout), 2, 2)), 2e-5)
        # (64 output channels, K <= 512 per parity class, 4096 tall tiles: the 256 x 64 block of round 5 -- four 64 x 64 wave tiles stacked)
        assert name == "igemm_kernel<256, 64, 4, true, 16, false>", name
    else:
        B, H, Cin, Cout = 256, 16, 128, 32                                                         # F direction over dy [B,32,32,32] -> dx [B,16,16,128]
        x = rnd((B, H, H, Cin), 1).requires_grad_(True)
        w = rnd((5, 5, Cout, Cin), 2, 0.05)
        y = R.deconv2d(x, w, torch.zeros(Cout), (B, 2 * H, 2 * H, Cout), 2, 2)
        dy = rnd(tuple(y.shape), 4)
        (y * dy).sum().backward()
        got = K.deconv2d_bwd_data(dy.to(d), w.to(d), (H, H), 2, 2)
        name = lib.last_kernel()
        close(got, x.grad, 2e-5)
        assert name == "igemm_kernel<128, 128, 4, true, 16, true>", name


@pytest.mark.parametrize("B,H,Cin,Cout,k", [(64, 16, 32, 128, 5), (1024, 8, 64, 64, 5), (5, 6, 32, 64, 3), (130, 8, 32, 68, 4), (512, 8, 64, 256, 5),
                                            (3, 10, 32, 384, 3),
                                            # the reference's batch size on D's last layers: 16-64 tiles, split over K with the statistics
                                            # left by the reduce pass (splitk_reduce_stats_kernel)
                                            (64, 4, 256, 512, 5), (64, 8, 128, 256, 5), (64, 8, 256, 512, 5), (33, 14, 64, 128, 5)])
def test_conv_with_fused_bn_statistics(B, H, Cin, Cout, k, contraction):
    """cgs_conv2d_nhwc_fwd_stats: the conv's output is unchanged, its per-block partial sums reduce to the per-channel sum and sum
    of squares of that output, and cgs_bn_train_lrelu_fwd_from_partials gives the batch norm of the plain two-kernel path -- also on
    pixel-major / balanced tile orders (B = 1024) and with ragged last tiles / channel counts off the 64 grid."""
    from cgs_amd import kernels as K, lib
    d = dev()
    x, w, b = rnd((B, H, H, Cin), 1).to(d), rnd((k, k, Cin, Cout), 2, 0.05).to(d), rnd((Cout,), 3, 0.1).to(d)
    G = K.conv_stat_partials(tuple(x.shape), tuple(w.shape), 2, 2)
    assert G == 2 * ((B * (H // 2) ** 2 + 127) // 128)
    part = torch.full((G, 2, Cout), float("nan"), device=d)
    y = K.conv2d_fwd_stats(x, w, b, part, 2, 2)
    assert lib.last_kernel().startswith("igemm_bx6_kernel" if contraction == "bx6" else "igemm_kernel")
    y_plain = K.conv2d_fwd(x, w, b, 2, 2)
    close(y, y_plain, 2e-6)          # (not bit-equal in general: small grids take the split-K path without the statistics)
    close(y, R.conv2d(x.cpu(), w.cpu(), b.cpu(), 2, 2), 2e-5)
    sums = part.double().sum(0)                                   # partial rows of tiles past M are never written: must not exist
    assert torch.isfinite(sums).all()
    flat = y.double().reshape(-1, Cout)
    close(sums[0].float(), flat.sum(0).float(), 2e-5)
    close(sums[1].float(), (flat * flat).sum(0).float(), 2e-5)
    gamma, beta = (rnd((Cout,), 4, 0.2) + 1).to(d), rnd((Cout,), 5, 0.1).to(d)
    got = K.bn_train_lrelu_fwd_from_partials(y, part, gamma, beta, 0.2)
    want = K.bn_train_lrelu_fwd(y, gamma, beta, 0.2)
    for a, c in zip(got, want):
        close(a, c, 2e-6)
    # determinism
    part2 = torch.empty_like(part)
    K.conv2d_fwd_stats(x, w, b, part2, 2, 2)
    assert torch.equal(part, part2)


@pytest.mark.parametrize("case", [
    # (op, B, H, Cin, Cout, k, stride, group_images)
    ("conv", 8, 32, 32, 64, 4, 2, 1),          # image-major, one group per sample (PatchGAN conv + instance norm)
    ("conv", 8, 32, 32, 64, 4, 1, 1),          # stride 1
    ("conv", 6, 16, 32, 36, 3, 2, 2),          # groups of two samples, channel count off the 64 grid, ragged last tile
    ("deconv", 8, 16, 64, 32, 3, 2, 1),        # transposed: four parity classes, a group's rows in four segments
    ("deconv", 4, 32, 32, 64, 5, 2, 1),
    ("conv", 256, 16, 32, 64, 5, 2, 64),       # pixel-major whole tiles: four logical batches of 64 (fused batches of D's batch norm)
    ("conv", 512, 8, 64, 128, 4, 2, 128),
    ("deconv", 256, 4, 64, 32, 4, 2, 64),      # pixel-major transposed
    ("conv", 8, 16, 64, 128, 3, 1, 1),         # 128 / 256 output channels: the shapes the split-bf16 kernel serves too
    ("deconv", 4, 16, 64, 256, 3, 2, 1),
    ("conv", 256, 8, 32, 128, 5, 2, 128),      # ... pixel-major whole tiles
    ("deconv", 256, 4, 64, 128, 4, 2, 64),
], ids=lambda c: "-".join(str(v) for v in c))
def test_fused_statistics_per_group_of_images(case, contraction):
    """cgs_conv_stat_layout + cgs_conv2d_nhwc_fwd_stats / cgs_deconv2d_nhwc_fwd_stats + cgs_groupnorm_lrelu_fwd_from_partials: the
    statistics of every group of consecutive images come out of the producing convolution's epilogue (either direction, image- and
    pixel-major row orders) and give the instance norm / per-logical-batch batch norm of the separate-pass kernels."""
    from cgs_amd import kernels as K, lib
    op, B, H, Cin, Cout, k, s_, grp = case
    bx6 = contraction == "bx6"
    if bx6 and not (Cin % 32 == 0 and Cout % 64 == 0):
        pytest.skip("the forward direction of this shape is not served by the split-bf16 kernel")
    d = dev()
    x = rnd((B, H, H, Cin), 1).to(d)
    if op == "conv":
        w, b = rnd((k, k, Cin, Cout), 2, 0.05).to(d), rnd((Cout,), 3, 0.1).to(d)
        Ho = -(-H // s_)
        lay = K.conv_stat_layout(lib.CONV_FWD, B, H, H, Cin, 0, 0, Cout, k, k, s_, s_, grp)
        assert lay is not None
        part = torch.full((lay[0], 2, Cout), float("nan"), device=d)
        y = K.conv2d_fwd_stats(x, w, b, part, s_, s_)
        y_plain = K.conv2d_fwd(x, w, b, s_, s_)
    else:
        w, b = rnd((k, k, Cout, Cin), 2, 0.05).to(d), rnd((Cout,), 3, 0.1).to(d)
        Ho = H * s_
        lay = K.conv_stat_layout(lib.DECONV_FWD, B, H, H, Cin, Ho, Ho, Cout, k, k, s_, s_, grp)
        assert lay is not None
        part = torch.full((lay[0], 2, Cout), float("nan"), device=d)
        y = K.deconv2d_fwd(x, w, b, (Ho, Ho), s_, s_, part=part)
        y_plain = K.deconv2d_fwd(x, w, b, (Ho, Ho), s_, s_)
    assert lib.last_kernel().startswith("igemm_bx6_kernel" if bx6 else "igemm_kernel")
    close(y, y_plain, 2e-6)
    rows, rps, nseg, stride = lay
    groups = B // grp
    # the rows the layout assigns to group g sum to that group's column sums
    yg = y.double().reshape(groups, -1, Cout)
    for g in range(groups):
        idx = torch.tensor([sg * stride + g * rps + i for sg in range(nseg) for i in range(rps)], device=d)
        sums = part[idx].double().sum(0)
        assert torch.isfinite(sums).all()
        close(sums[0].float(), yg[g].sum(0).float(), 3e-5)
        close(sums[1].float(), (yg[g] * yg[g]).sum(0).float(), 3e-5)
    scale, offset = (rnd((Cout,), 4, 0.2) + 1).to(d), rnd((Cout,), 5, 0.1).to(d)
    got = K.groupnorm_lrelu_fwd_from_partials(y, part, lay, groups, scale, offset, 0.2)
    want = K.instnorm_lrelu_fwd(y.view(groups, -1, Cout), scale, offset, 0.2)
    close(got[0].reshape(-1), want[0].reshape(-1), 1e-5)
    close(got[1], want[1], 1e-5)
    close(got[2], want[2], 1e-5)
    part2 = torch.empty_like(part)                                 # determinism
    if op == "conv":
        K.conv2d_fwd_stats(x, w, b, part2, s_, s_)
    else:
        K.deconv2d_fwd(x, w, b, (Ho, Ho), s_, s_, part=part2)
    assert torch.equal(part, part2)


@pytest.mark.parametrize("case", [
    # (op of the layer ABOVE the norm, B, H of the gradient it produces, C of that gradient, channels above, k, stride, group_images, leak)
    ("conv", 10, 16, 64, 128, 5, 2, 10, 0.2),        # batch norm over the whole batch (D's bn1 <- conv2's backward-data), image-major, ragged last tile
    ("conv", 256, 8, 128, 256, 5, 2, 256, 0.2),      # pixel-major whole tiles, one group
    ("conv", 256, 16, 64, 128, 5, 2, 64, 0.2),       # pixel-major, four logical batches of 64 (fused batches)
    ("conv", 8, 32, 64, 128, 4, 2, 1, 0.2),          # instance norm: a group per sample (PatchGAN), image-major, four parity classes
    ("conv", 8, 16, 64, 64, 3, 1, 1, 0.0),           # stride 1 (the 3x3 convolutions of a residual block), relu
    ("conv", 6, 16, 32, 36, 3, 2, 2, 1.0),           # groups of two samples, plain norm (leak 1), channel count off the 64 grid above
    ("deconv", 8, 16, 64, 32, 3, 2, 1, 0.0),         # the layer above is a transposed conv (up-sampling): its backward-data is a strided conv
    ("deconv", 256, 8, 64, 32, 4, 2, 64, 0.2),       # ... pixel-major
    ("deconv", 4, 8, 128, 64, 5, 2, 4, 0.2),
    ("deconv", 3, 8, 128, 3, 3, 2, 1, 0.0),          # M = 3 x 64 rows: the last tile's second wave lies past M (its parameter loads must stay inside); generic-K kernel (3 channels above)
    ("conv", 3, 8, 64, 96, 3, 1, 1, 0.2),            # the same raggedness in the 32-channel-chunk kernels
    ("conv", 64, 8, 256, 512, 5, 2, 64, 0.2),        # batch 64: an under-filled grid, split over K -- the reduce pass leaves the sums
    ("conv", 64, 4, 256, 512, 5, 2, 64, 0.2),
], ids=lambda c: "-".join(str(v) for v in c))
def test_backward_data_leaves_the_norm_backward_sums(case):
    """Round 6 (VERDICT r5 #5a): the backward-data launch of the layer ABOVE a norm produces the gradient at the norm's output, so it
    leaves the norm backward's two column sums -- sum d, sum d * xhat with d = dy * lrelu'(gamma * xhat + beta) -- per group of images
    (cgs_conv_stat_layout with the *_BWD_DATA op, cgs_*_bwd_data_nstats); cgs_norm_lrelu_bwd_from_partials then needs one pass over
    dy and x instead of two.  Held to: the gradient itself is the plain backward-data's (2e-6: another tile plan may sum in another order); the rows the layout assigns to a
    group sum to that group's float64 sums; the norm's input gradient equals the separate-pass kernels' (which the oracle pins,
    test_bn_train_lrelu_fwd_bwd) to 1e-5 and the float64 formula to 2e-5; run-to-run bit-identical."""
    from cgs_amd import kernels as K, lib
    op, B, H, C, Cabove, k, s_, grp, leak = case
    d = dev()
    groups = B // grp
    x = rnd((B, H, H, C), 1).to(d)                                   # the norm's input (= output of the layer below it)
    gamma, beta = (rnd((C,), 4, 0.2) + 1).to(d), rnd((C,), 5, 0.1).to(d)
    _, mean, invstd = K.instnorm_lrelu_fwd(x.view(groups, -1, C), gamma, beta, leak)         # the statistics the forward pass saved
    if op == "conv":       # above: conv [B,H,H,C] -> [B,Ho,Ho,Cabove]; its backward-data maps g [B,Ho,Ho,Cabove] -> dy [B,H,H,C]
        Ho = -(-H // s_)
        w = rnd((k, k, C, Cabove), 2, 0.05).to(d)
        g = rnd((B, Ho, Ho, Cabove), 3).to(d)
        lay = K.conv_stat_layout(lib.CONV_BWD_DATA, B, H, H, C, 0, 0, Cabove, k, k, s_, s_, grp)
        plain = lambda: K.conv2d_bwd_data(g, w, (H, H), s_, s_)
        fused = lambda ns: K.conv2d_bwd_data(g, w, (H, H), s_, s_, nstat=ns)
    else:                  # above: deconv [B,H,H,C] -> [B,H*s,H*s,Cabove]; its backward-data is a strided conv of g
        Ho = H * s_
        w = rnd((k, k, Cabove, C), 2, 0.05).to(d)
        g = rnd((B, Ho, Ho, Cabove), 3).to(d)
        lay = K.conv_stat_layout(lib.DECONV_BWD_DATA, B, H, H, C, Ho, Ho, Cabove, k, k, s_, s_, grp)
        plain = lambda: K.deconv2d_bwd_data(g, w, (H, H), s_, s_)
        fused = lambda ns: K.deconv2d_bwd_data(g, w, (H, H), s_, s_, nstat=ns)
    assert lay is not None
    part = torch.full((lay[0], 2, C), float("nan"), device=d)
    ns = K.NormBwdStats(x, mean, invstd, gamma, beta, leak, grp, part, lay)
    dy = fused(ns)
    assert lib.last_kernel().startswith("igemm_ns_kernel")            # the twin with the sums epilogue; every other launch keeps igemm_kernel
    dy_plain = plain()
    close(dy, dy_plain, 2e-6)                                        # the gradient itself (a launch that leaves statistics may take another tile / split plan: same products, another summation order)
    # the partial rows of every group against float64 sums of the definition
    rows, rps, nseg, stride = lay
    xg, dyg = x.double().view(groups, -1, C), dy.double().view(groups, -1, C)
    mu, inv = mean.double().view(groups, 1, C), invstd.double().view(groups, 1, C)
    xhat = (xg - mu) * inv
    dmask = dyg * torch.where(gamma.double() * xhat + beta.double() > 0, 1.0, float(leak))
    for gi in range(groups):
        idx = torch.tensor([sg * stride + gi * rps + i for sg in range(nseg) for i in range(rps)], device=d)
        sums = part[idx].double().sum(0)
        assert torch.isfinite(sums).all()
        scale = dmask[gi].abs().sum(0).max().item() + 1e-30         # (sums of signed terms: the bar is relative to the sum of magnitudes)
        assert (sums[0] - dmask[gi].sum(0)).abs().max().item() <= 2e-6 * scale
        assert (sums[1] - (dmask[gi] * xhat[gi]).sum(0)).abs().max().item() <= 2e-6 * (dmask[gi] * xhat[gi]).abs().sum(0).max().item() + 1e-30
    # the norm's input gradient: from the partials == the separate-pass kernels == the float64 formula
    got = K.norm_lrelu_bwd_from_partials(dy.clone(), x, ns, groups)
    want = K.instnorm_lrelu_bwd_data(dy.view(groups, -1, C).clone(), x.view(groups, -1, C), gamma, beta, mean, invstd, leak)
    close(got.reshape(-1), want.reshape(-1), 1e-5)
    m1, m2 = dmask.mean(1, keepdim=True), (dmask * xhat).mean(1, keepdim=True)
    ref = gamma.double() * inv * (dmask - m1 - xhat * m2)
    close(got.reshape(-1), ref.reshape(-1).float(), 2e-5)
    inplace = dy.clone()                                             # the engine runs it in place
    assert K.norm_lrelu_bwd_from_partials(inplace, x, ns, groups, out=inplace) is inplace and torch.equal(inplace, got)
    part2 = torch.full_like(part, float("nan"))                      # determinism
    ns2 = K.NormBwdStats(x, mean, invstd, gamma, beta, leak, grp, part2, lay)
    assert torch.equal(fused(ns2), dy) and torch.equal(part2, part)


def test_norm_backward_sums_are_refused_where_unsupported():
    from cgs_amd import kernels as K, lib
    assert K.conv_stat_layout(lib.CONV_BWD_DATA, 8, 6, 6, 32, 0, 0, 64, 3, 3, 2, 2, 1) is None      # 9 pixels per sample and class: not whole 64-row pieces
    assert K.conv_stat_layout(lib.CONV_BWD_DATA, 8, 64, 64, 3, 0, 0, 64, 5, 5, 2, 2, 8) is None     # a 3-channel gradient: another kernel family
    assert K.conv_stat_layout(lib.CONV_BWD_DATA, 8, 16, 16, 6, 0, 0, 64, 3, 3, 1, 1, 8) is None     # C % 4 != 0
    assert K.conv_stat_layout(lib.DECONV_BWD_DATA, 256, 8, 8, 64, 16, 16, 32, 4, 4, 2, 2, 32) is None   # pixel-major: a group of 32 shares a row
    K.set_contraction("bx6_all")
    try:       # the split-bf16 kernel does not leave them: the layout says so and the engine keeps the separate pass
        assert K.conv_stat_layout(lib.CONV_BWD_DATA, 8, 16, 16, 64, 0, 0, 128, 3, 3, 1, 1, 1) is None
    finally:
        K.set_contraction("f32")
    d = dev()
    x = rnd((8, 6, 6, 32), 1).to(d)
    ns = K.NormBwdStats(x, x[:, 0, 0], x[:, 0, 0], x[0, 0, 0], x[0, 0, 0], 0.2, 1, torch.empty((64, 2, 32), device=d), (64, 1, 1, 0))
    with pytest.raises(lib.CgsError, match="not available"):
        K.conv2d_bwd_data(rnd((8, 3, 3, 64), 2).to(d), rnd((3, 3, 32, 64), 3).to(d), (6, 6), 2, 2, nstat=ns)


def test_group_statistics_layout_is_refused_where_a_group_does_not_own_whole_rows():
    from cgs_amd import kernels as K, lib
    assert K.conv_stat_layout(lib.CONV_FWD, 8, 6, 6, 32, 0, 0, 64, 3, 3, 2, 2, 1) is None          # 9 pixels per sample: not a multiple of 64 rows
    assert K.conv_stat_layout(lib.CONV_FWD, 256, 16, 16, 32, 0, 0, 64, 5, 5, 2, 2, 32) is None      # pixel-major: a logical batch of 32 shares a row with its neighbour
    assert K.conv_stat_layout(lib.DECONV_FWD, 8, 5, 5, 32, 9, 9, 64, 3, 3, 2, 2, 8) is None         # odd output: parity classes of different size
    assert K.conv_stat_layout(lib.CONV_FWD, 8, 32, 32, 3, 0, 0, 64, 5, 5, 2, 2, 1) is None          # another kernel family (3 channels)


def test_fused_statistics_not_offered_where_unsupported():
    from cgs_amd import kernels as K
    assert K.conv_stat_partials((8, 64, 64, 3), (5, 5, 3, 64), 2, 2) == 0       # the 3-channel patch kernel serves this conv
    assert K.conv_stat_partials((8, 16, 16, 32), (5, 5, 32, 6), 2, 2) == 0      # Cout % 4 != 0


def unpack_signs(words, P, C):
    """include/cgs_hip.h "sign masks": word[(c / 32) * P + p], bit 8 * (c % 4) + (c % 32) / 4  ->  bool [P, C]."""
    w = words.cpu().numpy().view(np.uint32).reshape(C // 32, P)
    c = np.arange(C)
    return ((w[c // 32, :].T >> (8 * (c % 4) + (c % 32) // 4)) & 1).astype(bool)


def pack_signs(pos):
    """bool [P, C] -> the mask words (int32 [C/32 * P]) in the documented layout."""
    P, C = pos.shape
    c = np.arange(C)
    words = np.zeros((C // 32, P), dtype=np.uint32)
    np.bitwise_or.at(words, ((c // 32)[None, :].repeat(P, 0), np.arange(P)[:, None].repeat(C, 1)),
                     pos.astype(np.uint32) << (8 * (c % 4) + (c % 32) // 4).astype(np.uint32)[None, :])
    return words.view(np.int32).reshape(-1)


@pytest.mark.parametrize("epi", ["affine_relu", "lrelu"])
@pytest.mark.parametrize("B,H,Cin,Cout", [(128, 8, 128, 64), (130, 8, 64, 96), (128, 8, 64, 128)])
def test_forward_epilogue_leaves_the_sign_mask(B, H, Cin, Cout, epi, contraction):
    """cgs_deconv2d_nhwc_fwd_signs: the same output, bit for bit, as the plain call, plus the documented bitmask of (y > 0):
    BN = 64 and BN = 128 tiles (8 / 16 lanes per row), a channel count that is not a multiple of 64, a ragged batch."""
    from cgs_amd import kernels as K, lib
    d = dev()
    x, w, b = rnd((B, H, H, Cin), 1).to(d), rnd((5, 5, Cout, Cin), 2, 0.05).to(d), rnd((Cout,), 3, 0.1).to(d)
    a, c = (rnd((Cout,), 4).abs() + 0.5).to(d), rnd((Cout,), 5, 0.1).to(d)
    e = lib.EPI_AFFINE_RELU if epi == "affine_relu" else lib.EPI_LRELU
    ea, ec = (a, c) if epi == "affine_relu" else (None, None)
    assert K.conv_signs_ok(lib.DECONV_FWD, B, H, H, Cin, 2 * H, 2 * H, Cout, 5, 5, 2, 2, e)
    y0 = K.deconv2d_fwd(x, w, b, (2 * H, 2 * H), 2, 2, e, ea, ec)
    signs = torch.full((y0.numel() // 32,), -1, dtype=torch.int32, device=d)
    y1 = K.deconv2d_fwd(x, w, b, (2 * H, 2 * H), 2, 2, e, ea, ec, signs=signs)
    assert torch.equal(y0, y1)
    got = unpack_signs(signs, B * 4 * H * H, Cout)
    assert np.array_equal(got, (y1 > 0).cpu().numpy().reshape(-1, Cout))
    bx6 = contraction == "bx6" and Cin % 32 == 0 and Cout % 64 == 0          # this forward runs on the split-bf16 kernel (same mask, same layout)
    assert lib.last_kernel().startswith("igemm_bx6_kernel" if bx6 else "igemm_kernel")
    # not offered: split-K grids (tiny batch; the split-bf16 kernel never splits over K and does offer it), channel counts off the 32
    # granule, epilogues without a kink
    assert bool(K.conv_signs_ok(lib.DECONV_FWD, 2, H, H, Cin, 2 * H, 2 * H, Cout, 5, 5, 2, 2, e)) == bx6
    assert not K.conv_signs_ok(lib.DECONV_FWD, B, H, H, Cin, 2 * H, 2 * H, 40, 5, 5, 2, 2, e)
    assert not K.conv_signs_ok(lib.DECONV_FWD, B, H, H, Cin, 2 * H, 2 * H, Cout, 5, 5, 2, 2, lib.EPI_TANH)
    if bx6:
        small = torch.full((2 * 4 * H * H * Cout // 32,), -1, dtype=torch.int32, device=d)
        y2 = K.deconv2d_fwd(x[:2].contiguous(), w, b, (2 * H, 2 * H), 2, 2, e, ea, ec, signs=small)
        assert np.array_equal(unpack_signs(small, 2 * 4 * H * H, Cout), (y2 > 0).cpu().numpy().reshape(-1, Cout))
    else:
        with pytest.raises(lib.CgsError):
            K.deconv2d_fwd(x[:2], w, b, (2 * H, 2 * H), 2, 2, e, ea, ec, signs=signs)


@pytest.mark.parametrize("mode", ["relu_affine", "lrelu"])
@pytest.mark.parametrize("B,H,C", [(3, 16, 64), (5, 8, 32), (2, 32, 96)])
def test_backward_epilogue_reads_the_sign_mask(B, H, C, mode):
    """cgs_deconv2d_nhwc_bwd_data_signs (the 5x5x3 stride-2 LDS-patch kernel): relu' / lrelu' from the 1-bit mask gives exactly
    what the fp32 aux tensor gives -- masks built here in the documented layout, aux values include exact zeros and negatives."""
    from cgs_amd import kernels as K, lib
    d = dev()
    W = 16 if H < 32 else 32
    dy, w = rnd((B, 2 * H, 2 * W, 3), 1).to(d), rnd((5, 5, 3, C), 2, 0.05).to(d)
    aux = rnd((B, H, W, C), 3)
    aux[aux.abs() < 0.3] = 0.0                                   # relu outputs are exactly 0 on a third of the elements
    a = (rnd((C,), 4).abs() + 0.5).to(d)
    e = lib.EPI_RELU_BWD_AFFINE if mode == "relu_affine" else lib.EPI_LRELU_BWD
    assert K.conv_signs_ok(lib.DECONV_BWD_DATA, B, H, W, C, 2 * H, 2 * W, 3, 5, 5, 2, 2, e)
    want = K.deconv2d_bwd_data(dy, w, (H, W), 2, 2, epilogue=e, ep_a=a if mode == "relu_affine" else None, ep_aux=aux.to(d))
    assert lib.last_kernel() == "conv_patch2_kernel<5, 16, 15, 8, 1>"
    pos = (aux > 0).numpy().reshape(-1, C)
    assert np.array_equal(unpack_signs(torch.from_numpy(pack_signs(pos)), pos.shape[0], C), pos)
    signs = torch.from_numpy(pack_signs(pos)).to(d)
    got = K.deconv2d_bwd_data(dy, w, (H, W), 2, 2, epilogue=e, ep_a=a if mode == "relu_affine" else None, ep_signs=signs)
    assert lib.last_kernel() == "conv_patch2_kernel<5, 16, 15, 8, 2>"
    assert torch.equal(got, want)
    assert not K.conv_signs_ok(lib.DECONV_BWD_DATA, B, H, W, C, 2 * H, 2 * W, 3, 5, 5, 2, 2, lib.EPI_TANH_BWD)


def test_engine_uses_sign_masks_and_matches_the_fp32_aux_path(monkeypatch):
    """dcgan64 / dcgan32: g_h3's forward leaves the mask, g_h4's backward-data reads it; the refinement is bit-equal to the
    engine built with CGS_NO_SIGN_MASKS (the fp32 aux read)."""
    from cgs_amd import engine as E
    from cgs_amd.nets import init_params
    d = dev()
    B, Ks = 128, 2
    P = init_params("dcgan64", d, seed=5)
    z = rnd((B, 100), 7).to(d)
    eng = E.RefineEngine("dcgan64", P, B, d)
    tail = [st for st in eng.g_tail.stages if isinstance(st, E._Deconv)]
    assert tail[-2].signs is not None and tail[-1].bwd_signs is tail[-2].signs
    got = [t.clone() for t in eng.refine_from_z(z, Ks, 0.1)]
    monkeypatch.setenv("CGS_NO_SIGN_MASKS", "1")
    ref = E.RefineEngine("dcgan64", P, B, d)
    assert all(st.signs is None for st in ref.g_tail.stages if isinstance(st, E._Deconv))
    want = ref.refine_from_z(z, Ks, 0.1)
    for a, b in zip(got, want):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,H,W,N", [(5, 28, 28, 64), (3, 27, 13, 32), (2, 10, 18, 96), (67, 14, 14, 128), (1, 2, 2, 64)])
def test_taps_form_of_the_4x4_conv_from_one_channel(B, H, W, N):
    """conv_taps_kernel (K = 16 taps on the 32x32x2 MFMA, no packed weights): forward with every forward epilogue, and as the
    backward-data of a 4x4 stride-2 deconv TO one channel with relu' / lrelu' / tanh' taken from the fp32 aux tensor and -- where
    a sign suffices -- from the sign mask, bit-equal to the aux form.  Odd sizes, a ragged last 32-pixel tile, 1..4 column tiles."""
    from cgs_amd import kernels as K, lib
    d = dev()
    x, w, b = rnd((B, H, W, 1), 1), rnd((4, 4, 1, N), 2, 0.2), rnd((N,), 3, 0.1)
    a, c = rnd((N,), 4).abs() + 0.5, rnd((N,), 5, 0.1)
    lin = R.conv2d(x, w, b, 2, 2)
    Hs, Ws = lin.shape[1], lin.shape[2]
    for epi, want in ((lib.EPI_NONE, lin), (lib.EPI_LRELU, R.lrelu(lin)), (lib.EPI_AFFINE_RELU, torch.relu(a * lin + c)), (lib.EPI_TANH, torch.tanh(lin))):
        got = K.conv2d_fwd(x.to(d), w.to(d), b.to(d), 2, 2, epi, a.to(d) if epi == lib.EPI_AFFINE_RELU else None,
                           c.to(d) if epi == lib.EPI_AFFINE_RELU else None)
        assert lib.last_kernel() == f"conv_taps_kernel<{N // 32}, 0>", lib.last_kernel()
        close(got, want, 2e-5)
    # deconv [4,4,Cout=1,Cin=N]: backward-data dy [B,H,W,1] -> dx [B,Hs,Ws,N] (needs H == 2 Hs: the even-size cases)
    if H % 2 == 0 and W % 2 == 0:
        wd = rnd((4, 4, 1, N), 6, 0.2)
        xs = rnd((B, Hs, Ws, N), 7).requires_grad_(True)
        dy = rnd((B, H, W, 1), 8)
        (R.deconv2d(xs, wd, torch.zeros(1), (B, H, W, 1), 2, 2) * dy).sum().backward()
        aux = rnd((B, Hs, Ws, N), 9)
        aux[aux.abs() < 0.3] = 0.0
        for e, want in ((lib.EPI_NONE, xs.grad), (lib.EPI_RELU_BWD_AFFINE, xs.grad * a * (aux > 0)),
                        (lib.EPI_LRELU_BWD, xs.grad * torch.where(aux > 0, 1.0, 0.2)), (lib.EPI_TANH_BWD, xs.grad * (1 - aux * aux))):
            got = K.deconv2d_bwd_data(dy.to(d), wd.to(d), (Hs, Ws), 2, 2, epilogue=e, ep_a=a.to(d) if e == lib.EPI_RELU_BWD_AFFINE else None,
                                      ep_aux=aux.to(d) if e != lib.EPI_NONE else None)
            assert lib.last_kernel() == f"conv_taps_kernel<{N // 32}, {0 if e == lib.EPI_NONE else 1}>", lib.last_kernel()
            close(got, want, 2e-5)
            if e in (lib.EPI_RELU_BWD_AFFINE, lib.EPI_LRELU_BWD):
                assert K.conv_signs_ok(lib.DECONV_BWD_DATA, B, Hs, Ws, N, H, W, 1, 4, 4, 2, 2, e)
                signs = torch.from_numpy(pack_signs((aux > 0).numpy().reshape(-1, N))).to(d)
                got2 = K.deconv2d_bwd_data(dy.to(d), wd.to(d), (Hs, Ws), 2, 2, epilogue=e, ep_a=a.to(d) if e == lib.EPI_RELU_BWD_AFFINE else None,
                                           ep_signs=signs)
                assert lib.last_kernel() == f"conv_taps_kernel<{N // 32}, 2>", lib.last_kernel()
                assert torch.equal(got2, got)


@pytest.mark.parametrize("B,H,W,Cin,N,k,s", [(8, 32, 32, 512, 1, 4, 1), (3, 9, 7, 128, 1, 3, 1), (2, 8, 8, 64, 3, 5, 2), (2, 6, 6, 256, 4, 3, 1),
                                             (1, 5, 5, 68, 2, 4, 1)])
def test_deep_reduction_small_n_conv(B, H, W, Cin, N, k, s):
    """conv_dot_kernel (a wave per output pixel; the PatchGAN logit head 4x4 x 512 -> 1 of config 5): forward with the forward
    epilogues, odd sizes, stride 2, 1..4 output channels, a channel count that is not a multiple of 256."""
    from cgs_amd import kernels as K, lib
    d = dev()
    x, w, b = rnd((B, H, W, Cin), 1), rnd((k, k, Cin, N), 2, 0.05), rnd((N,), 3, 0.1)
    lin = R.conv2d(x, w, b, s, s)
    for epi, want in ((lib.EPI_NONE, lin), (lib.EPI_LRELU, R.lrelu(lin)), (lib.EPI_TANH, torch.tanh(lin))):
        got = K.conv2d_fwd(x.to(d), w.to(d), b.to(d), s, s, epi)
        assert lib.last_kernel() == f"conv_dot_kernel<{N}>", lib.last_kernel()
        # (tanh has slope 1 at 0: its output carries the ABSOLUTE error of the pre-activation, 2e-5 of max|lin| over a K <= 8192 reduction)
        close(got, want, 2e-5 * (max(1.0, lin.abs().max().item()) if epi == lib.EPI_TANH else 1.0))
    got = K.conv2d_fwd(x.to(d), w.to(d), None, s, s)
    close(got, R.conv2d(x, w, torch.zeros(N), s, s), 2e-5)


def test_fused_logit_head_and_row_selects_equal_their_parts():
    """Round 5 launch-count fusions of the loop's per-step bookkeeping: cgs_linear_out1_bce = linear_fwd(N = 1) + bce_ones_grad_rowmean,
    cgs_refine_select2 = refine_select_rows + refine_select -- bit for bit, deterministic and probabilistic mode."""
    from cgs_amd import kernels as K
    d = dev()
    B, Kin = 77, 1024
    x, w, b = rnd((B, Kin), 1).to(d), rnd((Kin, 1), 2, 0.1).to(d), rnd((1,), 3).to(d)
    y = K.linear_fwd(x, w, b)
    dl, lm = K.bce_ones_grad_rowmean(y)
    y2, dl2, lm2 = torch.empty_like(y), torch.empty_like(y), torch.empty(B, device=d)
    K.linear_out1_bce(x, w, b, y2, dl2, lm2)
    assert torch.equal(y, y2) and torch.equal(dl, dl2) and torch.equal(lm, lm2)
    for forced in (None, torch.randint(0, 4, (B,), dtype=torch.int32, device=d)):
        rows, theta, logit = rnd((B, 28, 28, 1), 4).to(d), rnd((B, 7, 7, 16), 5).to(d), rnd((B,), 6).to(d)
        best = rnd((B,), 7).to(d)
        a = [torch.zeros_like(rows), torch.zeros_like(theta), best.clone(), torch.ones(B, device=d)]
        c = [t.clone() for t in a]
        K.refine_select_rows(rows, logit, forced, 2, a[0], a[2])
        K.refine_select(theta, logit, forced, 2, a[1], a[2], a[3])
        K.refine_select2(rows, c[0], theta, c[1], logit, forced, 2, c[2], c[3])
        assert all(torch.equal(u, v) for u, v in zip(a, c))
        # ... and with the scalars in the same launch (the last block of a selected sample writes them; the tickets come back zero)
        e = [torch.zeros_like(rows), torch.zeros_like(theta), best.clone(), torch.ones(B, device=d)]
        tickets = torch.zeros(B, dtype=torch.int32, device=d)
        for _ in range(2):                 # (twice: the second call selects nothing new in deterministic mode, the same rows in forced mode)
            K.refine_select2(rows, e[0], theta, e[1], logit, forced, 2, e[2], e[3], tickets)
        assert all(torch.equal(u, v) for u, v in zip(a, e)) and not tickets.any()
        upd = (forced == 2) if forced is not None else (logit > best)
        assert torch.equal(c[3], torch.where(upd, torch.full_like(best, 3.0), torch.ones_like(best))) and upd.any() and not upd.all()
    # maps of more than 64 x 1024 elements per sample (config 5's 64x64x256): the ticketed launch strides 64 blocks per sample over the rows
    B2 = 5
    rows, theta, logit, best = rnd((B2, 32, 32, 3), 8).to(d), rnd((B2, 48, 48, 64), 9).to(d), rnd((B2,), 10).to(d), rnd((B2,), 11).to(d)
    a = [torch.zeros_like(rows), torch.zeros_like(theta), best.clone(), torch.ones(B2, device=d)]
    c = [t.clone() for t in a]
    tickets = torch.zeros(B2, dtype=torch.int32, device=d)
    K.refine_select2(rows, a[0], theta, a[1], logit, None, 6, a[2], a[3])
    K.refine_select2(rows, c[0], theta, c[1], logit, None, 6, c[2], c[3], tickets)
    upd = logit > best
    assert all(torch.equal(u, v) for u, v in zip(a, c)) and not tickets.any() and upd.any() and not upd.all()
    assert torch.equal(c[1][upd], theta[upd]) and not c[1][~upd].any() and torch.equal(c[3], torch.where(upd, torch.full_like(best, 7.0), torch.ones_like(best)))
