"""Seeded random shapes through the C ABI against the CPU oracle: odd sizes, k in {1,3,4,5,7}, stride 1/2, channel counts on
and off the vector paths (1, 3, 4, 24, 32, 48, 64, 96, 160), batch 1..130 -- forward, backward-data and the fused
epilogues of conv2d / conv2d_transpose.  24 cases per op by default (the torch-CPU checker dominates the run time); every case names its shape in the test id."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ops_ref as R

CH = [1, 3, 4, 24, 32, 48, 64, 96, 160]
N_CASES = int(os.environ.get("CGS_FUZZ_N", "24"))          # CGS_FUZZ_N=400 for a longer hunt
SEED = int(os.environ.get("CGS_FUZZ_SEED", "0"))


@pytest.fixture(autouse=True)
def _plain_cpu_convolutions():
    """The checker runs on torch-CPU; its oneDNN convolution backward corrupts the heap on some degenerate shapes (1x1 kernel,
    stride 2, one input channel: 'double free or corruption' in this image's torch 2.10), which a random-shape hunt does reach.
    The native CPU kernels are slower and fine."""
    with torch.backends.mkldnn.flags(enabled=False):
        yield


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).float()


def ktol(k, c):
    """2e-5 for reductions up to 1000 terms, growing with the square root of the length beyond (fp32 summation order)."""
    return 2e-5 * max(1.0, (k * k * c / 1000.0) ** 0.5)


def close(got, want, tol=2e-5):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item() + 1e-30
    assert err <= tol * ref, f"max|delta|={err:.3e} vs max|ref|={ref:.3e}"


def cases(seed, n):
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        B = int(rs.choice([1, 2, 5, 130]))
        k, s = int(rs.choice([1, 3, 4, 5, 7])), int(rs.choice([1, 2]))
        H, W = int(rs.randint(1, 19)), int(rs.randint(1, 19))
        cin, cout = int(rs.choice(CH)), int(rs.choice(CH))
        if B * H * W * max(cin, cout) > 3_000_000 or B * H * W * cin * cout * k * k > 6e9:
            continue
        out.append((B, H, W, cin, cout, k, s))
    return out


def wide_cases(seed, n):
    """Channel counts the split-bf16 kernel serves (a reduction over whole 32-channel chunks into a multiple of 128 channels, in at
    least one direction): 128 x 256 and 256 x 128 block tiles, ragged and one-pixel tiles, every kernel size and both strides."""
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        B = int(rs.choice([1, 3, 130, 256]))
        k, s = int(rs.choice([1, 3, 4, 5, 7])), int(rs.choice([1, 2]))
        H, W = int(rs.randint(1, 13)), int(rs.randint(1, 13))
        cin, cout = int(rs.choice([32, 64, 96, 128, 256])), int(rs.choice([128, 256, 384]))
        if rs.randint(2):
            cin, cout = cout, cin
        if B * H * W * max(cin, cout) > 3_000_000 or B * H * W * cin * cout * k * k > 1.2e9:      # (the checker's CPU backward takes a minute per 5e9 here)
            continue
        out.append((B, H, W, cin, cout, k, s))
    return out


N_WIDE = max(8, N_CASES // 2)


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s", cases(11 + SEED, N_CASES) + wide_cases(111 + SEED, N_WIDE))
def test_conv_random_shapes(B, H, W, Cin, Cout, k, s, contraction):
    from cgs_amd import kernels as K, lib
    d = dev()
    x = rnd((B, H, W, Cin), 1).requires_grad_(True)
    w, b = rnd((k, k, Cin, Cout), 2, 0.1), rnd((Cout,), 3, 0.2)
    want = R.lrelu(R.conv2d(x, w, b, s, s))
    got = K.conv2d_fwd(x.detach().to(d), w.to(d), b.to(d), s, s, lib.EPI_LRELU)
    assert tuple(got.shape) == tuple(want.shape)
    close(got, want, ktol(k, Cin))
    y = R.conv2d(x, w, torch.zeros(Cout), s, s)
    dy = rnd(tuple(y.shape), 4)
    (y * dy).sum().backward()
    close(K.conv2d_bwd_data(dy.to(d), w.to(d), (H, W), s, s), x.grad, ktol(k, Cout))


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s", cases(23 + SEED, N_CASES) + wide_cases(123 + SEED, N_WIDE))
def test_deconv_random_shapes(B, H, W, Cin, Cout, k, s, contraction):
    """conv2d_transpose from [B,H,W,Cin] to an output whose SAME conv maps back to (H, W): both admissible output sizes
    for stride 2 (2H and 2H-1) are drawn."""
    from cgs_amd import kernels as K, lib
    d = dev()
    rs = np.random.RandomState(B * 1000 + H * 37 + W)
    Ho = H * s - (int(rs.randint(0, s)) if H * s > 1 else 0)
    Wo = W * s - (int(rs.randint(0, s)) if W * s > 1 else 0)
    x = rnd((B, H, W, Cin), 1).requires_grad_(True)
    w, b = rnd((k, k, Cout, Cin), 2, 0.1), rnd((Cout,), 3, 0.2)
    want = torch.tanh(R.deconv2d(x, w, b, (B, Ho, Wo, Cout), s, s))
    got = K.deconv2d_fwd(x.detach().to(d), w.to(d), b.to(d), (Ho, Wo), s, s, lib.EPI_TANH)
    close(got, want, 1.5 * ktol(k, Cin))          # tanh of sums several units large: the slope near 0 passes the sum's rounding through
    y = R.deconv2d(x, w, torch.zeros(Cout), (B, Ho, Wo, Cout), s, s)
    dy = rnd(tuple(y.shape), 4)
    (y * dy).sum().backward()
    close(K.deconv2d_bwd_data(dy.to(d), w.to(d), (H, W), s, s), x.grad, ktol(k, Cout))


def lin_cases(seed, n):
    rs = np.random.RandomState(seed)
    return [(int(rs.choice([1, 3, 64, 200])), int(rs.choice([1, 7, 32, 100, 1024, 6272])), int(rs.choice([1, 3, 10, 64, 1000])))
            for _ in range(n)]


@pytest.mark.parametrize("B,fin,fout", lin_cases(5 + SEED, max(10, N_CASES // 2)))
def test_linear_random_shapes(B, fin, fout):
    from cgs_amd import kernels as K
    d = dev()
    x = rnd((B, fin), 1).requires_grad_(True)
    w, b = rnd((fin, fout), 2, 0.1), rnd((fout,), 3, 0.2)
    y = R.linear(x, w, b)
    close(K.linear_fwd(x.detach().to(d), w.to(d), b.to(d)), y, ktol(1, fin))
    dy = rnd((B, fout), 4)
    (y * dy).sum().backward()
    close(K.linear_bwd_data(dy.to(d), w.to(d)), x.grad, ktol(1, fout))


def bn_cases(seed, n):
    rs = np.random.RandomState(seed)
    return [(int(rs.choice([1, 2, 17, 64, 1000, 5000])), int(rs.choice([4, 8, 64, 100, 128, 1024])), float(rs.choice([0.2, 1.0, 0.0])))
            for _ in range(n)]


@pytest.mark.parametrize("M,C,leak", bn_cases(9 + SEED, max(10, N_CASES // 2)))
def test_bn_random_shapes(M, C, leak):
    from cgs_amd import kernels as K
    d = dev()
    x = (rnd((M, C), 1) * 1.3 + 0.4).requires_grad_(True)
    gamma, beta = rnd((C,), 2).abs() + 0.5, rnd((C,), 3, 0.3)
    z = R.bn_train(x, gamma, beta)
    want = torch.where(z > 0, z, leak * z)
    y, mean, invstd = K.bn_train_lrelu_fwd(x.detach().to(d), gamma.to(d), beta.to(d), leak)
    if M == 1:                              # a single row: variance 0, output = beta through the activation
        close(y, want, 1e-4)
        return
    close(y, want, 1e-5 if M > 2 else 1e-3)
    dy = rnd((M, C), 4)
    (want * dy).sum().backward()
    got = K.bn_train_lrelu_bwd_data(dy.to(d), x.detach().to(d), gamma.to(d), beta.to(d), mean, invstd, leak)
    # rows whose pre-activation sits within rounding of the lrelu kink may take the other slope: compare away from it
    safe = (z.detach().abs() > 1e-4).all(dim=1) if leak != 1.0 else torch.ones(M, dtype=torch.bool)
    if safe.all():
        close(got, x.grad, 2e-4 if M > 2 else 2e-2)


def wgrad_cases(seed, n):
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        B, k, s = int(rs.choice([1, 3, 33])), int(rs.choice([1, 3, 4, 5])), int(rs.choice([1, 2]))
        H, W = int(rs.randint(1, 15)), int(rs.randint(1, 15))
        cin, cout = int(rs.choice(CH)), int(rs.choice(CH))
        if B * H * W * cin * cout * k * k > 3e9:
            continue
        out.append((B, H, W, cin, cout, k, s))
    return out


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s", wgrad_cases(31 + SEED, max(10, N_CASES // 2)))
def test_conv_weight_gradient_random_shapes(B, H, W, Cin, Cout, k, s):
    from cgs_amd import kernels as K
    d = dev()
    x = rnd((B, H, W, Cin), 1)
    w = rnd((k, k, Cin, Cout), 2, 0.1).requires_grad_(True)
    b = torch.zeros(Cout, requires_grad=True)
    y = R.conv2d(x, w, b, s, s)
    dy = rnd(tuple(y.shape), 4)
    (y * dy).sum().backward()
    got = K.conv2d_bwd_weight(x.to(d), dy.to(d), k, k, s, s)
    close(got, w.grad, 2e-5 * max(1.0, (B * y.shape[1] * y.shape[2] / 1000.0) ** 0.5))
    close(K.bias_grad(dy.to(d)), b.grad, 1e-5 * max(1.0, (B * y.shape[1] * y.shape[2] / 1000.0) ** 0.5))


def big_cases(seed, n):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        H = int(rs.choice([8, 16, 32]))
        out.append((1024 if H <= 16 else 256, H, int(rs.choice([32, 64, 96])), int(rs.choice([32, 64, 128])), int(rs.choice([3, 4, 5]))))
    return out


def big_wide_cases(seed, n):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        H = int(rs.choice([4, 8, 16]))
        cin, cout = int(rs.choice([64, 128])), int(rs.choice([128, 256]))
        if rs.randint(2):
            cin, cout = cout, cin
        out.append((1024 if H <= 8 else 256, H, cin, cout, int(rs.choice([3, 4, 5]))))
    return out


@pytest.mark.parametrize("B,H,Cin,Cout,k", big_cases(77 + SEED, max(6, N_CASES // 4)) + big_wide_cases(177 + SEED, max(6, N_CASES // 4)))
def test_many_block_launch_equals_its_small_batch_pieces(B, H, Cin, Cout, k, contraction):
    """Large grids take other code paths than small ones (16-deep K tiles at four blocks per CU, per-XCD block decode, pixel-major
    rows with zero-tap skipping, parity-first taps) and are too big for the CPU oracle.  A convolution is independent per sample,
    so the big launch must agree with the same op run on 64-sample pieces -- which take the small-grid paths the oracle tests
    pin.  (Same tolerance as against the oracle: the pieces may split K and so add in another order.)
    (In the split-bf16 mode the big launch runs igemm_bx6 and the pieces stay on the exact-fp32 kernels: the same comparison pins
    the new kernel's large-grid paths to the oracle-tested ones.)"""
    from cgs_amd import kernels as K
    d = dev()
    x = rnd((B, H, H, Cin), 1).to(d)
    w, b = rnd((k, k, Cin, Cout), 2, 0.1).to(d), rnd((Cout,), 3, 0.2).to(d)
    mode = K.CONTRACTION

    def pieces(fn, *ts):
        K.set_contraction("f32")
        try:
            return torch.cat([fn(*[t[i:i + 64].contiguous() for t in ts]) for i in range(0, B, 64)])
        finally:
            K.set_contraction(mode)
    y = K.conv2d_fwd(x, w, b, 2, 2)
    assert torch.allclose(y, pieces(lambda xx: K.conv2d_fwd(xx, w, b, 2, 2), x), rtol=0, atol=2e-5 * y.abs().max().item())
    dy = rnd(tuple(y.shape), 4).to(d)
    dx = K.conv2d_bwd_data(dy, w, (H, H), 2, 2)
    assert torch.allclose(dx, pieces(lambda g: K.conv2d_bwd_data(g, w, (H, H), 2, 2), dy), rtol=0, atol=2e-5 * dx.abs().max().item())
    wt = rnd((k, k, Cout, Cin), 5, 0.1).to(d)                      # deconv Cin -> Cout at twice the resolution
    z = K.deconv2d_fwd(x, wt, b, (2 * H, 2 * H), 2, 2)
    assert torch.allclose(z, pieces(lambda xx: K.deconv2d_fwd(xx, wt, b, (2 * H, 2 * H), 2, 2), x), rtol=0, atol=2e-5 * z.abs().max().item())
    dz = rnd(tuple(z.shape), 6).to(d)
    dxx = K.deconv2d_bwd_data(dz, wt, (H, H), 2, 2)
    assert torch.allclose(dxx, pieces(lambda g: K.deconv2d_bwd_data(g, wt, (H, H), 2, 2), dz), rtol=0, atol=2e-5 * dxx.abs().max().item())


def nstat_cases(seed, n):
    """Random (layer above the norm, batch, gradient map, channels, kernel, stride, images per statistics group, leak) for the norm-backward
    sums fused into a backward-data launch: channel counts on and off the 32-channel vector path, ragged last tiles (M % 128 != 0), whole
    batch / per sample / groups of samples, batches that take the pixel-major order and that are split over K."""
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        op = str(rs.choice(["conv", "deconv"]))
        B = int(rs.choice([1, 2, 3, 6, 16, 64, 128, 256]))
        H = int(rs.choice([4, 6, 8, 12, 16]))
        C = int(rs.choice([4, 8, 32, 64, 96, 128]))
        Cab = int(rs.choice([3, 32, 36, 64, 128, 256]))
        k, s = int(rs.choice([3, 4, 5])), int(rs.choice([1, 2]))
        if op == "deconv" and s == 1:
            s = 2
        grp = int(rs.choice([g for g in (1, 2, 64, B) if B % g == 0]))
        leak = float(rs.choice([0.0, 0.2, 1.0]))
        if B * H * H * max(C, Cab) * (s * s if op == "deconv" else 1) > 6_000_000:
            continue
        out.append((op, B, H, C, Cab, k, s, grp, leak))
    return out


@pytest.mark.parametrize("case", nstat_cases(4000 + SEED, max(12, N_CASES // 2)), ids=lambda c: "-".join(str(v) for v in c))
def test_fuzz_norm_backward_sums_in_the_backward_data_epilogue(case):
    """cgs_conv_stat_layout(*_BWD_DATA) + cgs_*_bwd_data_nstats + cgs_norm_lrelu_bwd_from_partials on random shapes: wherever the library offers
    the fusion, the norm's input gradient from the fused sums equals the separate-pass kernels' (1e-5) -- which the oracle pins -- and the
    gradient map itself is the plain backward-data's."""
    from cgs_amd import kernels as K, lib
    op, B, H, C, Cab, k, s_, grp, leak = case
    d = dev()
    if op == "conv":
        Ho = -(-H // s_)
        lay = K.conv_stat_layout(lib.CONV_BWD_DATA, B, H, H, C, 0, 0, Cab, k, k, s_, s_, grp)
    else:
        Ho = H * s_
        lay = K.conv_stat_layout(lib.DECONV_BWD_DATA, B, H, H, C, Ho, Ho, Cab, k, k, s_, s_, grp)
    if lay is None:
        pytest.skip("the fusion is not offered for this call (kernel family, channel count, or a group that does not own whole 64-row pieces)")
    groups = B // grp
    x = rnd((B, H, H, C), 1).to(d)
    gamma, beta = (rnd((C,), 4, 0.2) + 1).to(d), rnd((C,), 5, 0.1).to(d)
    _, mean, invstd = K.instnorm_lrelu_fwd(x.view(groups, -1, C), gamma, beta, leak)
    g = rnd((B, Ho, Ho, Cab), 3).to(d)
    part = torch.full((lay[0], 2, C), float("nan"), device=d)
    ns = K.NormBwdStats(x, mean, invstd, gamma, beta, leak, grp, part, lay)
    if op == "conv":
        w = rnd((k, k, C, Cab), 2, 0.05).to(d)
        dy, dy_plain = K.conv2d_bwd_data(g, w, (H, H), s_, s_, nstat=ns), K.conv2d_bwd_data(g, w, (H, H), s_, s_)
    else:
        w = rnd((k, k, Cab, C), 2, 0.05).to(d)
        dy, dy_plain = K.deconv2d_bwd_data(g, w, (H, H), s_, s_, nstat=ns), K.deconv2d_bwd_data(g, w, (H, H), s_, s_)
    close(dy, dy_plain, 2e-6)
    rows, rps, nseg, stride = lay
    used = sorted({sg * stride + gi * rps + i for gi in range(groups) for sg in range(nseg) for i in range(rps)})
    assert torch.isfinite(part[torch.tensor(used, device=d)]).all()              # every row a group owns was written
    got = K.norm_lrelu_bwd_from_partials(dy.clone(), x, ns, groups)
    want = K.instnorm_lrelu_bwd_data(dy.view(groups, -1, C).clone(), x.view(groups, -1, C), gamma, beta, mean, invstd, leak)
    close(got.reshape(-1), want.reshape(-1), 1e-5)
