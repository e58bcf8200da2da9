"""Network oracle: the G head / G tail / D the refiner differentiates.

``mnist`` restates nsgan/GAN.py:59-101 (infoGAN architecture, 4x4 stride-2
kernels).  ``dcgan32`` / ``dcgan64`` are NOT in the reference tree (SURVEY.md
fact 2): they are build-defined from the nsgan/ops.py defaults (5x5, stride 2,
'SAME'; nsgan/ops.py:37,48) in the carpedm20 DCGAN-tensorflow convention the
reference acknowledges (README.md:75-76), with the refine point at G.h1
(SURVEY.md 8d).  Parameter names follow the TF variable scopes
(nsgan/ops.py:38-43,49-61,75-79).

Layer lists are interpreted by ``run_layers``; G batch-norm runs on moving
averages (is_training=False, nsgan/GAN.py:87,94), D batch-norm on batch
statistics (is_training=True, nsgan/GAN.py:175).
"""
import numpy as np
import torch

from . import ops_ref as R


def _dcgan(img, z_dim=100):
    s = img // 16
    return dict(
        z_dim=z_dim, img=(img, img, 3), k=5,
        g_head=[("linear", "g_h0_lin", s * s * 512), ("reshape", (s, s, 512)), ("bn", "g_bn0"), ("relu",),
                ("deconv", "g_h1", (2 * s, 2 * s, 256)), ("bn", "g_bn1"), ("relu",)],
        g_tail=[("deconv", "g_h2", (4 * s, 4 * s, 128)), ("bn", "g_bn2"), ("relu",),
                ("deconv", "g_h3", (8 * s, 8 * s, 64)), ("bn", "g_bn3"), ("relu",),
                ("deconv", "g_h4", (16 * s, 16 * s, 3)), ("tanh",)],
        d=[("conv", "d_h0_conv", 64), ("lrelu",),
           ("conv", "d_h1_conv", 128), ("bn", "d_bn1"), ("lrelu",),
           ("conv", "d_h2_conv", 256), ("bn", "d_bn2"), ("lrelu",),
           ("conv", "d_h3_conv", 512), ("bn", "d_bn3"), ("lrelu",),
           ("flatten",), ("linear", "d_h4_lin", 1)],
        feature=(2 * s, 2 * s, 256),
    )


ARCHS = {
    # nsgan/GAN.py:59-101
    "mnist": dict(
        z_dim=62, img=(28, 28, 1), k=4,
        g_head=[("linear", "g_fc1", 1024), ("bn", "g_bn1"), ("relu",),
                ("linear", "g_fc2", 128 * 7 * 7), ("bn", "g_bn2"), ("relu",), ("reshape", (7, 7, 128))],
        g_tail=[("deconv", "g_dc3", (14, 14, 64)), ("bn", "g_bn3"), ("relu",),
                ("deconv", "g_dc4", (28, 28, 1)), ("tanh",)],
        d=[("conv", "d_conv1", 64), ("lrelu",),
           ("conv", "d_conv2", 128), ("bn", "d_bn2"), ("lrelu",),
           ("flatten",), ("linear", "d_fc3", 1024), ("bn", "d_bn3"), ("lrelu",),
           ("linear", "d_fc4", 1)],
        feature=(7, 7, 128),
    ),
    "dcgan32": _dcgan(32),
    "dcgan64": _dcgan(64),
}


def _cyclegan(img, n_res, ngf=64, ndf=64):
    """CycleGAN ResNet G / PatchGAN D with instance norm (BASELINE config 5).  NOT in the reference tree and without
    reference code of any kind: parity for it is oracle-vs-kernel only (unpinned)."""
    q = img // 4
    res = [("res", [("conv", f"g_r{i}_c1", 4 * ngf, 3, 1), ("instnorm", f"g_r{i}_in1"), ("relu",),
                    ("conv", f"g_r{i}_c2", 4 * ngf, 3, 1), ("instnorm", f"g_r{i}_in2")]) for i in range(n_res)]
    return dict(
        z_dim=None, g_in=(img, img, 3), img=(img, img, 3), k=3,
        g_head=[("conv", "g_c1", ngf, 7, 1), ("instnorm", "g_in1"), ("relu",),
                ("conv", "g_d1", 2 * ngf, 3, 2), ("instnorm", "g_in2"), ("relu",),
                ("conv", "g_d2", 4 * ngf, 3, 2), ("instnorm", "g_in3"), ("relu",)] + res,
        g_tail=[("deconv", "g_u1", (img // 2, img // 2, 2 * ngf), 3, 2), ("instnorm", "g_in4"), ("relu",),
                ("deconv", "g_u2", (img, img, ngf), 3, 2), ("instnorm", "g_in5"), ("relu",),
                ("conv", "g_c2", 3, 7, 1), ("tanh",)],
        d=[("conv", "d_c1", ndf, 4, 2), ("lrelu",),
           ("conv", "d_c2", 2 * ndf, 4, 2), ("instnorm", "d_in2"), ("lrelu",),
           ("conv", "d_c3", 4 * ndf, 4, 2), ("instnorm", "d_in3"), ("lrelu",),
           ("conv", "d_c4", 8 * ndf, 4, 1), ("instnorm", "d_in4"), ("lrelu",),
           ("conv", "d_c5", 1, 4, 1)],
        feature=(q, q, 4 * ngf))


ARCHS["cyclegan256"] = _cyclegan(256, 9)
ARCHS["cyclegan_tiny"] = _cyclegan(32, 2, ngf=16, ndf=16)


def _ks(L, k):
    return (L[3], L[4]) if len(L) >= 5 else (k, 2)


def _walk_shapes(layers, shape, k, scope, out):
    """Collect parameter shapes by walking a layer list from an input shape (no batch dim)."""
    for L in layers:
        kind = L[0]
        if kind == "linear":
            out[f"{scope}/{L[1]}/Matrix"] = (int(np.prod(shape)), L[2]); out[f"{scope}/{L[1]}/bias"] = (L[2],)
            shape = (L[2],)
        elif kind == "reshape":
            shape = tuple(L[1])
        elif kind == "flatten":
            shape = (int(np.prod(shape)),)
        elif kind == "conv":
            kk, ss = _ks(L, k)
            out[f"{scope}/{L[1]}/w"] = (kk, kk, shape[-1], L[2]); out[f"{scope}/{L[1]}/biases"] = (L[2],)
            shape = (R.conv_out_size_same(shape[0], ss), R.conv_out_size_same(shape[1], ss), L[2])
        elif kind == "deconv":
            kk, ss = _ks(L, k)
            out[f"{scope}/{L[1]}/w"] = (kk, kk, L[2][2], shape[-1]); out[f"{scope}/{L[1]}/biases"] = (L[2][2],)
            shape = tuple(L[2])
        elif kind == "instnorm":
            out[f"{scope}/{L[1]}/scale"] = (shape[-1],); out[f"{scope}/{L[1]}/offset"] = (shape[-1],)
        elif kind == "res":
            assert tuple(_walk_shapes(L[1], shape, k, scope, out)) == tuple(shape)
        elif kind == "bn":
            for v in ("beta", "gamma", "moving_mean", "moving_variance"):
                out[f"{scope}/{L[1]}/{v}"] = (shape[-1],)
    return shape


def param_shapes(arch):
    A = ARCHS[arch]
    out = {}
    feat = _walk_shapes(A["g_head"], tuple(A["g_in"]) if A.get("g_in") else (A["z_dim"],), A["k"], "generator", out)
    assert tuple(feat) == tuple(A["feature"])
    img = _walk_shapes(A["g_tail"], feat, A["k"], "generator", out)
    assert tuple(img) == tuple(A["img"])
    _walk_shapes(A["d"], img, A["k"], "discriminator", out)
    return out


def init_params(arch, seed=2019, perturb=True):
    """Random-init parameters as nsgan/ops.py does (conv trunc-normal sigma .02 :40,
    deconv/linear normal sigma .02 :52,77, biases 0 :43,61,79; BN gamma 1, beta 0,
    moving stats (0,1)).  ``perturb`` draws non-degenerate BN affine / moving
    statistics and small biases so that every term of the math is exercised."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for name, shp in param_shapes(arch).items():
        leaf = name.rsplit("/", 1)[1]
        if leaf == "w" and "conv" in name:
            t = torch.empty(shp)
            torch.nn.init.trunc_normal_(t, 0.0, 0.02, -0.04, 0.04, generator=g)
        elif leaf in ("w", "Matrix"):
            t = torch.randn(shp, generator=g) * 0.02
        elif leaf in ("biases", "bias", "beta"):
            t = torch.randn(shp, generator=g) * 0.02 if perturb else torch.zeros(shp)
        elif leaf in ("gamma", "scale"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g) if perturb else torch.ones(shp)
        elif leaf == "offset":
            t = torch.randn(shp, generator=g) * 0.02 if perturb else torch.zeros(shp)
        elif leaf == "moving_mean":
            t = 0.05 * torch.randn(shp, generator=g) if perturb else torch.zeros(shp)
        elif leaf == "moving_variance":
            t = (1.0 + 0.2 * torch.rand(shp, generator=g)) * 0.02 if perturb else torch.ones(shp)
        else:
            raise KeyError(name)
        P[name] = t.float()
    return P


def run_layers(layers, x, P, scope, bn_training, k_stride=2):
    for L in layers:
        kind = L[0]
        if kind == "linear":
            x = R.linear(x, P[f"{scope}/{L[1]}/Matrix"], P[f"{scope}/{L[1]}/bias"])
        elif kind == "reshape":
            x = x.reshape((x.shape[0],) + tuple(L[1]))
        elif kind == "flatten":
            x = x.reshape(x.shape[0], -1)          # NHWC -> (h,w,c)-major rows, nsgan/GAN.py:66
        elif kind == "conv":
            ss = L[4] if len(L) >= 5 else k_stride
            x = R.conv2d(x, P[f"{scope}/{L[1]}/w"], P[f"{scope}/{L[1]}/biases"], ss, ss)
        elif kind == "deconv":
            ss = L[4] if len(L) >= 5 else k_stride
            x = R.deconv2d(x, P[f"{scope}/{L[1]}/w"], P[f"{scope}/{L[1]}/biases"],
                           (x.shape[0],) + tuple(L[2]), ss, ss)
        elif kind == "instnorm":
            s_ = f"{scope}/{L[1]}"
            x = R.instance_norm(x, P[s_ + "/scale"], P[s_ + "/offset"])
        elif kind == "res":
            x = x + run_layers(L[1], x, P, scope, bn_training, k_stride)
        elif kind == "bn":
            s = f"{scope}/{L[1]}"
            if bn_training:
                x = R.bn_train(x, P[s + "/gamma"], P[s + "/beta"])
            else:
                x = R.bn_infer(x, P[s + "/gamma"], P[s + "/beta"], P[s + "/moving_mean"], P[s + "/moving_variance"])
        elif kind == "relu":
            x = torch.relu(x)
        elif kind == "lrelu":
            x = R.lrelu(x)
        elif kind == "tanh":
            x = torch.tanh(x)
        else:
            raise KeyError(kind)
    return x


def input_to_feature(arch, P, z):
    """nsgan/GAN.py:87-92 (G head, BN on moving averages)."""
    return run_layers(ARCHS[arch]["g_head"], z, P, "generator", bn_training=False)


def feature_to_data(arch, P, feat):
    """nsgan/GAN.py:94-101 (G tail)."""
    return run_layers(ARCHS[arch]["g_tail"], feat, P, "generator", bn_training=False)


def discriminator(arch, P, x):
    """nsgan/GAN.py:59-70 with is_training=True (bound at :175)."""
    return run_layers(ARCHS[arch]["d"], x, P, "discriminator", bn_training=True)


def macs_per_sample(arch):
    """conv/deconv/fc MACs of (G tail, D) per sample (SURVEY 8d FLOP model)."""
    A = ARCHS[arch]

    def walk(layers, shape):
        m = 0
        for L in layers:
            if L[0] == "conv":
                kk, ss = _ks(L, A["k"])
                o = (R.conv_out_size_same(shape[0], ss), R.conv_out_size_same(shape[1], ss), L[2])
                m += o[0] * o[1] * o[2] * kk ** 2 * shape[2]; shape = o
            elif L[0] == "deconv":
                kk, ss = _ks(L, A["k"])
                m += shape[0] * shape[1] * shape[2] * kk ** 2 * L[2][2]; shape = tuple(L[2])
            elif L[0] == "res":
                m += walk(L[1], shape)
            elif L[0] == "linear":
                m += int(np.prod(shape)) * L[2]; shape = (L[2],)
            elif L[0] == "flatten":
                shape = (int(np.prod(shape)),)
        return m
    return walk(A["g_tail"], A["feature"]), walk(A["d"], A["img"])
