"""Networks the refiner differentiates, as data (layer lists) + device parameter stores.

``mnist`` is the reference's own NS-GAN (nsgan/GAN.py:59-101).  ``dcgan32`` / ``dcgan64`` are the
CIFAR-10 / CelebA DCGANs of BASELINE configs 2-4; the reference ships no code for them (its
README links a missing image/ package), so they are defined here from the nsgan/ops.py defaults
(5x5 kernels, stride 2, 'SAME'; nsgan/ops.py:37,48) in the carpedm20 DCGAN-tensorflow layout the
reference credits (nsgan/ops.py:1-3), refined at G.h1.

A layer list is a sequence of tuples:
    ("linear", scope, out) ("reshape", (h,w,c)) ("flatten",) ("conv", scope, cout[, k, stride])
    ("deconv", scope, (ho,wo,cout)[, k, stride]) ("bn", scope) ("instnorm", scope) ("relu",) ("lrelu",) ("tanh",)
    ("res", [inner layers])  -- a residual block  x + F(x)
(conv / deconv take the arch-wide kernel size and stride unless they carry their own.)
Parameter names are the TF variable names (``generator/g_dc3/w`` ...; nsgan/ops.py:38-43,49-61,75-79)
so a converted checkpoint drops in unchanged.
"""
import math

import numpy as np
import torch


def _dcgan(img, z_dim=100):
    s = img // 16
    return dict(
        z_dim=z_dim, img=(img, img, 3), k=5, stride=2,
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
        feature=(2 * s, 2 * s, 256))


ARCHS = {
    "mnist": dict(
        z_dim=62, img=(28, 28, 1), k=4, stride=2,
        g_head=[("linear", "g_fc1", 1024), ("bn", "g_bn1"), ("relu",),
                ("linear", "g_fc2", 128 * 7 * 7), ("bn", "g_bn2"), ("relu",), ("reshape", (7, 7, 128))],
        g_tail=[("deconv", "g_dc3", (14, 14, 64)), ("bn", "g_bn3"), ("relu",),
                ("deconv", "g_dc4", (28, 28, 1)), ("tanh",)],
        d=[("conv", "d_conv1", 64), ("lrelu",),
           ("conv", "d_conv2", 128), ("bn", "d_bn2"), ("lrelu",),
           ("flatten",), ("linear", "d_fc3", 1024), ("bn", "d_bn3"), ("lrelu",),
           ("linear", "d_fc4", 1)],
        feature=(7, 7, 128)),
    "dcgan32": _dcgan(32),
    "dcgan64": _dcgan(64),
}


def cyclegan(img=256, n_res=9, ngf=64, ndf=64):
    """BASELINE config 5: a CycleGAN ResNet generator (c7s1-64, d128, d256, R256 x n, u128, u64, c7s1-3) refined at the
    output of its residual trunk, and the 70x70 PatchGAN discriminator (C64-C128-C256-C512-1), both with instance norm.
    The reference ships NO code for this config: the layers are defined here with the
    operator defaults of nsgan/ops.py ('SAME' zero padding instead of the paper's reflection padding); the
    generator's input is an image, so ``z_dim`` is replaced by ``g_in`` = the image shape."""
    q = img // 4
    res = [("res", [("conv", f"g_r{i}_c1", 4 * ngf, 3, 1), ("instnorm", f"g_r{i}_in1"), ("relu",),
                    ("conv", f"g_r{i}_c2", 4 * ngf, 3, 1), ("instnorm", f"g_r{i}_in2")]) for i in range(n_res)]
    return dict(
        z_dim=None, g_in=(img, img, 3), img=(img, img, 3), k=3, stride=2,
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


ARCHS["cyclegan256"] = cyclegan(256, 9)
ARCHS["cyclegan_tiny"] = cyclegan(32, 2, ngf=16, ndf=16)       # same topology, test-sized


def _same_out(size, stride):
    return int(math.ceil(float(size) / float(stride)))


def walk_shapes(layers, shape, k, stride, scope):
    """Yield (param_name, shape) for a layer list applied to an input of ``shape`` (no batch dim);
    returns the output shape through StopIteration.value."""
    for L in layers:
        kind = L[0]
        if kind == "linear":
            yield f"{scope}/{L[1]}/Matrix", (int(np.prod(shape)), L[2])
            yield f"{scope}/{L[1]}/bias", (L[2],)
            shape = (L[2],)
        elif kind == "reshape":
            shape = tuple(L[1])
        elif kind == "flatten":
            shape = (int(np.prod(shape)),)
        elif kind == "conv":
            kk, ss = layer_ks(L, k, stride)
            yield f"{scope}/{L[1]}/w", (kk, kk, shape[-1], L[2])
            yield f"{scope}/{L[1]}/biases", (L[2],)
            shape = (_same_out(shape[0], ss), _same_out(shape[1], ss), L[2])
        elif kind == "deconv":
            kk, ss = layer_ks(L, k, stride)
            yield f"{scope}/{L[1]}/w", (kk, kk, L[2][2], shape[-1])
            yield f"{scope}/{L[1]}/biases", (L[2][2],)
            shape = tuple(L[2])
        elif kind == "bn":
            for v in ("beta", "gamma", "moving_mean", "moving_variance"):
                yield f"{scope}/{L[1]}/{v}", (shape[-1],)
        elif kind == "instnorm":
            yield f"{scope}/{L[1]}/scale", (shape[-1],)
            yield f"{scope}/{L[1]}/offset", (shape[-1],)
        elif kind == "res":
            inner = yield from walk_shapes(L[1], shape, k, stride, scope)
            assert tuple(inner) == tuple(shape), "a residual block must preserve its input shape"
    return shape


def layer_ks(L, k, stride):
    """(kernel size, stride) of a conv / deconv layer tuple: its own if it carries them, else the arch-wide ones."""
    return (L[3], L[4]) if len(L) >= 5 else (k, stride)


def param_shapes(arch):
    A = ARCHS[arch] if isinstance(arch, str) else arch
    out = {}

    def run(layers, shape, scope):
        gen = walk_shapes(layers, shape, A["k"], A["stride"], scope)
        while True:
            try:
                n, s = next(gen)
                out[n] = s
            except StopIteration as e:
                return e.value
    feat = run(A["g_head"], g_input_shape(A), "generator")
    img = run(A["g_tail"], feat, "generator")
    run(A["d"], img, "discriminator")
    return out


def g_input_shape(A):
    """Shape (no batch dim) of what the generator head consumes: a z vector, or an image for image-to-image nets."""
    return tuple(A["g_in"]) if A.get("g_in") else (A["z_dim"],)


def init_params(arch, device, seed=2019):
    """Fresh parameters with the reference's initialisers (nsgan/ops.py:40,43,52,61,77-79): conv
    truncated-normal(0.02), deconv / linear normal(0.02), biases 0, bn gamma 1 / beta 0 / moving (0,1)."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for name, shp in param_shapes(arch).items():
        leaf = name.rsplit("/", 1)[1]
        if leaf == "w" and "conv" in name:
            t = torch.empty(shp)
            torch.nn.init.trunc_normal_(t, 0.0, 0.02, -0.04, 0.04, generator=g)
        elif leaf in ("w", "Matrix"):
            t = torch.randn(shp, generator=g) * 0.02
        elif leaf in ("gamma", "moving_variance", "scale"):
            t = torch.ones(shp)
        else:
            t = torch.zeros(shp)
        P[name] = t.float().to(device)
    return P


def to_device(params, device):
    """name -> tensor/ndarray  ==>  name -> contiguous fp32 device tensor."""
    return {k: torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).float().contiguous().to(device)
            for k, v in params.items()}


def macs_per_sample(arch):
    """(G-tail MACs, D MACs) per sample: conv / deconv / fc only (SURVEY.md 8d work model)."""
    A = ARCHS[arch] if isinstance(arch, str) else arch

    def walk(layers, shape):
        m = 0
        for L in layers:
            if L[0] == "conv":
                kk, ss = layer_ks(L, A["k"], A["stride"])
                o = (_same_out(shape[0], ss), _same_out(shape[1], ss), L[2])
                m += o[0] * o[1] * o[2] * kk ** 2 * shape[2]
                shape = o
            elif L[0] == "deconv":
                kk, ss = layer_ks(L, A["k"], A["stride"])
                m += shape[0] * shape[1] * shape[2] * kk ** 2 * L[2][2]
                shape = tuple(L[2])
            elif L[0] == "res":
                m += walk(L[1], shape)
            elif L[0] == "linear":
                m += int(np.prod(shape)) * L[2]
                shape = (L[2],)
            elif L[0] == "flatten":
                shape = (int(np.prod(shape)),)
            elif L[0] == "reshape":
                shape = tuple(L[1])
        return m
    return walk(A["g_tail"], A["feature"]), walk(A["d"], A["img"])


def refine_flops_per_sample(arch, steps):
    """2*[(2K+1)*MAC_fwd(Gtail+D) + MAC_fwd(Gtail)]: K+1 forward, K backward-data, 1 final render."""
    gt, d = macs_per_sample(arch)
    return 2.0 * ((2 * steps + 1) * (gt + d) + gt)
