"""The fused refinement loop on one GPU (sampling/collaborator.py:26-88 as an eager device program).

The reference unrolls K copies of (G tail -> D -> d loss/d theta) into a static TF graph.  Here the
layer lists of ``nets`` are compiled ONCE into a tape of stages over libcgs_hip.so kernels:

  * peephole fusion: deconv+bn(infer)+relu and deconv+tanh, conv+lrelu, linear+lrelu become one
    implicit-GEMM launch with a fused epilogue; D's bn(train)+lrelu one fused stage;
  * every activation / gradient buffer is allocated once (single step live, reused across the K
    steps and across batches); refinement state (theta, momentum, best theta/logit/step) stays in
    HBM for the whole call; weights are packed once;
  * no host synchronisation inside the loop, so the whole K-step program can be captured into a
    hipGraph (``use_graph=True``) and replayed per batch -- the launch-bound regime of small batches.

Only forward + backward-DATA exist: the weights are frozen, no weight gradient is ever formed.
"""
import gc

import numpy as np
import torch

from . import kernels as K
from . import lib as L
from .nets import ARCHS, _same_out, g_input_shape, layer_ks


# ----------------------------------------------------------------------------- stages
class _Stage:
    out = None
    bwd_epi = (L.EPI_NONE, None, None)   # (mode, a, aux): activation gradient of the stage BELOW, folded into this stage's bwd-data
    pre_folded = False                   # True: the stage ABOVE already applied this stage's activation gradient
    bwd_nstat = None                     # conv / deconv stages: a K.NormBwdStats -- this stage's backward-data also leaves the column sums of
                                         # the norm stage right below (whose output gradient it produces); link_norm_backward_stats
    bstat = None                         # norm stages: the same object -- backward = finalize + one pass from those sums

    def fwd(self, x):
        raise NotImplementedError

    def bwd(self, dy):
        raise NotImplementedError


class _View(_Stage):
    def __init__(self, in_shape, out_shape):
        self.in_shape, self.out_shape = in_shape, out_shape

    def fwd(self, x):
        return x.view(self.out_shape)

    def bwd(self, dy):
        return dy.view(self.in_shape)


class _Conv(_Stage):
    """conv2d 'SAME' + bias [+ lrelu]   (nsgan/ops.py:37-46, 69-70)."""

    def __init__(self, B, in_shape, w, b, stride, epi, dev):
        self.w, self.b, self.s, self.epi = w, b, stride, epi
        H, W, _ = in_shape
        self.in_hw = (H, W)
        self.out = torch.empty((B, _same_out(H, stride), _same_out(W, stride), w.shape[3]), dtype=torch.float32, device=dev)
        self.dx = torch.empty((B,) + tuple(in_shape), dtype=torch.float32, device=dev)

    part = None          # [G, 2, Cout]: per-block column sums of the output for the norm that follows (fused statistics)
    part_layout = None   # (rows, rows_per_seg, nseg, seg_stride) of the groups the norm asked for (kernels.conv_stat_layout)

    def want_stats(self, B, group_images=None):
        """Called by the norm stage right above: have this conv leave the statistics partials, if the library can.
        ``group_images``: the norm keeps one set of statistics per that many consecutive images (instance norm: 1; the batch
        norm of fused logical batches: one batch); None = the whole batch.  Returns the partials buffer or None."""
        import os
        if os.environ.get("CGS_NO_FUSED_BN_STATS") or self.epi != L.EPI_NONE:       # (the first: A/B switch for measurements)
            return None
        H, W, Cin = self.dx.shape[1:]
        kh, kw, _, Cout = self.w.shape
        if group_images is None:             # whole-batch statistics: any row order of the launch will do
            G = K.conv_stat_partials((B, H, W, Cin), tuple(self.w.shape), self.s, self.s)
            lay = (G, G, 1, 0) if G > 0 else None
        else:
            lay = K.conv_stat_layout(L.CONV_FWD, B, H, W, Cin, 0, 0, Cout, kh, kw, self.s, self.s, group_images)
        if lay is not None:
            self.part = torch.empty((lay[0], 2, Cout), dtype=torch.float32, device=self.out.device)
            self.part_layout = lay
        return self.part

    def fwd(self, x):
        self.x_in = x
        if self.part is not None:
            return K.conv2d_fwd_stats(x, self.w, self.b, self.part, self.s, self.s, out=self.out)
        return K.conv2d_fwd(x, self.w, self.b, self.s, self.s, self.epi, out=self.out)

    def bwd(self, dy):
        if not self.pre_folded:
            if self.epi == L.EPI_LRELU:
                dy = K.lrelu_bwd(dy, self.out, out=dy)
            elif self.epi == L.EPI_TANH:
                dy = K.tanh_bwd(dy, self.out, out=dy)
        e, a, aux = self.bwd_epi
        return K.conv2d_bwd_data(dy, self.w, self.in_hw, self.s, self.s, out=self.dx, epilogue=e, ep_a=a, ep_aux=aux, nstat=self.bwd_nstat)


class _Deconv(_Stage):
    """conv2d_transpose + bias [+ folded inference bn + relu | + tanh]   (nsgan/ops.py:48-67, nsgan/GAN.py:96-100)."""

    def __init__(self, B, in_shape, out_shape, w, b, stride, epi, a, c, dev):
        self.w, self.b, self.s, self.epi, self.a, self.c = w, b, stride, epi, a, c
        self.in_hw, self.out_hw = (in_shape[0], in_shape[1]), (out_shape[0], out_shape[1])
        self.out = torch.empty((B,) + tuple(out_shape), dtype=torch.float32, device=dev)
        self.dx = torch.empty((B,) + tuple(in_shape), dtype=torch.float32, device=dev)

    signs = None        # int32 [B*Ho*Wo*C/32]: this stage's forward leaves the sign mask of its (relu'd) output there ...
    bwd_signs = None    # ... and this stage's backward-data reads the mask of the stage below instead of its fp32 output
    part = None         # statistics partials of the output for the (instance) norm that follows, as in _Conv
    part_layout = None

    def want_stats(self, B, group_images=None):
        import os
        if os.environ.get("CGS_NO_FUSED_BN_STATS") or self.epi != L.EPI_NONE:
            return None
        lay = K.conv_stat_layout(L.DECONV_FWD, *self.call_dims(), self.s, self.s, group_images or B)
        if lay is not None:
            self.part = torch.empty((lay[0], 2, self.w.shape[2]), dtype=torch.float32, device=self.out.device)
            self.part_layout = lay
        return self.part

    def fwd(self, x):
        return K.deconv2d_fwd(x, self.w, self.b, self.out_hw, self.s, self.s, self.epi, self.a, self.c, out=self.out, signs=self.signs,
                              part=self.part)

    def bwd(self, dy):
        if not self.pre_folded:
            if self.epi == L.EPI_AFFINE_RELU:
                dy = K.affine_relu_bwd(dy, self.out, self.a, out=dy)
            elif self.epi == L.EPI_TANH:
                dy = K.tanh_bwd(dy, self.out, out=dy)
        e, a, aux = self.bwd_epi
        return K.deconv2d_bwd_data(dy, self.w, self.in_hw, self.s, self.s, out=self.dx, epilogue=e, ep_a=a, ep_aux=aux,
                                   ep_signs=self.bwd_signs, nstat=self.bwd_nstat)

    def call_dims(self):
        """(B, H, W, Cin, Ho, Wo, Cout, kh, kw) of the forward call."""
        kh, kw, Cout, Cin = self.w.shape
        return (self.out.shape[0], self.in_hw[0], self.in_hw[1], Cin, self.out_hw[0], self.out_hw[1], Cout, kh, kw)


class _Linear(_Stage):
    def __init__(self, B, w, b, epi, dev):
        self.w, self.b, self.epi = w, b, epi
        self.out = torch.empty((B, w.shape[1]), dtype=torch.float32, device=dev)
        self.dx = torch.empty((B, w.shape[0]), dtype=torch.float32, device=dev)

    def fwd(self, x):
        self.x_in = x
        return K.linear_fwd(x, self.w, self.b, self.epi, out=self.out)

    def bwd(self, dy):
        if self.epi == L.EPI_LRELU:
            dy = K.lrelu_bwd(dy, self.out, out=dy)
        return K.linear_bwd_data(dy, self.w, out=self.dx)


class _BnTrainLrelu(_Stage):
    """Batch-statistics bn [+ lrelu]: D inside the differentiated path (nsgan/GAN.py:65,67,175)."""

    def __init__(self, B, shape, gamma, beta, leak, dev):
        self.gamma, self.beta, self.leak = gamma, beta, leak
        self.out = torch.empty((B,) + tuple(shape), dtype=torch.float32, device=dev)
        self.mean = torch.empty(shape[-1], dtype=torch.float32, device=dev)
        self.invstd = torch.empty(shape[-1], dtype=torch.float32, device=dev)
        self.x = None

    sync = None          # a BnSync: the batch is one shard of a logical batch spread over the ranks of a process group
    groups = 1           # > 1: the engine batch is ``groups`` logical batches back to back, each with its OWN statistics
    part = None          # the producing conv's statistics partials (fused: no statistics pass over the tensor here)

    part_layout = None   # groups > 1: where a group's partial rows lie (the producing conv's part_layout)

    def set_groups(self, groups, producer=None):
        """Statistics per group of rows: the instance-norm kernels with one 'sample' = one logical batch.  ``producer``: the
        conv stage right below, asked to leave its statistics partials per logical batch."""
        self.groups = int(groups)
        self.part = self.part_layout = None
        if producer is not None:
            producer.part = producer.part_layout = None
            B = self.out.shape[0]
            self.part = producer.want_stats(B, B // self.groups)
            self.part_layout = producer.part_layout
        C = self.out.shape[-1]
        self.mean = torch.empty((self.groups, C), dtype=torch.float32, device=self.out.device)
        self.invstd = torch.empty((self.groups, C), dtype=torch.float32, device=self.out.device)

    def _g(self, t):
        return t.view(self.groups, -1, t.shape[-1])

    def fwd(self, x):
        self.x = x
        if self.bstat is not None:
            self.bstat.x = x
        if self.groups > 1:
            if self.part is not None:
                K.groupnorm_lrelu_fwd_from_partials(x, self.part, self.part_layout, self.groups, self.gamma, self.beta, self.leak,
                                                    out=self.out, stats=(self.mean, self.invstd))
            else:
                K.instnorm_lrelu_fwd(self._g(x), self.gamma, self.beta, self.leak, out=self._g(self.out), stats=(self.mean, self.invstd))
            return self.out
        if self.sync is None:
            if self.part is not None:
                K.bn_train_lrelu_fwd_from_partials(x, self.part, self.gamma, self.beta, self.leak, out=self.out, stats=(self.mean, self.invstd))
            else:
                K.bn_train_lrelu_fwd(x, self.gamma, self.beta, self.leak, out=self.out, stats=(self.mean, self.invstd))
            return self.out
        M = x.numel() // x.shape[-1]
        sums = self.sync.sums(x.shape[-1], x.device)
        K.bn_sync_fwd_sums(x, sums)
        self.sync.all_reduce(sums)
        K.bn_sync_fwd_apply(x, self.gamma, self.beta, sums, self.sync.world * M, self.leak, out=self.out,
                            stats=(self.mean, self.invstd))
        return self.out

    def bwd(self, dy):
        if self.bstat is not None:           # the backward-data launch above left the two column sums: finalize + one pass
            return K.norm_lrelu_bwd_from_partials(dy, self.x, self.bstat, self.groups, out=dy)
        if self.groups > 1:
            K.instnorm_lrelu_bwd_data(self._g(dy), self._g(self.x), self.gamma, self.beta, self.mean, self.invstd, self.leak,
                                      out=self._g(dy))
            return dy
        if self.sync is None:
            return K.bn_train_lrelu_bwd_data(dy, self.x, self.gamma, self.beta, self.mean, self.invstd, self.leak, out=dy)
        M = self.x.numel() // self.x.shape[-1]
        sums = self.sync.sums(self.x.shape[-1], dy.device)
        K.bn_sync_bwd_sums(dy, self.x, self.gamma, self.beta, self.mean, self.invstd, sums, self.leak)
        self.sync.all_reduce(sums)
        return K.bn_sync_bwd_apply(dy, self.x, self.gamma, self.beta, self.mean, self.invstd, sums, self.sync.world * M,
                                   self.leak, out=dy)


class BnSync:
    """All-reduce of the per-channel batch-norm sums over a process group (one logical batch of W*B samples
    split W ways reproduces the reference's whole-batch statistics, nsgan/GAN.py:175).  2*C doubles per bn layer and pass.
    "nccl" (RCCL) reduces the device buffer in place; any other backend (gloo in the tests) goes through the host."""

    def __init__(self, group=None):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise L.CgsError("sync_bn needs an initialised torch.distributed process group")
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.on_device = dist.get_backend(group) == "nccl"
        self._sums = {}

    def sums(self, C, dev):
        key = (C, dev)
        if key not in self._sums:
            self._sums[key] = torch.empty((2, C), dtype=torch.float64, device=dev)
        return self._sums[key]

    def all_reduce(self, sums):
        if self.world == 1:
            return
        if self.on_device:
            self.dist.all_reduce(sums, group=self.group)
        else:
            h = sums.cpu()
            self.dist.all_reduce(h, group=self.group)
            sums.copy_(h)


class _InstNormAct(_Stage):
    """Instance norm [+ relu / lrelu]: per-(sample, channel) statistics over the pixels (CycleGAN G, PatchGAN D)."""

    def __init__(self, B, shape, scale, offset, leak, dev):
        self.scale, self.offset, self.leak = scale, offset, leak
        self.out = torch.empty((B,) + tuple(shape), dtype=torch.float32, device=dev)
        self.mean = torch.empty((B, shape[-1]), dtype=torch.float32, device=dev)
        self.invstd = torch.empty((B, shape[-1]), dtype=torch.float32, device=dev)
        self.x = None

    part = None          # the producing conv / deconv's statistics partials (fused: no statistics pass over the tensor here)
    part_layout = None

    def fwd(self, x):
        self.x = x
        if self.bstat is not None:
            self.bstat.x = x
        if self.part is not None:
            K.groupnorm_lrelu_fwd_from_partials(x, self.part, self.part_layout, x.shape[0], self.scale, self.offset, self.leak,
                                                out=self.out, stats=(self.mean, self.invstd))
        else:
            K.instnorm_lrelu_fwd(x, self.scale, self.offset, self.leak, out=self.out, stats=(self.mean, self.invstd))
        return self.out

    def bwd(self, dy):
        if self.bstat is not None:
            return K.norm_lrelu_bwd_from_partials(dy, self.x, self.bstat, dy.shape[0], out=dy)
        return K.instnorm_lrelu_bwd_data(dy, self.x, self.scale, self.offset, self.mean, self.invstd, self.leak, out=dy)


class _Residual(_Stage):
    """x + F(x): forward adds the skip, backward adds the two gradient branches."""

    def __init__(self, B, shape, inner, dev):
        self.inner = inner
        self.out = torch.empty((B,) + tuple(shape), dtype=torch.float32, device=dev)
        self.dskip = torch.empty((B,) + tuple(shape), dtype=torch.float32, device=dev)

    def fwd(self, x):
        y = x
        for st in self.inner:
            y = st.fwd(y)
        return K.add(x, y, out=self.out)

    def bwd(self, dy):
        self.dskip.copy_(dy)                     # the inner stages transform their incoming gradient in place
        g = dy
        for st in reversed(self.inner):
            g = st.bwd(g)
        return K.add(self.dskip, g, out=self.dskip)


class _AffineRelu(_Stage):
    """Inference-mode bn + relu that could not be folded into a producer epilogue (G head)."""

    def __init__(self, B, shape, a, c, dev):
        self.a, self.c = a, c
        self.out = torch.empty((B,) + tuple(shape), dtype=torch.float32, device=dev)

    def fwd(self, x):
        return K.affine_relu_fwd(x, self.a, self.c, out=self.out)

    def bwd(self, dy):
        return K.affine_relu_bwd(dy, self.out, self.a, out=dy)


class _Unary(_Stage):
    def __init__(self, B, shape, kind, dev):
        self.kind = kind
        self.out = torch.empty((B,) + tuple(shape), dtype=torch.float32, device=dev)

    def fwd(self, x):
        if self.kind == "lrelu":
            return K.lrelu_fwd(x, out=self.out)
        if self.kind == "tanh":
            return K.tanh_fwd(x, out=self.out)
        return K.lrelu_fwd(x, 0.0, out=self.out)     # relu = lrelu with leak 0

    def bwd(self, dy):
        if self.kind == "lrelu":
            return K.lrelu_bwd(dy, self.out, out=dy)
        if self.kind == "tanh":
            return K.tanh_bwd(dy, self.out, out=dy)
        return K.lrelu_bwd(dy, self.out, 0.0, out=dy)


def compile_layers(layers, in_shape, P, scope, B, k, stride, bn_training, dev, folds=None):
    """Layer list -> stage tape with peephole fusion.  Returns (stages, out_shape).
    ``folds`` collects (a, c, gamma, beta, moving_mean, moving_variance) of every folded inference bn: values DERIVED from
    the parameters, which the engine refreshes in place when the parameters are restored / updated under it."""
    stages, shape, i = [], tuple(in_shape), 0
    n = len(layers)

    def kind(j):
        return layers[j][0] if j < n else None

    def fold(name):
        s = f"{scope}/{name}"
        src = (P[s + "/gamma"], P[s + "/beta"], P[s + "/moving_mean"], P[s + "/moving_variance"])
        a, c = K.bn_fold(*src)
        if folds is not None:
            folds.append((a, c) + src)
        return a, c

    while i < n:
        Lr = layers[i]
        t = Lr[0]
        if t == "deconv":
            w, b = P[f"{scope}/{Lr[1]}/w"], P[f"{scope}/{Lr[1]}/biases"]
            _, stride_l = layer_ks(Lr, k, stride)
            epi, a, c, used = L.EPI_NONE, None, None, 1
            if kind(i + 1) == "bn" and kind(i + 2) == "relu" and not bn_training:
                a, c = fold(layers[i + 1][1]); epi, used = L.EPI_AFFINE_RELU, 3
            elif kind(i + 1) == "tanh":
                epi, used = L.EPI_TANH, 2
            stages.append(_Deconv(B, shape, Lr[2], w, b, stride_l, epi, a, c, dev))
            shape = tuple(Lr[2]); i += used
        elif t == "conv":
            w, b = P[f"{scope}/{Lr[1]}/w"], P[f"{scope}/{Lr[1]}/biases"]
            _, stride_l = layer_ks(Lr, k, stride)
            epi, used = (L.EPI_LRELU, 2) if kind(i + 1) == "lrelu" else (L.EPI_TANH, 2) if kind(i + 1) == "tanh" else (L.EPI_NONE, 1)
            st = _Conv(B, shape, w, b, stride_l, epi, dev)
            stages.append(st)
            shape = tuple(st.out.shape[1:]); i += used
        elif t == "linear":
            w, b = P[f"{scope}/{Lr[1]}/Matrix"], P[f"{scope}/{Lr[1]}/bias"]
            epi, used = (L.EPI_LRELU, 2) if kind(i + 1) == "lrelu" else (L.EPI_NONE, 1)
            stages.append(_Linear(B, w, b, epi, dev))
            shape = (Lr[2],); i += used
        elif t == "bn":
            s = f"{scope}/{Lr[1]}"
            if bn_training:
                leak, used = (K.LEAK, 2) if kind(i + 1) == "lrelu" else (1.0, 1)
                bn_stage = _BnTrainLrelu(B, shape, P[s + "/gamma"], P[s + "/beta"], leak, dev)
                if stages and isinstance(stages[-1], _Conv):      # statistics ride on the producing conv's epilogue
                    bn_stage.part = stages[-1].want_stats(B)
                stages.append(bn_stage)
                i += used
            else:
                if kind(i + 1) != "relu":
                    raise NotImplementedError("inference-mode bn is only supported when followed by relu")
                a, c = fold(Lr[1])
                stages.append(_AffineRelu(B, shape, a, c, dev)); i += 2
        elif t == "instnorm":
            sc = f"{scope}/{Lr[1]}"
            leak, used = (K.LEAK, 2) if kind(i + 1) == "lrelu" else (0.0, 2) if kind(i + 1) == "relu" else (1.0, 1)
            in_stage = _InstNormAct(B, shape, P[sc + "/scale"], P[sc + "/offset"], leak, dev)
            if stages and isinstance(stages[-1], (_Conv, _Deconv)):       # statistics ride on the producing conv's epilogue, per sample
                in_stage.part = stages[-1].want_stats(B, 1)
                in_stage.part_layout = stages[-1].part_layout
            stages.append(in_stage); i += used
        elif t == "res":
            inner, ishape = compile_layers(Lr[1], shape, P, scope, B, k, stride, bn_training, dev, folds)
            assert tuple(ishape) == tuple(shape)
            stages.append(_Residual(B, shape, inner, dev)); i += 1
        elif t in ("relu", "lrelu", "tanh"):
            stages.append(_Unary(B, shape, t, dev)); i += 1
        elif t == "reshape":
            stages.append(_View((B,) + shape, (B,) + tuple(Lr[1]))); shape = tuple(Lr[1]); i += 1
        elif t == "flatten":
            flat = (int(np.prod(shape)),)
            stages.append(_View((B,) + shape, (B,) + flat)); shape = flat; i += 1
        else:
            raise KeyError(t)
    return stages, shape


def link_backward_fusion(stages):
    """For each conv-family stage whose predecessor ends in relu(a*x+b) / lrelu / tanh, fold that activation's
    gradient into this stage's backward-data epilogue (one elementwise pass over the gradient map less)."""
    for below, above in zip(stages[:-1], stages[1:]):
        if not isinstance(above, (_Conv, _Deconv)) or below.pre_folded:
            continue
        if isinstance(below, _Deconv) and below.epi == L.EPI_AFFINE_RELU:
            above.bwd_epi = (L.EPI_RELU_BWD_AFFINE, below.a, below.out)
            # relu' needs the sign of below.out only: where the consumer is the HBM-bound 3-channel backward-data kernel and the
            # producer can leave a bitmask from its epilogue, the 1-bit mask replaces the fp32 read (include/cgs_hip.h, "sign masks")
            import os
            if isinstance(above, _Deconv) and not os.environ.get("CGS_NO_SIGN_MASKS"):      # (A/B switch for measurements)
                s_ = (below.s, below.s)
                if (K.conv_signs_ok(L.DECONV_FWD, *below.call_dims(), *s_, L.EPI_AFFINE_RELU)
                        and K.conv_signs_ok(L.DECONV_BWD_DATA, *above.call_dims(), above.s, above.s, L.EPI_RELU_BWD_AFFINE)):
                    below.signs = torch.empty(below.out.numel() // 32, dtype=torch.int32, device=below.out.device)
                    above.bwd_signs = below.signs
        elif isinstance(below, _Deconv) and below.epi == L.EPI_TANH:
            above.bwd_epi = (L.EPI_TANH_BWD, None, below.out)
        elif isinstance(below, _Conv) and below.epi == L.EPI_LRELU:
            above.bwd_epi = (L.EPI_LRELU_BWD, None, below.out)
        elif isinstance(below, _Conv) and below.epi == L.EPI_TANH:
            above.bwd_epi = (L.EPI_TANH_BWD, None, below.out)
        else:
            continue
        below.pre_folded = True


# Largest norm input (bytes) whose backward sums ride on the backward-data launch above it.  The fusion pays where launches are latency-bound
# (the reference's batch 64, single calls of <= 256 images, config 5's 8-image batches: -0.7 ... -2.8 % per call); on the
# matrix-bound launches of the big batches (dcgan64 at batch 1024: 67 / 134 MB norm inputs) it is a wash -- the sums twin's dearer epilogue and the
# saved pass cancel, -0.2 % +- the noise of a same-process A/B (profiles/r06_m_*, r06_o_*) -- and those launches would merely move from the
# headline's dominant kernel to its twin's name in every kernel table.  (CGS_NSTAT_MAX_MB overrides it for A/B measurements: 0 = never.)
NSTAT_MAX_BYTES = 48 * 2 ** 20


def link_norm_backward_stats(stages, B):
    """For each norm stage (batch statistics, per logical batch or per sample) whose successor is a conv / deconv stage: that stage's
    backward-data launch produces the gradient at the norm's output, so it can leave the norm backward's two column sums in its
    epilogue (include/cgs_hip.h, cgs_*_bwd_data_nstats) -- the norm's own sums pass over dy and x, 2 of its 5 tensor passes, is gone.
    Only where the library offers it for the call (exact-fp32 implicit GEMM, groups that end on 64-row boundaries of the launch's row
    order); everything else keeps the three-kernel backward.  Call after the norms' groups are final (set_groups re-allocates the
    saved statistics)."""
    import os
    if os.environ.get("CGS_NO_FUSED_BN_BWD_STATS"):           # (A/B switch for measurements)
        return
    for below, above in zip(stages[:-1], stages[1:]):
        if isinstance(below, _Residual):
            link_norm_backward_stats(below.inner, B)
        if not isinstance(above, (_Conv, _Deconv)) or above.bwd_epi[0] != L.EPI_NONE or above.bwd_nstat is not None:
            continue
        if isinstance(below, _BnTrainLrelu) and below.sync is None:
            gamma, beta, gimg = below.gamma, below.beta, B // below.groups
        elif isinstance(below, _InstNormAct):
            gamma, beta, gimg = below.scale, below.offset, 1
        else:
            continue
        C = below.out.shape[-1]
        limit = float(os.environ["CGS_NSTAT_MAX_MB"]) * 2 ** 20 if os.environ.get("CGS_NSTAT_MAX_MB") else NSTAT_MAX_BYTES
        if limit is not None and below.out.numel() * 4 > limit:
            continue
        if isinstance(above, _Conv):
            H, W, Cin = above.dx.shape[1:]
            kh, kw, _, Cout = above.w.shape
            lay = K.conv_stat_layout(L.CONV_BWD_DATA, B, H, W, Cin, 0, 0, Cout, kh, kw, above.s, above.s, gimg)
        else:
            lay = K.conv_stat_layout(L.DECONV_BWD_DATA, *above.call_dims(), above.s, above.s, gimg)
        if lay is None:
            continue
        part = torch.empty((lay[0], 2, C), dtype=torch.float32, device=below.out.device)
        below.bstat = above.bwd_nstat = K.NormBwdStats(None, below.mean, below.invstd, gamma, beta, below.leak, gimg, part, lay)
    if stages and isinstance(stages[-1], _Residual):
        link_norm_backward_stats(stages[-1].inner, B)


class Tape:
    """A compiled layer list: forward keeps what backward-data needs; no weight gradients."""

    def __init__(self, layers, in_shape, P, scope, B, k, stride, bn_training, dev):
        self.folds = []
        self.stages, self.out_shape = compile_layers(layers, in_shape, P, scope, B, k, stride, bn_training, dev, self.folds)

    def forward(self, x):
        for st in self.stages:
            x = st.fwd(x)
        return x

    def backward(self, dy):
        for st in reversed(self.stages):
            dy = st.bwd(dy)
        return dy


# ----------------------------------------------------------------------------- the loop
def _entry(fn):
    """A public entry point of an engine: the engine's contraction mode is in force for the call and the calling thread's previous
    mode is restored after it (ops.* calls, a DShaper step or another engine do not inherit this engine's opt-in)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        prev = K.set_contraction(self.contraction)
        try:
            return fn(self, *a, **kw)
        finally:
            if prev != self.contraction:
                K.set_contraction(prev)
    return wrapped


class RefineEngine:
    """K-step collaborative refinement of a batch of G activation maps on one GPU."""

    def __init__(self, arch, params, batch_size, device=None, use_graph=False, sync_bn=None, bn_groups=1, contraction="f32"):
        self.A = ARCHS[arch] if isinstance(arch, str) else arch
        # "f32": every contraction on the exact-fp32 matrix instructions (the reference's precision; default).  "bx6": opt-in, the
        # layers with a multiple of 64 output channels, Cred % 32 == 0 and GPU-filling grids through split-bf16 MFMA (include/cgs_hip.h, cgs_set_contraction;
        # csrc/igemm_bx6.hip) -- same results to fp32 rounding, less matrix time.  The mode is a property of the ENGINE: it is put in
        # force at construction (the layer compilation asks the library which kernel families it will get) and at every entry point.
        if contraction not in L.CONTRACTIONS:
            raise L.CgsError(f"contraction {contraction!r}: expected one of {sorted(L.CONTRACTIONS)}")
        self.contraction = contraction
        self.dev = torch.device(device if device is not None else "cuda:0")
        if self.dev.type != "cuda":
            raise L.CgsError("RefineEngine needs a GPU device (there is no CPU path)")
        L.load()
        prev_mode = K.set_contraction(self.contraction)
        try:
            self._build(params, batch_size, use_graph, sync_bn, bn_groups)
        finally:
            if prev_mode != self.contraction:
                K.set_contraction(prev_mode)

    def _build(self, params, batch_size, use_graph, sync_bn, bn_groups):
        A, B = self.A, int(batch_size)
        self.B, self.P = B, params
        with torch.cuda.device(self.dev):
            self.g_head = Tape(A["g_head"], g_input_shape(A), params, "generator", B, A["k"], A["stride"], False, self.dev)
            self.g_tail = Tape(A["g_tail"], A["feature"], params, "generator", B, A["k"], A["stride"], False, self.dev)
            self.d = Tape(A["d"], A["img"], params, "discriminator", B, A["k"], A["stride"], True, self.dev)
            link_backward_fusion(self.g_tail.stages + self.d.stages)     # across the G-tail / D seam too
            fs = (B,) + tuple(A["feature"])
            f32 = dict(dtype=torch.float32, device=self.dev)
            self.theta, self.mom, self.best_theta = torch.empty(fs, **f32), torch.empty(fs, **f32), torch.empty(fs, **f32)
            self.logit, self.best_logit = torch.empty(B, **f32), torch.empty(B, **f32)
            self.default_logit, self.best_step = torch.empty(B, **f32), torch.empty(B, **f32)
            self.tickets = torch.zeros(B, dtype=torch.int32, device=self.dev)      # cgs_refine_select2: zero between launches
            self.dlogits = torch.empty((B,) + tuple(self.d.out_shape), **f32)
            self.forced = torch.zeros(B, dtype=torch.int32, device=self.dev)
            self.images = torch.empty((B,) + tuple(A["img"]), **f32)
        self.use_graph = use_graph
        self._graphs = {}
        self._gstream = None                 # the side stream hipGraphs are warmed up and captured on
        self._ws_epoch = K.WS.epoch          # the parameter state the folded affines / graph workspaces were derived from
        # sync_bn = True (default process group) or a process group: this engine's batch is one shard of a logical batch of
        # world_size * batch_size samples; D's batch statistics are all-reduced so the result equals the unsplit batch's
        # bn_groups = G: ``batch_size`` holds G logical batches of batch_size / G samples back to back.  Convolutions do not
        # care; D's batch norm keeps one set of statistics per logical batch, so the result is that of G separate calls while
        # every launch is G times larger (small batches then fill the GPU from one stream).
        self.bn_groups = int(bn_groups)
        if self.bn_groups > 1:
            if B % self.bn_groups:
                raise L.CgsError(f"bn_groups={bn_groups} does not divide the engine batch {B}")
            if sync_bn:
                raise L.CgsError("bn_groups and sync_bn are mutually exclusive")

            def walk_g(stages):
                for below, st in zip([None] + stages[:-1], stages):
                    if isinstance(st, _BnTrainLrelu):
                        st.set_groups(self.bn_groups, below if isinstance(below, _Conv) else None)
                    if isinstance(st, _Residual):
                        walk_g(st.inner)
            walk_g(self.d.stages); walk_g(self.g_tail.stages)
        self.sync_bn = None
        if sync_bn is not None and sync_bn is not False:
            if use_graph:
                raise L.CgsError("sync_bn runs eagerly: the all-reduce between the two halves of a bn pass is not captured")
            self.sync_bn = BnSync(None if sync_bn is True else sync_bn)

            def walk(stages):
                for st in stages:
                    if isinstance(st, _BnTrainLrelu):
                        st.sync = self.sync_bn
                    if isinstance(st, _Residual):
                        walk(st.inner)
            walk(self.d.stages); walk(self.g_tail.stages)

        # synchronised statistics do not use the fused partials: switch them off in the producing convs too
        def drop_partials(stages):
            for below, st in zip([None] + stages[:-1], stages):
                if isinstance(st, _BnTrainLrelu) and st.sync is not None:
                    st.part = None
                    if isinstance(below, _Conv):
                        below.part = None
                if isinstance(st, _Residual):
                    drop_partials(st.inner)
        drop_partials(self.d.stages); drop_partials(self.g_tail.stages)
        # the norm backward's column sums ride on the backward-data launch of the conv above (across the G-tail / D seam too)
        link_norm_backward_stats(self.g_tail.stages + self.d.stages, B)

    # -- parameters changed under the engine ---------------------------------------------------
    def _sync_weights(self):
        """Every entry point starts here: one integer compare.  The parameters are read by pointer (raw gamma / beta / bias /
        the N == 1 linear), through packed copies (conv / deconv / wide linear weights, cached in ``K.WS``) and through folded
        inference-bn affines; after ``ops.set_variables`` / a shaping step (``K.WS.invalidate()``) the last two are stale, and a
        captured hipGraph would replay a MIX of old and new state.  Any engine -- also one built directly, not through
        ``model.GAN`` -- re-derives them here before its next use."""
        if self._ws_epoch != K.WS.epoch:
            self._resync()

    def _resync(self):
        # the eager forward + backward below re-packs the workspaces of THIS engine's kernel families (the cache is keyed by family):
        # the engine's own contraction must be in force, whichever engine of the process ran last
        K.set_contraction(self.contraction)
        epoch = K.WS.epoch
        with torch.cuda.device(self.dev):
            for tape in (self.g_head, self.g_tail, self.d):
                for a, c, gamma, beta, mm, mv in tape.folds:
                    K.bn_fold(gamma, beta, mm, mv, out=(a, c))
            if self._gstream is not None and self._graphs:
                # the captured graphs read the workspaces packed on the capture stream: one eager forward + backward there
                # re-packs every layer into those SAME buffers (eager calls on other streams re-pack lazily by themselves)
                cur = torch.cuda.current_stream(self.dev)
                self._gstream.wait_stream(cur)
                with torch.cuda.stream(self._gstream):
                    self.forward_logits(self.theta, self.logit)
                    self.backward_to_feature()
                cur.wait_stream(self._gstream)
        self._ws_epoch = epoch

    # -- pieces (sampling/collaborator.py:26-39) ------------------------------------------------
    @_entry
    def input_to_feature(self, z):
        """G head (nsgan/GAN.py:87-92)."""
        self._sync_weights()
        return self.g_head.forward(z)

    @_entry
    def feature_to_data(self, feat):
        """G tail (nsgan/GAN.py:94-101)."""
        self._sync_weights()
        return self.g_tail.forward(feat)

    @_entry
    def discriminator(self, x):
        """D with batch-statistics bn (nsgan/GAN.py:59-70 bound at :175)."""
        self._sync_weights()
        return self.d.forward(x)

    @_entry
    def generate(self, z):
        """fake_images = generator(z, is_training=False): G head + G tail, no refinement (nsgan/GAN.py:153).  Engine-owned buffer."""
        self._sync_weights()
        return self.g_tail.forward(self.g_head.forward(z))

    @_entry
    def score(self, images, out=None):
        """fake_sigmoids = sigmoid(D(images)) with D on batch statistics -- per logical batch under ``bn_groups`` -- as [B, 1]
        (nsgan/GAN.py:154-155): what Rejector / IndependenceSampler read.  Stays on the device."""
        self._sync_weights()
        return K.sigmoid_rowmean(self.d.forward(images), out=out)

    def forward_logits(self, theta, logit_out):
        x = self.g_tail.forward(theta)
        head = self.d.stages[-1]
        if isinstance(head, _Linear) and head.w.shape[1] == 1 and head.epi == L.EPI_NONE:
            # D ends in the one-logit linear head (nsgan/GAN.py:68): head + loss seed + per-sample mean logit in ONE launch
            for st in self.d.stages[:-1]:
                x = st.fwd(x)
            head.x_in = x
            return K.linear_out1_bce(x, head.w, head.b, head.out, self.dlogits, logit_out)
        logits = self.d.forward(x)
        K.bce_ones_grad_rowmean(logits, self.dlogits, logit_out)
        return logits

    def backward_to_feature(self):
        """d sum_b softplus(-logit_b) / d theta, through D then the G tail (collaborator.py:31)."""
        return self.g_tail.backward(self.d.backward(self.dlogits))

    @_entry
    def compute_forward_logits_and_grad(self, feature):
        self._sync_weights()
        self.forward_logits(feature, self.logit)
        return self.logit, self.backward_to_feature()

    # -- the K-step program -------------------------------------------------------------------
    def _program(self, steps, rate, alpha, probabilistic, vmin, vmax):
        th = self.theta
        render = self.g_tail.stages[-1].out                         # x = G_tail(theta) of the current step
        self.forward_logits(th, self.logit)
        self.default_logit.copy_(self.logit)
        self.best_logit.copy_(self.logit)
        self.best_theta.copy_(th)
        self.images.copy_(render)
        self.best_step.fill_(1.0)                                   # collaborator.py:60 (starts at 1)
        self.tickets.zero_()                                        # (already zero unless an earlier call died between its launches)
        forced = self.forced if probabilistic else None
        for i in range(steps):
            g = self.backward_to_feature()
            K.refine_update(th, self.mom, g, rate, alpha, first=(i == 0), vmin=vmin, vmax=vmax)
            self.forward_logits(th, self.logit)                     # the K-th gradient is never formed (Q4)
            # collaborator.py:88 renders G_tail(best_theta) once more at the end; the very same image was already
            # rendered in the step that selected it, so it is kept by the same row-select instead (bit-identical)
            K.refine_select2(render, self.images, th, self.best_theta, self.logit, forced, i, self.best_logit, self.best_step, self.tickets)

    @_entry
    def refine(self, feature0, steps, rate, method="momentum", mode="deterministic", indices=None,
               vmin=None, vmax=None):
        """collaborator.py:41-88 on device.  Returns (images, default_logit, optimal_logit, optimal_step,
        optimal_feature) -- engine-owned buffers, valid until the next call."""
        if method == "momentum":
            alpha = 0.9                                             # policy.py:10
        elif method == "sgd":
            alpha = 0.0
        elif method == "ladam":
            raise L.CgsError("ladam needs a loss argument the map-space refiner never passes "
                             "(sampling/collaborator.py:66 vs policy.py:48-51); use momentum or sgd")
        else:
            raise NotImplementedError(method)
        if mode not in ("deterministic", "probabilistic"):
            raise NotImplementedError(mode)
        prob = mode == "probabilistic"
        if prob:
            if indices is None:
                raise L.CgsError("probabilistic mode needs the step indices (np.random.randint(K+1, size=B))")
            self.forced.copy_(torch.as_tensor(np.asarray(indices), dtype=torch.int32))
        if tuple(feature0.shape) != tuple(self.theta.shape):
            raise L.CgsError(f"feature batch {tuple(feature0.shape)} != engine shape {tuple(self.theta.shape)}")
        self._sync_weights()
        with torch.cuda.device(self.dev):
            self.theta.copy_(feature0)
            key = (steps, float(rate), alpha, prob, vmin, vmax)
            if not self.use_graph:
                self._program(steps, rate, alpha, prob, vmin, vmax)
            else:
                g = self._graphs.get(key)
                if g is None:
                    # Warm-up and capture run on ONE explicit side stream: the packed-weight and bn workspaces are keyed by
                    # the stream, so the warm-up packs the very buffers the capture then finds (ws_prepacked = 1 inside the
                    # graph: no pack kernels are recorded, a replay does not re-pack ~100 MB of weights).
                    if self._gstream is None:
                        self._gstream = torch.cuda.Stream(self.dev)
                    cur = torch.cuda.current_stream(self.dev)
                    self._gstream.wait_stream(cur)
                    with torch.cuda.stream(self._gstream):
                        self._program(steps, rate, alpha, prob, vmin, vmax)
                        self.theta.copy_(feature0)
                    torch.cuda.synchronize(self.dev)
                    # (thread-local capture: other threads of the process -- RCCL proxies, the distributed watchdog, a second engine's
                    # host thread -- may make HIP calls meanwhile without invalidating this capture)
                    # (no cyclic garbage collection inside the capture: a finalizer that reaches HIP -- a dead engine's graph, stream or event
                    # waiting in a reference cycle -- makes a call the capturing thread may not make, and the runtime ABORTS the process instead
                    # of refusing the capture; seen once the GPU suite ran in one process: the 1,460 launches of a K=50 program allocate
                    # enough small objects to trigger a collection.  torch.cuda.graph collects once itself BEFORE the capture begins.)
                    gc_was_on = gc.isenabled()
                    gc.disable()
                    try:
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=self._gstream, capture_error_mode="thread_local"):
                            self._program(steps, rate, alpha, prob, vmin, vmax)
                    except L.CgsError:
                        raise
                    except Exception as ex:                          # noqa: BLE001 (HIP / allocator / another thread's HIP call refused the capture)
                        raise L.GraphCaptureError(f"{type(ex).__name__}: {str(ex)[:300]}") from ex
                    finally:
                        if gc_was_on:
                            gc.enable()
                    cur.wait_stream(self._gstream)
                    self._graphs[key] = g
                g.replay()
        return self.images, self.default_logit, self.best_logit, self.best_step, self.best_theta

    @_entry
    def refresh_weights(self):
        """Call after the parameter tensors were updated in place by a kernel torch does not see (``shaping.DShaper.step``):
        marks every packed copy stale (for ALL engines: each one re-derives its state at its next call, ``_sync_weights``)
        and brings this engine up to date now -- folded affines, and the workspaces its captured hipGraphs read."""
        K.WS.invalidate()
        self._resync()

    @_entry
    def refine_from_z(self, z, steps, rate, **kw):
        """Propose (G head) + refine + render: one whole unit of the BASELINE metric."""
        return self.refine(self.input_to_feature(z), steps, rate, **kw)
