"""GAN: the sampler-facing half of the reference's model class (nsgan/GAN.py:59-101,164-183),
written against ``ops`` exactly as nsgan/GAN.py is written against nsgan/ops.py.

Training, evaluation, checkpoint and plotting code of nsgan/GAN.py is out of scope;
what is kept is what the refiner calls: ``discriminator`` (batch-statistics bn), ``generator``,
``input_to_feature`` (G head), ``feature_to_data`` (G tail), ``loss_refine``, and the wiring
``build_refiner()`` of nsgan/GAN.py:171-183.  The layer lists come from ``nets.ARCHS`` so the in-tree
MNIST net and the DCGAN-32/64 nets share one code path.

Two ways to run the refinement:
  * generic  -- the bound methods below are plain callables over ``ops`` (autograd over HIP kernels);
  * engine   -- ``GAN.engine(batch)`` compiles the same layer lists into the fused ``RefineEngine``;
                ``collaborator.Refiner`` picks it automatically when handed this object's methods.
"""
import numpy as np
import torch

from . import ops
from .nets import ARCHS, g_input_shape, layer_ks


class GAN(object):
    model_name = "GAN"

    def __init__(self, arch="mnist", batch_size=64, device="cuda:0", params=None):
        self.arch = arch
        self.A = ARCHS[arch] if isinstance(arch, str) else arch
        self.batch_size = batch_size
        self.z_dim = self.A["z_dim"]
        self.image_dims = list(self.A["img"])
        self.device = torch.device(device)
        ops.set_device(self.device)
        if params is not None:
            ops.set_variables(params, self.device)
        self._engines = {}

    # -- layer-list interpreter over the operator API ------------------------------------------
    def _run(self, layers, net, is_training):
        k0, s0 = self.A["k"], self.A["stride"]
        for L in layers:
            kind = L[0]
            k, s = layer_ks(L, k0, s0) if kind in ("conv", "deconv") else (k0, s0)      # per-layer kernel size / stride
            if kind == "linear":
                net = ops.linear(net, L[2], scope=L[1])
            elif kind == "reshape":
                net = net.reshape([net.shape[0]] + list(L[1]))
            elif kind == "flatten":
                net = net.reshape([net.shape[0], -1])
            elif kind == "conv":
                net = ops.conv2d(net, L[2], k, k, s, s, name=L[1])
            elif kind == "deconv":
                net = ops.deconv2d(net, [net.shape[0]] + list(L[2]), k, k, s, s, name=L[1])
            elif kind == "bn":
                net = ops.bn(net, is_training=is_training, scope=L[1])
            elif kind == "relu":
                net = ops.relu(net)
            elif kind == "lrelu":
                net = ops.lrelu(net)
            elif kind == "tanh":
                net = ops.tanh(net)
            elif kind == "instnorm":
                net = ops.instance_norm(net, scope=L[1])
            elif kind == "res":
                net = ops.add(net, self._run(L[1], net, is_training))
            else:
                raise KeyError(kind)
        return net

    def discriminator(self, x, is_training=True, reuse=False):
        """nsgan/GAN.py:59-70."""
        with ops.variable_scope("discriminator", reuse=reuse):
            return self._run(self.A["d"], x, is_training)

    def generator(self, z, is_training=True, reuse=False):
        """nsgan/GAN.py:72-85."""
        with ops.variable_scope("generator", reuse=reuse):
            return self._run(self.A["g_head"] + self.A["g_tail"], z, is_training)

    def input_to_feature(self, z, is_training=False):
        """nsgan/GAN.py:87-92."""
        with ops.variable_scope("generator", reuse=True):
            return self._run(self.A["g_head"], z, is_training)

    def feature_to_data(self, net, is_training=False):
        """nsgan/GAN.py:94-101."""
        with ops.variable_scope("generator", reuse=True):
            return self._run(self.A["g_tail"], net, is_training)

    # -- what nsgan/GAN.py:171-183 wires into the refiner ---------------------------------------
    def discriminator_refine(self, x):
        """partial(self.discriminator, is_training=True, reuse=True)  (nsgan/GAN.py:175)."""
        return self.discriminator(x, is_training=True, reuse=True)

    @staticmethod
    def loss_refine(logits):
        """nsgan/GAN.py:176-177."""
        return ops.sigmoid_cross_entropy_with_logits_ones(logits)

    def build_variables(self):
        """Create every variable (the reference does this by building the training graph, :117-121)."""
        with torch.no_grad():
            z = torch.zeros((2,) + g_input_shape(self.A), device=self.device)      # a z vector, or a source image (image-to-image nets)
            self.discriminator(self.generator(z, is_training=False, reuse=False), is_training=False, reuse=False)
        return ops.variables()

    @staticmethod
    def _engine_key(batch_size, use_graph, contraction="f32", bn_groups=1):
        """The one place the engine cache's key is written."""
        return (int(batch_size), bool(use_graph), contraction, int(bn_groups))

    def engine(self, batch_size=None, use_graph=False, contraction="f32", bn_groups=1):
        """The fused device program for this net at a batch size (compiled once, cached).  ``contraction``: "f32" (exact fp32 MFMA,
        the default) or the opt-in "bx6" (include/cgs_hip.h, cgs_set_contraction).  ``bn_groups`` = G: ``batch_size`` holds G
        logical batches of batch_size / G samples back to back -- one launch per layer for all of them, D's batch statistics kept
        per logical batch (nsgan/GAN.py:175 at the reference's own batch size, nsgan/main.py:32), so the result is that of G calls."""
        from .engine import RefineEngine
        B = int(batch_size or self.batch_size)
        key = self._engine_key(B, use_graph, contraction, bn_groups)
        hit = self._engines.get(key)
        if hit is None or hit[1] != ops.generation():       # a checkpoint was loaded since: re-fold the G bn affines, re-pack
            hit = self._engines[key] = (RefineEngine(self.A, self.build_variables(), B, self.device, use_graph=use_graph, contraction=contraction,
                                                     bn_groups=int(bn_groups)), ops.generation())
        return hit[0]

    def drop_engine(self, batch_size, use_graph, contraction="f32", bn_groups=1):
        """Forget a cached engine (its activation buffers are freed with it)."""
        self._engines.pop(self._engine_key(batch_size, use_graph, contraction, bn_groups), None)

    def demote_engine_to_eager(self, batch_size, contraction="f32", bn_groups=1):
        """A hipGraph capture of the cached graph engine was refused: keep the SAME engine (and its buffers), launched kernel by
        kernel from now on, under the eager key -- unless an eager engine of that signature is already cached, which is then the one
        to use (the demoted one is dropped).  Returns the engine to run."""
        gkey = self._engine_key(batch_size, True, contraction, bn_groups)
        ekey = self._engine_key(batch_size, False, contraction, bn_groups)
        hit = self._engines.pop(gkey, None)
        have = self._engines.get(ekey)
        if have is not None and have[1] == ops.generation():
            return have[0]
        if hit is None:
            return self.engine(batch_size, use_graph=False, contraction=contraction, bn_groups=bn_groups)
        hit[0].use_graph = False
        self._engines[ekey] = (hit[0], ops.generation())
        return hit[0]

    def build_refiner(self, rollout_steps, rollout_rate, rollout_method="momentum"):
        """nsgan/GAN.py:179-181."""
        from .sampling.collaborator import Refiner
        refiner = Refiner(rollout_steps=rollout_steps, rollout_rate=rollout_rate, rollout_method=rollout_method)
        refiner.set_env(self.discriminator_refine, self.feature_to_data, self.loss_refine)
        return refiner
