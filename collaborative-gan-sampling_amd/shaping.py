"""Discriminator shaping step: the other caller of the refinement path.

The reference alternates "refine a batch probabilistically" with ONE Adam step of D on
``d_loss = mean BCE(D(real), 1) + mean BCE(D(refined), 0)`` (nsgan/GAN.py:126-131,141-146,270-272; D in training
mode: batch statistics in both passes; Adam beta1 = 0.5, learning rate 1e-5 in run_shaping.sh).  This module runs
that step on the GPU over the same stage tape the refiner uses: forward, backward-data with the activation
gradients folded into the epilogues, plus the weight / bias / gamma / beta gradients (wgrad.hip) and the Adam update,
in place on the parameter tensors the ``RefineEngine`` reads -- so the next refinement sees the shaped D.
"""
import math

import torch

from . import kernels as K
from . import lib as L
from .engine import Tape, _BnTrainLrelu, _Conv, _Linear, _View, link_backward_fusion
from .nets import ARCHS


class DShaper:
    def __init__(self, arch, params, batch_size, device="cuda:0", learning_rate=2e-4, beta1=0.5, beta2=0.999, eps=1e-8):
        self.A = ARCHS[arch] if isinstance(arch, str) else arch
        self.dev = torch.device(device)
        self.B = int(batch_size)
        self.lr, self.b1, self.b2, self.eps, self.t = learning_rate, beta1, beta2, eps, 0
        A = self.A
        with torch.cuda.device(self.dev):
            self.tape = Tape(A["d"], A["img"], params, "discriminator", self.B, A["k"], A["stride"], True, self.dev)
            link_backward_fusion(self.tape.stages)
            f32 = dict(dtype=torch.float32, device=self.dev)
            self.dlogits = torch.empty((self.B,) + tuple(self.tape.out_shape), **f32)
            self.loss_buf = torch.zeros(2, **f32)
            # trainable tensors (the d_vars of nsgan/GAN.py:136) with gradient and Adam slots
            self.slots = []        # (param, grad, m, v)
            for st in self.tape.stages:
                names = ("w", "b") if isinstance(st, (_Conv, _Linear)) else ("gamma", "beta") if isinstance(st, _BnTrainLrelu) else ()
                for n in names:
                    p = getattr(st, n)
                    g = torch.zeros_like(p)
                    setattr(st, "g_" + n, g)
                    self.slots.append((p, g, torch.zeros_like(p), torch.zeros_like(p)))

    # -- one backward pass with parameter gradients ------------------------------------------------------
    def _backward(self, dy, accumulate):
        stages = self.tape.stages
        for idx in range(len(stages) - 1, -1, -1):
            st = stages[idx]
            first = idx == 0
            if isinstance(st, _View):
                dy = st.bwd(dy)
            elif isinstance(st, _BnTrainLrelu):
                dy = st.bwd(dy)                                            # dx in place; statistics stay in the bn workspace
                K.bn_train_param_grads(st.x, st.g_gamma, st.g_beta, accumulate)
            elif isinstance(st, _Linear):
                if st.epi == L.EPI_LRELU and not st.pre_folded:
                    dy = K.lrelu_bwd(dy, st.out, out=dy)
                K.linear_bwd_weight(st.x_in, dy, out=st.g_w, accumulate=accumulate)
                K.bias_grad(dy, out=st.g_b, accumulate=accumulate)
                if not first:
                    dy = K.linear_bwd_data(dy, st.w, out=st.dx)
            elif isinstance(st, _Conv):
                if st.epi == L.EPI_LRELU and not st.pre_folded:
                    dy = K.lrelu_bwd(dy, st.out, out=dy)
                kh, kw = st.w.shape[0], st.w.shape[1]
                K.conv2d_bwd_weight(st.x_in, dy, kh, kw, st.s, st.s, out=st.g_w, accumulate=accumulate)
                K.bias_grad(dy, out=st.g_b, accumulate=accumulate)
                if not first:
                    e, a, aux = st.bwd_epi
                    dy = K.conv2d_bwd_data(dy, st.w, st.in_hw, st.s, st.s, out=st.dx, epilogue=e, ep_a=a, ep_aux=aux)
            else:
                raise NotImplementedError(type(st).__name__)

    def loss_and_grads(self, real, refined):
        """d_loss (device scalar tensor) and the gradients of every D variable (left in the ``g_*`` buffers)."""
        with torch.cuda.device(self.dev):
            for i, (x, target) in enumerate(((real, 1.0), (refined, 0.0))):
                logits = self.tape.forward(x.contiguous())
                n = logits.numel()
                K.bce_logits_grad(logits, target, 1.0 / n, self.dlogits, self.loss_buf[i:i + 1])
                self._backward(self.dlogits, accumulate=(i == 1))
            return self.loss_buf.sum()

    def step(self, real, refined):
        """One Adam step of D (nsgan/GAN.py:272).  Returns d_loss before the update."""
        loss = self.loss_and_grads(real, refined)
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)      # tf.train.AdamOptimizer
        with torch.cuda.device(self.dev):
            for p, g, m, v in self.slots:
                K.adam_step(p, g, m, v, lr_t, self.b1, self.b2, self.eps)
        K.WS.invalidate()                   # packed copies of the old weights are stale (re-packed on next use)
        return loss

    def grads(self):
        return [g for _, g, _, _ in self.slots]


def shape_step(engine, shaper, z, real, steps, rate, indices=None):
    """One iteration of the reference's D-shaping loop (nsgan/GAN.py:266-272):
    ``batch_refine = sess.run(g_refine_proba, {z, inputs}); sess.run([d_optim, d_loss], {inputs: real, G: batch_refine})``.
    ``indices``: the probabilistic step draw; the reference bakes ONE draw into the graph for all batches
    (collaborator.py:54-56) -- pass the same array every call to reproduce that, or None to redraw per call."""
    import numpy as np
    if indices is None:
        indices = np.random.randint(steps + 1, size=engine.B)
    refined = engine.refine_from_z(z, steps, rate, mode="probabilistic", indices=indices)[0]
    loss = shaper.step(real, refined)
    engine.refresh_weights()
    return loss
