"""Refiner: collaborative refinement of a G activation map (reference sampling/collaborator.py:7-88).

Same constructor, ``set_env`` / ``set_constraints`` / ``compute_forward_logits_and_grad`` /
``build_refiner`` and the same readable attributes (``default_logit``, ``optimal_logit``,
``optimal_step``, ``optimal_feature``).  The reference's ``build_refiner`` emits a K-times unrolled
TF graph that a later ``sess.run`` executes; here it executes: it returns the refined images.

Execution paths (both all-HIP, no CPU fallback):
  * engine:  if ``discriminator`` / ``feature_to_data`` are the bound methods of a ``cgs_amd.model.GAN``
             and ``func_loss`` its ``loss_refine``, the whole loop runs as the fused device program
             (``engine.RefineEngine``): state resident in HBM, no host sync, optional hipGraph replay.
  * generic: any callables built from ``cgs_amd.ops``; the loop below differentiates them with
             ``torch.autograd`` (each op's backward-data is a HIP kernel) and applies the fused
             update / select kernels.
"""
import numpy as np
import torch

from .policy import PolicyAdaptive
from .. import kernels as K


class Refiner():
    def __init__(self, rollout_steps, rollout_rate, rollout_method="momentum"):
        self.forward_steps = rollout_steps
        self.optimizer = PolicyAdaptive(rollout_rate, rollout_method)
        self.log = False
        self.vmin = None
        self.vmax = None
        self.use_graph = False          # engine path: capture the K-step program into a hipGraph
        self.indices_batch = None       # last probabilistic draw

    def set_env(self, discriminator, feature_to_data, func_loss):
        self.discriminator = discriminator
        self.feature_to_data = feature_to_data
        self.func_loss = func_loss

    def set_constraints(self, vmin, vmax):
        self.vmin = vmin
        self.vmax = vmax
        print("set_constraints: self.vmin = {:.2f}, self.vmax = {:.2f}".format(self.vmin, self.vmax))

    # -- one forward/backward evaluation (collaborator.py:26-39) ---------------------------------
    def compute_forward_logits_and_grad(self, current_feature, need_grad=True):
        feature = current_feature.detach().requires_grad_(need_grad)
        with torch.set_grad_enabled(need_grad):
            forward_logits = self.discriminator(self.feature_to_data(feature))
            forward_grad = None
            if need_grad:
                forward_loss = self.func_loss(forward_logits)
                forward_grad = torch.autograd.grad(forward_loss.sum(), feature)[0]   # tf.gradients sums ys
        flat = forward_logits.detach().reshape(forward_logits.shape[0], -1).contiguous()
        return K.bce_ones_grad_rowmean(flat)[1], forward_grad                        # per-sample mean logit (:34-37)

    # -- engine detection ------------------------------------------------------------------------
    def _engine_for(self, batch):
        from ..model import GAN
        owner = getattr(self.discriminator, "__self__", None)
        if not isinstance(owner, GAN) or getattr(self.feature_to_data, "__self__", None) is not owner:
            return None
        if getattr(self.discriminator, "__func__", None) is not GAN.discriminator_refine:
            return None
        if getattr(self.feature_to_data, "__func__", None) is not GAN.feature_to_data or self.func_loss is not GAN.loss_refine:
            return None
        if self.optimizer.method not in ("momentum", "sgd"):
            return None
        return owner.engine(batch, use_graph=self.use_graph)

    def build_refiner(self, fake_feature, real_batch, mode='deterministic', indices=None):
        """collaborator.py:41-88.  ``real_batch`` only feeds statistics the reference computes and never
        uses (:44-45); it is accepted and ignored.  ``indices`` (extension) replays a fixed probabilistic
        draw; by default one is drawn per call with np.random.randint(K+1, size=B) (:54-56)."""
        K_steps = self.forward_steps
        B = fake_feature.shape[0]
        if mode == 'probabilistic':
            self.indices_batch = np.asarray(indices) if indices is not None else np.random.randint(K_steps + 1, size=B)
        elif mode != 'deterministic':
            raise NotImplementedError

        eng = self._engine_for(B)
        if eng is not None:
            img, d_l, o_l, o_s, o_f = eng.refine(fake_feature, K_steps, self.optimizer.lambda_, self.optimizer.method,
                                                 mode, self.indices_batch if mode == 'probabilistic' else None,
                                                 self.vmin, self.vmax)
            # the engine returns its own (cached, reused) buffers: hand out copies, so that a second build_refiner -- the
            # reference builds a deterministic and a probabilistic refiner side by side, nsgan/GAN.py:182-183 -- does not
            # overwrite the first one's results in place
            self.default_logit, self.optimal_logit = d_l.clone(), o_l.clone()
            self.optimal_step, self.optimal_feature = o_s.clone(), o_f.clone()
            self.optimizer.reset_moving_average()
            return img.clone()

        # ---- generic path -----------------------------------------------------------------------
        self.current_feature = fake_feature.detach().clone().contiguous()
        self.current_logit, self.forward_grad = self.compute_forward_logits_and_grad(self.current_feature)
        self.default_logit = self.current_logit
        self.optimal_feature = self.current_feature.clone()
        self.optimal_logit = self.current_logit.clone().contiguous()
        self.optimal_step = torch.ones_like(self.optimal_logit)
        forced = None
        if mode == 'probabilistic':
            forced = torch.as_tensor(self.indices_batch, dtype=torch.int32).to(fake_feature.device)

        for i in range(K_steps):
            self.current_feature = self.optimizer.apply_gradient(self.current_feature, self.forward_grad)
            if self.vmin and self.vmax:                       # the reference's truthiness test (:69)
                self.current_feature = K.clip(self.current_feature, self.vmin, self.vmax)
            self.current_logit, self.forward_grad = self.compute_forward_logits_and_grad(
                self.current_feature, need_grad=(i + 1 < K_steps))     # the K-th gradient is dead code in the reference graph
            K.refine_select(self.current_feature, self.current_logit.contiguous(), forced, i,
                            self.optimal_feature, self.optimal_logit, self.optimal_step)

        self.optimizer.reset_moving_average()
        with torch.no_grad():
            return self.feature_to_data(self.optimal_feature)
