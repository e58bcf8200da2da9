"""The 2-D (synthetic/) side of the reference on the GPU: BASELINE config 1.

* ``MLPDiscriminator``  -- the ReLU MLP D of synthetic/GAN.py:28-37 with the three tensors refiner_cpu fetches
  (``fake_samples`` / ``fake_sigmoid`` / ``fake_saliency``, synthetic/GAN.py:105-111) evaluated by one HIP kernel.
* ``Session`` / ``gan`` -- a ``sess.run(fetches, feed_dict)`` adaptor so the reference-shaped host loop
  ``sampling.refiner_cpu.Refiner`` drives the GPU discriminator unchanged (K+2 launches per batch).
* ``Refiner``           -- same class surface as ``refiner_cpu.Refiner`` (``Refiner(args)``, ``set_env``,
  ``manipulate_sample``), but the whole K-step loop -- D forward, saliency, ladam / momentum / sgd update, best-loss
  tracking, trajectory -- is ONE kernel launch (one wave per sample, weights in LDS).

Variable names follow tf.layers.dense: ``discriminator/d_fc<i>/kernel`` ([din, dout]) and ``.../bias``.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L

_METHODS = {"sgd": 0, "momentum": 1, "ladam": 2}


class MLPDiscriminator:
    def __init__(self, params, device="cuda:0"):
        """``params``: {"discriminator/d_fc1/kernel": [2,nh], "discriminator/d_fc1/bias": [nh], ...}."""
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise L.CgsError("MLPDiscriminator needs a GPU device (the host loop with a caller-supplied sess is sampling.refiner_cpu)")
        L.load()
        n = 1
        while f"discriminator/d_fc{n + 1}/kernel" in params:
            n += 1
        self.nlayers = n
        self.w = [torch.as_tensor(np.asarray(params[f"discriminator/d_fc{i + 1}/kernel"]), dtype=torch.float32).contiguous().to(self.dev) for i in range(n)]
        self.b = [torch.as_tensor(np.asarray(params[f"discriminator/d_fc{i + 1}/bias"]), dtype=torch.float32).contiguous().to(self.dev) for i in range(n)]
        self.nhidden = int(self.w[0].shape[1])
        if self.w[0].shape[0] != 2 or self.w[-1].shape[1] != 1 or self.nhidden > 64 or not 2 <= n <= 6:
            # every layer's weights (and their transposes) are LDS-resident: 6 layers = 133 KB of the CU's 160 KB
            raise L.CgsError(f"MLPDiscriminator: unsupported shape (2 -> {self.nhidden} x {n - 1} -> 1; need nhidden <= 64, 2..6 layers)")
        self._wp = (C.c_void_p * n)(*[t.data_ptr() for t in self.w])
        self._bp = (C.c_void_p * n)(*[t.data_ptr() for t in self.b])

    @classmethod
    def from_lists(cls, Ws, bs, device="cuda:0"):
        P = {}
        for i, (w, b) in enumerate(zip(Ws, bs)):
            P[f"discriminator/d_fc{i + 1}/kernel"], P[f"discriminator/d_fc{i + 1}/bias"] = np.asarray(w), np.asarray(b)
        return cls(P, device)

    def _x(self, x):
        return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).to(self.dev) if not isinstance(x, torch.Tensor) else x.float().contiguous().to(self.dev)

    def sigmoid_and_saliency(self, x, want_saliency=True):
        """-> (sigmoid [B,1], saliency [B,2] or None) as device tensors; saliency carries the 1/B factor (quirk Q8)."""
        xd = self._x(x)
        B = xd.shape[0]
        sig = torch.empty(B, dtype=torch.float32, device=self.dev)
        sal = torch.empty((B, 2), dtype=torch.float32, device=self.dev) if want_saliency else None
        L.call("cgs_mlp2d_sigmoid_saliency", self._wp, self._bp, self.nlayers, self.nhidden, xd.data_ptr(), sig.data_ptr(),
               None if sal is None else sal.data_ptr(), B, 1.0 / B, torch.cuda.current_stream(self.dev).cuda_stream)
        return sig.view(B, 1), sal

    def refine(self, fake_batch, real_sigmoid_mean, steps, rate, method="ladam", want_traj=False):
        """The fused K-step loop.  -> (best_x [B,2], best_step [B], traj [B,K+1,2] or None) device tensors."""
        if method not in _METHODS:
            raise NotImplementedError(method)
        xd = self._x(fake_batch)
        B = xd.shape[0]
        best = torch.empty((B, 2), dtype=torch.float32, device=self.dev)
        step = torch.empty(B, dtype=torch.float32, device=self.dev)
        traj = torch.empty((B, steps + 1, 2), dtype=torch.float32, device=self.dev) if want_traj else None
        tail = (1.0 / B, int(steps), float(rate), _METHODS[method], best.data_ptr(), step.data_ptr(),
                None if traj is None else traj.data_ptr(), B, torch.cuda.current_stream(self.dev).cuda_stream)
        if isinstance(real_sigmoid_mean, torch.Tensor):            # a device scalar: no host round trip, batches queue back to back
            base = real_sigmoid_mean.to(self.dev, torch.float32).reshape(1)
            L.call("cgs_refine2d_devbase", self._wp, self._bp, self.nlayers, self.nhidden, xd.data_ptr(), base.data_ptr(), *tail)
        else:
            L.call("cgs_refine2d", self._wp, self._bp, self.nlayers, self.nhidden, xd.data_ptr(), float(real_sigmoid_mean), *tail)
        return best, step, traj


class Gan:
    """The three graph handles refiner_cpu reads off the reference's GAN object (synthetic/GAN.py:105-111)."""
    fake_samples, fake_sigmoid, fake_saliency = "fake_samples", "fake_sigmoid", "fake_saliency"

    def __init__(self, discriminator):
        self.D = discriminator


class Session:
    """``sess.run([gan.fake_sigmoid, gan.fake_saliency], feed_dict={gan.fake_samples: x})`` on the GPU discriminator."""

    def __init__(self, gan):
        self.gan, self.n_runs = gan, 0

    def run(self, fetches, feed_dict):
        single = not isinstance(fetches, (list, tuple))
        names = [fetches] if single else list(fetches)
        sig, sal = self.gan.D.sigmoid_and_saliency(feed_dict[Gan.fake_samples], want_saliency=Gan.fake_saliency in names)
        self.n_runs += 1
        out = [{Gan.fake_sigmoid: sig, Gan.fake_saliency: sal}[n].cpu().numpy() for n in names]
        return out[0] if single else out


class Refiner:
    """refiner_cpu.Refiner's surface (sampling/refiner_cpu.py:8-81) with the loop fused on the device."""

    def __init__(self, args):
        self.forward_steps, self.step_size, self.method = args.rollout_steps, args.rollout_rate, args.rollout_method
        if self.method not in _METHODS:
            raise NotImplementedError(self.method)

    def set_env(self, gan, sess, data):
        self.gan, self.sess, self.data = gan, sess, data

    def manipulate_sample(self, fake_batch, mode='deterministic'):
        if mode not in ('deterministic', 'probabilistic'):
            raise NotImplementedError
        D = self.gan.D
        real = self.data.next_batch(fake_batch.shape[0])                        # consumes the global RNG like :22
        real_sig, _ = D.sigmoid_and_saliency(real, want_saliency=False)
        baseline = np.mean(real_sig.cpu().numpy())                              # np.mean(real_sigmoid), :28
        best, step, traj = D.refine(fake_batch, baseline, self.forward_steps, self.step_size, self.method,
                                    want_traj=(mode == 'probabilistic'))
        self.optimal_step = step.cpu().numpy()
        if mode == 'probabilistic':                                             # per-call draw, float64 out (:72-76)
            n = len(fake_batch)
            pick = np.random.randint(self.forward_steps + 1, size=n)
            return traj.cpu().numpy().astype(np.float64)[np.arange(n), pick, :]
        return best.cpu().numpy().astype(fake_batch.dtype, copy=False)
