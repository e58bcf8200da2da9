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


class DShaper:
    """The D update of the 2-D shaping loop (synthetic/main.py:366-370): one ``tf.train.GradientDescentOptimizer(lrd)`` step on
    d_loss = mean BCE(D(real), 1) + mean BCE(D(refined), 0) (synthetic/GAN.py:69-74,98-99), run on the device IN PLACE on the
    ``MLPDiscriminator``'s own weight tensors (two per-sample forward/backward launches + one gradient/update launch).
    The refiner reads the same tensors, so the next ``manipulate_sample`` sees the shaped D."""

    def __init__(self, discriminator, lrd=1e-2):                      # synthetic/main.py:39 (--lrd 1e-2)
        self.D, self.lrd = discriminator, float(lrd)
        self.loss = torch.zeros(2, dtype=torch.float32, device=discriminator.dev)
        self.gw = [torch.zeros_like(t) for t in discriminator.w]
        self.gb = [torch.zeros_like(t) for t in discriminator.b]
        n = discriminator.nlayers
        self._gwp = (C.c_void_p * n)(*[t.data_ptr() for t in self.gw])
        self._gbp = (C.c_void_p * n)(*[t.data_ptr() for t in self.gb])
        self._ws = None

    def _run(self, real, refined, lr):
        D = self.D
        xr, xf = D._x(real), D._x(refined)
        need = int(L.load().cgs_mlp2d_train_ws_bytes(xr.shape[0] + xf.shape[0], D.nlayers))
        if self._ws is None or self._ws.numel() * 4 < need:
            self._ws = torch.empty(need // 4 + 4, dtype=torch.float32, device=D.dev)
        L.call("cgs_mlp2d_d_step", D._wp, D._bp, D.nlayers, D.nhidden, xr.data_ptr(), xr.shape[0], xf.data_ptr(), xf.shape[0],
               float(lr), self._gwp, self._gbp, self.loss.data_ptr(), self._ws.data_ptr(), self._ws.numel() * 4,
               torch.cuda.current_stream(D.dev).cuda_stream)
        return self.loss

    def loss_and_grads(self, real, refined):
        """((d_loss_real, d_loss_fake) device tensor, [dW...], [db...]) without touching the weights."""
        return self._run(real, refined, 0.0), self.gw, self.gb

    def step(self, real, refined):
        """One SGD step of D; returns (d_loss_real, d_loss_fake) as evaluated BEFORE the update (device tensor)."""
        return self._run(real, refined, self.lrd)


def shape_step(refiner, shaper, noise_sample, real_batch):
    """One iteration of synthetic/main.py:366-370:
    ``refined = refiner.manipulate_sample(noise_sample, 'probabilistic'); sess.run(d_optim, {inputs: real, generates: refined})``."""
    refined = refiner.manipulate_sample(noise_sample, 'probabilistic')
    return shaper.step(real_batch, refined), refined


def proposer(refiner, generate, discriminator):
    """propose() / score() closures for ``evaluate.collaborate`` on the 2-D path (synthetic/main.py:240-243):
    ``generate()`` -> a generator batch (``sess.run(gan.generates, {z: noise.next_batch(n)})`` in the reference -- G is outside
    the hot path, so the caller supplies it); propose = the device refiner on it; score = the device D's sigmoid."""
    def propose():
        return refiner.manipulate_sample(generate())

    def score(batch):
        return discriminator.sigmoid_and_saliency(batch, want_saliency=False)[0].cpu().numpy()      # float32 [n, 1] like sess.run(gan.fake_sigmoid) (synthetic/main.py:232)

    return propose, score


def evaluate_collaborative(refiner, discriminator, generate, eval_batch, target_batch, centeroids, std, mh_sampler=None):
    """The "shape"-mode evaluation of synthetic/main.py:215-263 on the device refiner: refine the evaluation batch, report its
    quality, then D-score -> MH fill (thinning T = 20, chain seeded with mean(real_sigmoid)) until ``len(eval_batch)`` samples
    are accepted (proposal counter advancing only for productive batches, :251) and report the collaborative sample's
    quality.  Returns {"refinement": {...}, "collaborate": {..., "eff": accepted / proposed}} with the reference's four 2-D
    metrics (utils_sampling.py:132-184; ``thres`` = 4 std as in main.py:221)."""
    from . import metrics as Mx
    from .evaluate import collaborate
    from .sampling import IndependenceSampler
    thres = std * 4

    def quality(samples):
        mean_dist, good = Mx.metrics_distance(samples, centeroids, thres)
        return {"mean_dist": float(mean_dist), "good": float(good),
                "kl": float(Mx.metrics_diversity(target_batch, samples, centeroids, thres)),
                "js": float(Mx.metrics_distribution(target_batch, samples, centeroids, thres))}
    _, score = proposer(refiner, generate, discriminator)
    real_sigmoid = score(target_batch)                                                 # main.py:116
    out = {"standard": quality(eval_batch)}
    refined = refiner.manipulate_sample(eval_batch)                                    # :217
    out["refinement"] = quality(refined)
    mh = mh_sampler if mh_sampler is not None else IndependenceSampler(T=20)           # main.py:80
    propose = lambda: refiner.manipulate_sample(generate())                            # :239-241 (eval_size proposals per round)
    samples, eff = collaborate(propose, score, mh, len(eval_batch), float(np.mean(real_sigmoid)),
                               base=(refined, score(refined)), count_only_productive=True)      # :229-253
    out["collaborate"] = dict(quality(samples), eff=float(eff))
    return out
