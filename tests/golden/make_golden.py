#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE'S OWN
CLASSES (imported from /root/reference) in the build container.

Runs only where /root/reference exists; never on the GPU box.  Nothing of the
reference is copied: this script imports its modules behind a test-only
``tensorflow`` facade (TF is not installable here),
drives them with deterministic torch-CPU networks (the oracle's restated ops),
and stores inputs + outputs as small .npz fixtures.

    python tests/golden/make_golden.py

Fixtures:
  g1_refiner_cpu.npz   refiner_cpu.Refiner.manipulate_sample, BASELINE config 1
  g2_policy.npz        PolicyAdaptive sgd / momentum / ladam(numpy) traces
  g3_collab_<arch>_K<k>_<mode>.npz   collaborator.Refiner.build_refiner
  g6_rejector.npz      Rejector.sampling accept behaviour over 3 calls
  g7_mh.npz            IndependenceSampler.sampling over 2 calls
  g8_toy.npz           ToyDataset.next_batch draws
  g9_metrics.npz       utils_sampling 2-D metrics (distance / good rate / KL / JS)
  g10_shape2d.npz      one 2-D D-shaping iteration: reference refiner (probabilistic) -> D SGD step -> reference refiner
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import numpy as np
import torch

np.float = float      # removed in numpy>=1.24; reference rejector.py:12,18 uses it
np.int = int

from oracle import nets_ref as N          # noqa: E402  (deterministic D / G-tail for the harness)
from oracle import sampling_ref as S      # noqa: E402  (only mlp_* helpers: the harness' D)


# --------------------------------------------------------------------------- TF facade
class T:
    """Immutable value wrapper with TF1 tensor semantics (``-=`` rebinds)."""
    __array_priority__ = 1000

    def __init__(self, t):
        self.t = t

    @staticmethod
    def un(x):
        return x.t if isinstance(x, T) else x

    def _new(self, r):
        if r.is_floating_point():
            r = r.detach().requires_grad_(True)
        return T(r)

    def get_shape(self):
        return types.SimpleNamespace(as_list=lambda: list(self.t.shape))

    @property
    def shape(self):
        return tuple(self.t.shape)

    def __add__(self, o): return self._new(self.t + T.un(o))
    __radd__ = __add__
    def __sub__(self, o): return self._new(self.t - T.un(o))
    def __rsub__(self, o): return self._new(T.un(o) - self.t)
    def __isub__(self, o): return self._new(self.t - T.un(o))
    def __mul__(self, o): return self._new(self.t * T.un(o))
    __rmul__ = __mul__
    def __truediv__(self, o): return self._new(self.t / T.un(o))
    def __pow__(self, o): return self._new(self.t ** T.un(o))


def make_tf():
    tf = types.ModuleType("tensorflow")
    tf.identity = lambda x: T(T.un(x).detach().clone().requires_grad_(True))
    tf.gradients = lambda ys, xs: [T(torch.autograd.grad(T.un(ys).sum(), T.un(xs), retain_graph=True)[0])]
    tf.reshape = lambda x, s: T(T.un(x).reshape(tuple(s)))
    tf.reduce_mean = lambda x, axis=None: T(T.un(x).mean() if axis is None else T.un(x).mean(dim=axis))
    tf.squeeze = lambda x: T(T.un(x).squeeze())
    tf.ones_like = lambda x: T(torch.ones_like(T.un(x).detach()))
    tf.greater = lambda a, b: T(T.un(a).detach() > T.un(b).detach())
    tf.sqrt = lambda x: T(torch.sqrt(T.un(x)))
    tf.clip_by_value = lambda x, clip_value_min, clip_value_max: T(
        torch.clamp(T.un(x).detach(), clip_value_min, clip_value_max).requires_grad_(True))
    tf.shape = lambda x: types.SimpleNamespace(eval=lambda: np.array(T.un(x).shape))

    def where(c, a, b):
        c = T.un(c)
        if isinstance(c, np.ndarray):
            c = torch.from_numpy(c)
        a, b = T.un(a).detach(), T.un(b).detach()
        c = c.view(-1, *([1] * (a.dim() - 1)))
        r = torch.where(c, a, b)
        return T(r.requires_grad_(True) if r.is_floating_point() else r)
    tf.where = where
    return tf


sys.modules["tensorflow"] = make_tf()
sys.path.insert(0, os.path.join(REF, "sampling"))
sys.path.insert(0, os.path.join(REF, "synthetic"))
import collaborator            # noqa: E402  reference sampling/collaborator.py
import refiner_cpu             # noqa: E402  reference sampling/refiner_cpu.py
import policy                  # noqa: E402  reference sampling/policy.py
import rejector                # noqa: E402  reference sampling/rejector.py
import idpsampler              # noqa: E402  reference sampling/idpsampler.py
import Datasets                # noqa: E402  reference synthetic/Datasets.py


def save(name, **kw):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **kw)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} kB")


def params_checksum(P):
    return np.array([float(sum(v.double().abs().sum() for v in P.values()))])


# --------------------------------------------------------------------------- G1
def g1_refiner_cpu():
    """BASELINE config 1: Imbal-8Gaussians, B=512, K=10, ladam rate 0.1."""
    Ws, bs = S.mlp_init(64, 6, seed=2019, scale=2.0)
    calls = []

    class Gan:
        fake_samples, fake_sigmoid, fake_saliency = "fake_samples", "fake_sigmoid", "fake_saliency"

    class Sess:
        def run(self, fetches, feed_dict):
            sig, sal = S.mlp_sigmoid_and_saliency(Ws, bs, feed_dict[Gan.fake_samples])
            calls.append(1)
            return [{"fake_sigmoid": sig, "fake_saliency": sal}[f] for f in fetches]

    args = types.SimpleNamespace(rollout_steps=10, rollout_rate=0.1, rollout_method="ladam")
    data = Datasets.ToyDataset(distr="Imbal-8Gaussians", scale=10.0, ratio=0.9)
    fake = (3.0 * np.random.RandomState(7).randn(512, 2)).astype(np.float32)
    out = {}
    for mode in ("deterministic", "probabilistic"):
        ref = refiner_cpu.Refiner(args)
        ref.set_env(Gan, Sess(), data)
        np.random.seed(2019)
        calls.clear()
        fake_in = fake.copy()
        res = ref.manipulate_sample(fake_in, mode)
        assert np.array_equal(fake_in, fake)
        out[mode] = res
        out[mode + "_calls"] = np.array([len(calls)])
        out[mode + "_dtype"] = np.array([str(res.dtype)])
    save("g1_refiner_cpu.npz", fake=fake, mlp_seed=np.array([2019]), mlp_scale=np.array([2.0]),
         W=np.array([w.numpy() for w in Ws], dtype=object), b=np.array([b.numpy() for b in bs], dtype=object),
         **out)


# --------------------------------------------------------------------------- G10
def g10_shape2d():
    """One iteration of the 2-D D-shaping loop (synthetic/main.py:354-370, mode "shape") and the refinement that follows it:
    the reference's refiner_cpu.Refiner (probabilistic) refines the generator batch -> one GradientDescentOptimizer step of D
    on (real, refined) -> the reference's Refiner (deterministic) on the SHAPED D.  The refiner is the reference's class; the
    optimizer step is TF's (not runnable here): it is restated as var -= lr*grad with torch autograd on the harness MLP
    (oracle.mlp_d_sgd_step), canonical flags of synthetic/run_shaping.sh:2 (batch 1000, lrd 8e-3, ratio 0.9, K 50 -> 10 here)."""
    B, K, lrd = 1000, 10, 8e-3
    Ws, bs = S.mlp_init(64, 6, seed=2019, scale=2.0)
    state = {"W": Ws, "b": bs}

    class Gan:
        fake_samples, fake_sigmoid, fake_saliency = "fake_samples", "fake_sigmoid", "fake_saliency"

    class Sess:
        def run(self, fetches, feed_dict):
            sig, sal = S.mlp_sigmoid_and_saliency(state["W"], state["b"], feed_dict[Gan.fake_samples])
            return [{"fake_sigmoid": sig, "fake_saliency": sal}[f] for f in fetches]

    args = types.SimpleNamespace(rollout_steps=K, rollout_rate=0.1, rollout_method="ladam")
    data = Datasets.ToyDataset(distr="Imbal-8Gaussians", scale=10.0, ratio=0.9)
    ref = refiner_cpu.Refiner(args)
    ref.set_env(Gan, Sess(), data)
    np.random.seed(2019)
    real_batch = data.next_batch(B)                                   # main.py:359
    noise_sample = (3.0 * np.random.RandomState(11).randn(B, 2)).astype(np.float32)      # stands for sess.run(gan.generates)
    refined = ref.manipulate_sample(noise_sample, "probabilistic")    # main.py:369 (float64 out)
    W1, b1, losses = S.mlp_d_sgd_step(Ws, bs, real_batch.astype(np.float32), refined.astype(np.float32), lrd)   # main.py:370
    state["W"], state["b"] = W1, b1
    eval_batch = (3.0 * np.random.RandomState(12).randn(512, 2)).astype(np.float32)
    np.random.seed(7)
    refined_after = ref.manipulate_sample(eval_batch, "deterministic")   # main.py:217 on the shaped D
    save("g10_shape2d.npz", real_batch=real_batch, noise_sample=noise_sample, refined=refined, lrd=np.array([lrd]), K=np.array([K]),
         d_loss=np.array(losses), eval_batch=eval_batch, refined_after=refined_after,
         W0=np.array([w.numpy() for w in Ws], dtype=object), b0=np.array([b.numpy() for b in bs], dtype=object),
         W1=np.array([w.numpy() for w in W1], dtype=object), b1=np.array([b.numpy() for b in b1], dtype=object))


# --------------------------------------------------------------------------- G2
def g2_policy():
    rs = np.random.RandomState(3)
    th0 = rs.randn(16, 2).astype(np.float32)
    grads = rs.randn(3, 16, 2).astype(np.float32) * 0.1
    losses = rs.randn(3, 16).astype(np.float32) * 0.3
    out = dict(theta0=th0, grads=grads, losses=losses)
    for method in ("sgd", "momentum", "ladam"):
        p = policy.PolicyAdaptive(0.1, method)
        th = th0.copy()
        tr = []
        for i in range(3):
            th = p.apply_gradient(th, grads[i], losses[i])
            tr.append(th.copy())
        out[method] = np.stack(tr)
        p.reset_moving_average()
        assert p.momentum is None and p.mean_square is None and p.loss is None
    # map-shaped momentum through the facade tensors (collaborator path)
    th0m = rs.randn(4, 3, 3, 8).astype(np.float32)
    gm = rs.randn(3, 4, 3, 3, 8).astype(np.float32)
    p = policy.PolicyAdaptive(0.5, "momentum")
    th = T(torch.from_numpy(th0m))
    tr = []
    for i in range(3):
        th = p.apply_gradient(th, T(torch.from_numpy(gm[i])))
        tr.append(th.t.detach().numpy().copy())
    out.update(theta0_map=th0m, grads_map=gm, momentum_map=np.stack(tr))
    save("g2_policy.npz", **out)


# --------------------------------------------------------------------------- G3/G4
def collab_case(arch, B, K, mode, rate, seed, constraints=None, compact=False):
    torch.manual_seed(0)
    P = N.init_params(arch, seed=2019, perturb=True)
    A = N.ARCHS[arch]
    rs = np.random.RandomState(seed)
    z = rs.uniform(-1, 1, (B,) + (tuple(A["g_in"]) if A.get("g_in") else (A["z_dim"],))).astype(np.float32)   # z, or the source image
    real = rs.uniform(-1, 1, (B,) + tuple(A["img"])).astype(np.float32)
    with torch.no_grad():
        feat0 = N.input_to_feature(arch, P, torch.from_numpy(z))

    disc = lambda x: T(N.discriminator(arch, P, T.un(x)))
    g_tail = lambda f: T(N.feature_to_data(arch, P, T.un(f)))
    loss = lambda l: T(torch.nn.functional.softplus(-T.un(l)))

    ref = collaborator.Refiner(rollout_steps=K, rollout_rate=rate)      # default method "momentum"
    ref.set_env(disc, g_tail, loss)
    if constraints is not None:
        ref.set_constraints(*constraints)
    np.random.seed(seed)
    state = np.random.get_state()
    img = ref.build_refiner(T(feat0), T(torch.from_numpy(real)), mode)
    np.random.set_state(state)
    idx = np.random.randint(K + 1, size=B) if mode == "probabilistic" else np.zeros(0, dtype=np.int64)
    assert ref.optimizer.momentum is None
    tag = f"g3_collab_{arch}_K{K}_{mode}" + ("_clip" if constraints else "")
    feat0_np = feat0.numpy()
    if compact:      # the large-batch cases: `real` only feeds the reference's dead real_logits (collaborator.py:44-45)
        tag = f"g3_collab_{arch}_B{B}_K{K}_{mode}"
        real = np.zeros(0, dtype=np.float32)
        if feat0_np.nbytes > (1 << 21):      # feature0 = oracle G head of z (tests/conftest.py: golden_feature0 re-derives it)
            feat0_np = np.zeros(0, dtype=np.float32)
    save(tag + ".npz", arch=np.array([arch]), z=z, real=real, feature0=feat0_np, K=np.array([K]),
         rate=np.array([rate]), mode=np.array([mode]), indices=idx, np_seed=np.array([seed]),
         constraints=np.array(constraints if constraints else [np.nan, np.nan]),
         images=img.t.detach().numpy(), default_logit=ref.default_logit.t.detach().numpy(),
         optimal_logit=ref.optimal_logit.t.detach().numpy(), optimal_step=ref.optimal_step.t.detach().numpy(),
         optimal_feature=ref.optimal_feature.t.detach().numpy(), params_checksum=params_checksum(P))


# --------------------------------------------------------------------------- G6/G7/G8
def g6_rejector():
    rs = np.random.RandomState(11)
    out = {}
    for tag, pct in (("p60", 60.0), ("p100", 100.0), ("none", None)):
        rej = rejector.Rejector()
        np.random.seed(2019)
        for c in range(3):
            sig = rs.beta(2, 2 + c, size=(257, 1)).astype(np.float32)
            samples = np.arange(257, dtype=np.float32).reshape(-1, 1)
            good = rej.sampling(samples, sig, shift_percent=pct)
            mask = np.zeros(257, dtype=bool); mask[good[:, 0].astype(int)] = True
            out[f"{tag}_sig{c}"] = sig; out[f"{tag}_mask{c}"] = mask; out[f"{tag}_M{c}"] = np.array([rej.D_tilde_M])
    rej = rejector.Rejector(); rej.set_score_max(np.array(0.93, dtype=np.float32))
    out["set_score_max_M"] = np.array([rej.D_tilde_M])
    save("g6_rejector.npz", **out)


def g7_mh():
    rs = np.random.RandomState(5)
    mh = idpsampler.IndependenceSampler(T=20)
    mh.set_score_curr(0.4)
    np.random.seed(2019)
    out = {}
    for c in range(2):
        sig = rs.beta(2, 2, size=(400, 1))
        samples = np.arange(400, dtype=np.float32).reshape(-1, 1)
        good = mh.sampling(samples, sig)
        out[f"sig{c}"] = sig; out[f"accepted{c}"] = good[:, 0].astype(np.int64) if len(good) else np.zeros(0, np.int64)
        out[f"dtype{c}"] = np.array([str(good.dtype)])
    save("g7_mh.npz", **out)


def g8_toy():
    out = {}
    for distr, ratio, B in (("Imbal-8Gaussians", 0.9, 512), ("8Gaussians", 0.5, 100), ("25Gaussians", 0.5, 60)):
        np.random.seed(2019)
        d = Datasets.ToyDataset(distr=distr, scale=10.0, ratio=ratio)
        out[distr] = d.next_batch(B)
    np.random.seed(2019)
    out["noise"] = Datasets.NoiseDataset().next_batch(33)
    save("g8_toy.npz", **out)


def g9_metrics():
    """2-D quality metrics of sampling/utils_sampling.py:132-184 on seeded clouds."""
    import importlib
    for name in ("matplotlib", "matplotlib.pyplot", "matplotlib.gridspec"):      # plotting deps are irrelevant here
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except ImportError:
                sys.modules[name] = types.ModuleType(name)
    import utils_sampling as U           # reference sampling/utils_sampling.py
    rs = np.random.RandomState(3)
    d = Datasets.ToyDataset(distr="Imbal-8Gaussians", scale=10.0, ratio=0.9)
    np.random.seed(1)
    real = d.next_batch(2000)
    model = np.concatenate([d.next_batch(1500)[::-1] + rs.randn(1500, 2) * 0.05, rs.randn(500, 2) * 4.0])
    thres = d.std * 4
    md, good = U.metrics_distance(model, d.centeroids, thres)
    fv, fa = U.freq_category(model, d.centeroids, thres)
    save("g9_metrics.npz", real=real, model=model, centeroids=d.centeroids, thres=np.array([thres]),
         mean_dist=np.array([md]), rate_good=np.array([good]), freqs_valid=fv, freqs_all=fa,
         kl=np.array([U.metrics_diversity(real, model, d.centeroids, thres)]),
         js=np.array([U.metrics_distribution(real, model, d.centeroids, thres)]),
         far_kl=np.array([U.metrics_diversity(real, model + 100.0, d.centeroids, thres)]))


if __name__ == "__main__":
    # python tests/golden/make_golden.py            -> every fixture
    # python tests/golden/make_golden.py dcgan64_B64 -> only the cases whose name contains an argument
    torch.set_num_threads(8)
    C = collab_case
    cases = [("g1", g1_refiner_cpu), ("g2", g2_policy), ("g6", g6_rejector), ("g7", g7_mh), ("g8", g8_toy), ("g9", g9_metrics),
             ("g10", g10_shape2d)]
    cases += [(f"mnist_K{K}", lambda K=K: C("mnist", 8, K, "deterministic", 0.1, seed=100 + K)) for K in (1, 5, 20)]
    cases += [
        ("mnist_K5_probabilistic", lambda: C("mnist", 8, 5, "probabilistic", 0.1, seed=7)),
        ("mnist_K5_clip", lambda: C("mnist", 8, 5, "deterministic", 0.5, seed=9, constraints=(0.05, 1.5))),
        ("dcgan32_K5", lambda: C("dcgan32", 4, 5, "deterministic", 0.1, seed=21)),      # 5x5 kernels: asymmetric SAME
        ("dcgan32_K5_probabilistic", lambda: C("dcgan32", 4, 5, "probabilistic", 0.1, seed=22)),
        ("dcgan64_K2", lambda: C("dcgan64", 2, 2, "deterministic", 0.1, seed=31)),     # the headline architecture (BASELINE configs 3/4)
        ("cyclegan_tiny_K3", lambda: C("cyclegan_tiny", 3, 3, "deterministic", 0.1, seed=41)),   # PatchGAN logit map: collaborator.py:34-37
        # batch 64 at the reference's own rollout lengths (nsgan/main.py:32,47: B=64, K=50; BASELINE configs[1..2]: K=20): enough
        # samples for the "optimal_step agrees on >= 99 %" statistic of SURVEY.md section 7 to be expressible
        ("mnist_B64_K50", lambda: C("mnist", 64, 50, "deterministic", 0.1, seed=51, compact=True)),
        ("dcgan32_B64_K20", lambda: C("dcgan32", 64, 20, "deterministic", 0.1, seed=52, compact=True)),
        # the net the headline metric is quoted on, at the headline's K, through the reference's own class (VERDICT r2 missing #2)
        ("dcgan64_B64_K20", lambda: C("dcgan64", 64, 20, "deterministic", 0.1, seed=53, compact=True)),
    ]
    for name, fn in cases:
        if len(sys.argv) == 1 or any(a in name for a in sys.argv[1:]):
            fn()
