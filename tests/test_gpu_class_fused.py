"""The reference's own call pattern on the fast path: ``Refiner.logical_batch`` -- G reference-semantics batches
(nsgan/main.py:32: batch_size 64) per launch through the CLASS SURFACE -- and the evaluate fill loop built on it
(``evaluate.FusedProposer`` / ``collaborate_fused``, nsgan/GAN.py:398-426), held to the goldens captured from the reference's
collaborator.Refiner and to the one-batch-at-a-time forms of the same code."""
import os
import warnings
from functools import partial

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import GOLDEN, golden_feature0
from oracle import nets_ref as N
from test_gpu_refine import check_against_golden, dev, load_case, relerr


def _wire(arch, P, B, d):
    """nsgan/GAN.py:174-181 verbatim against cgs_amd."""
    from cgs_amd import ops
    from cgs_amd.model import GAN
    from cgs_amd.sampling.collaborator import Refiner
    ops.reset_variables()
    self = GAN(arch, batch_size=B, device=d, params=P)

    def build(rollout_steps, rollout_rate):
        discriminator_refine = partial(self.discriminator, is_training=True, reuse=True)

        def loss_refine(logits):
            return ops.sigmoid_cross_entropy_with_logits(logits=logits, labels=ops.ones_like(logits))
        refiner = Refiner(rollout_steps=rollout_steps, rollout_rate=rollout_rate)
        refiner.set_env(discriminator_refine, self.feature_to_data, loss_refine)
        return refiner
    return self, build


def _slices(refiner, img, lo, hi):
    return (img[lo:hi], refiner.default_logit[lo:hi], refiner.optimal_logit[lo:hi], refiner.optimal_step[lo:hi], refiner.optimal_feature[lo:hi])


@pytest.mark.parametrize("name,G,at", [("mnist_B64_K50_deterministic", 32, (0, 13, 31)), ("mnist_K5_probabilistic", 5, (0, 2, 4)),
                                       ("mnist_K5_deterministic_clip", 4, (1, 3)), ("dcgan32_K5_probabilistic", 3, (0, 2)),
                                       ("dcgan32_B64_K20_deterministic", 8, (0, 7))])
def test_logical_batches_through_the_refiner_match_the_reference_golden(name, G, at):
    """build_refiner(feature[G*b]) with refiner.logical_batch = b: the logical batches that hold a reference golden's input must
    reproduce the golden whatever their neighbours hold (here: other random batches) -- D's batch statistics, the probabilistic
    index draw and the best-sample select are per logical batch -- and EVERY logical batch must equal a separate b-sample call of
    the same wiring."""
    g, arch, P, vmin, vmax = load_case(os.path.join(GOLDEN, f"g3_collab_{name}.npz"))
    d = dev()
    b = len(g["z"])
    gan, build = _wire(arch, P, b, d)
    K_, rate, mode = int(g["K"][0]), float(g["rate"][0]), str(g["mode"][0])
    f_gold = torch.from_numpy(golden_feature0(g, arch, P)).to(d)
    rs = np.random.RandomState(77)
    feats, idx = [], []
    for j in range(G):
        if j in at:
            feats.append(f_gold)
            idx.append(g["indices"] if mode == "probabilistic" else None)
        else:
            z = torch.from_numpy(rs.uniform(-1, 1, (b, gan.z_dim)).astype(np.float32)).to(d)
            feats.append(gan.input_to_feature(z).clone())
            idx.append(rs.randint(K_ + 1, size=b) if mode == "probabilistic" else None)
    feature = torch.cat(feats)
    indices = np.concatenate(idx) if mode == "probabilistic" else None
    fused = build(K_, rate)
    fused.logical_batch = b
    if vmin is not None:
        fused.set_constraints(vmin, vmax)
    for call in range(2):                                                   # second call = hipGraph replay
        img = fused.build_refiner(feature, None, mode, indices=indices)
        assert fused.path == "engine" and fused.use_graph and fused.graph_fallback is None and fused.why_generic is None
        assert gan.engine(G * b, use_graph=True, bn_groups=G).bn_groups == G
        for j in at:
            check_against_golden(g, *_slices(fused, img, j * b, (j + 1) * b), oracle_render=lambda f: N.feature_to_data(arch, P, f))
    # every logical batch == its own b-sample call (the convolutions only see a larger batch; small-grid launches may split K)
    single = build(K_, rate)
    if vmin is not None:
        single.set_constraints(vmin, vmax)
    ties = 0
    tol = 2e-3 if K_ <= 20 else 6e-3        # two fp32 evaluations of one K-step trajectory (the small launch may split K): rounding amplified over K steps
    for j in range(G):
        one = single.build_refiner(feats[j], None, mode, indices=idx[j])
        a = _slices(fused, img, j * b, (j + 1) * b)
        assert torch.allclose(a[1], single.default_logit, rtol=1e-4, atol=1e-5)
        same = a[3] == single.optimal_step
        ties += int((~same).sum())
        # (each of the two is held to the golden / the oracle within ``tol`` of max|ref| elsewhere: against each other, twice that)
        sm = same.cpu().numpy()
        assert relerr(a[2].cpu().numpy()[sm], single.optimal_logit.cpu().numpy()[sm]) < 2 * tol
        assert relerr(a[2].cpu().numpy(), single.optimal_logit.cpu().numpy()) < 20 * tol       # (a flipped select: two near-equal candidates)
        assert relerr(a[4].cpu().numpy()[sm], single.optimal_feature.cpu().numpy()[sm]) < 2 * tol
    assert ties <= max(1, G * b // 100)                                     # a flipped select only on a numerical tie


def test_logical_batches_on_the_generic_path_and_the_index_draw():
    """The generic (ops + autograd) path honours logical_batch by running the G batches one after the other; the probabilistic draw
    is one np.random.randint(K+1, size=b) per logical batch, in order (collaborator.py:54-56 per reference call)."""
    g, arch, P, vmin, vmax = load_case(os.path.join(GOLDEN, "g3_collab_mnist_K5_deterministic.npz"))
    d = dev()
    b, G = len(g["z"]), 3
    gan, build = _wire(arch, P, b, d)
    from cgs_amd.sampling.collaborator import Refiner
    f_gold = torch.from_numpy(g["feature0"]).to(d)
    z = torch.from_numpy(np.random.RandomState(1).uniform(-1, 1, (b, gan.z_dim)).astype(np.float32)).to(d)
    feature = torch.cat([f_gold, gan.input_to_feature(z), f_gold])
    gen = Refiner(int(g["K"][0]), float(g["rate"][0]))
    gen.set_env(lambda x: gan.discriminator(x, is_training=True, reuse=True), gan.feature_to_data, gan.loss_refine)
    gen.logical_batch = b
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        img = gen.build_refiner(feature, None, "deterministic")
    assert gen.path == "generic" and "lambda" in gen.why_generic
    for j in (0, 2):
        check_against_golden(g, *_slices(gen, img, j * b, (j + 1) * b), render=lambda f: gan.feature_to_data(f))
    eng = build(int(g["K"][0]), 0.1)
    eng.logical_batch = b
    np.random.seed(5)
    want = np.concatenate([np.random.randint(5 + 1, size=b) for _ in range(G)])
    np.random.seed(5)
    eng.build_refiner(feature, None, "probabilistic")
    np.testing.assert_array_equal(eng.indices_batch, want)
    from cgs_amd import lib
    eng.logical_batch = 5
    with pytest.raises(lib.CgsError):
        eng.build_refiner(feature, None, "deterministic")


def test_generic_fall_warns_once_with_the_reason_and_the_probe_sees_a_clipped_loss():
    """A wiring the engine detection does not recognise runs (correctly, slowly) on the generic path: the first fall warns with the
    reason, ``why_generic`` keeps it; a loss that equals BCE-vs-ones only on [-15, 15] is NOT mistaken for it (the probe reaches
    +-40); ``force_generic`` opts a recognised wiring out; a bound-method loss is probed once."""
    from cgs_amd import ops
    from cgs_amd.sampling.collaborator import Refiner
    d = dev()
    P = N.init_params("mnist", 2019, True)
    gan, build = _wire("mnist", P, 4, d)
    z = torch.from_numpy(np.random.RandomState(0).uniform(-1, 1, (4, 62)).astype(np.float32)).to(d)
    f0 = gan.input_to_feature(z)
    D = partial(gan.discriminator, is_training=True, reuse=True)
    bce = lambda l: ops.sigmoid_cross_entropy_with_logits(labels=ops.ones_like(l), logits=l)      # noqa: E731
    Refiner._WARNED.clear()
    r = Refiner(2, 0.1)
    r.set_env(D, gan.feature_to_data, lambda l: bce(l.clamp(-15.0, 15.0)))
    with pytest.warns(RuntimeWarning, match="generic"):
        r.build_refiner(f0, None, "deterministic")
    assert r.path == "generic" and "softplus" in r.why_generic
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                       # the second fall for the same reason is silent
        r.build_refiner(f0, None, "deterministic")
    ok = build(2, 0.1)
    img_e = ok.build_refiner(f0, None, "deterministic")
    assert ok.path == "engine" and ok.why_generic is None
    ok.force_generic = True
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        img_g = ok.build_refiner(f0, None, "deterministic")
    assert ok.path == "generic" and "force_generic" in ok.why_generic
    assert relerr(img_g.cpu().numpy(), img_e.cpu().numpy()) < 2e-3

    class Losses:
        calls = 0

        def loss(self, l):
            Losses.calls += 1
            return bce(l)
    holder = Losses()
    rb = Refiner(2, 0.1)
    rb.set_env(D, gan.feature_to_data, holder.loss)                         # a bound method: a fresh object at every access
    for _ in range(3):
        rb.set_env(D, gan.feature_to_data, holder.loss)
        rb.build_refiner(f0, None, "deterministic")
        assert rb.path == "engine"
    assert Losses.calls == 1                                                # probed once, not per call
    ops.reset_variables()


def test_refused_capture_keeps_one_engine(monkeypatch):
    """the hipGraph fallback catches the capture failure only, and re-uses the SAME engine eagerly (no second set of
    activation buffers); an error of the eager warm-up is not swallowed."""
    from cgs_amd import ops, lib
    d = dev()
    P = N.init_params("mnist", 2019, True)
    gan, build = _wire("mnist", P, 4, d)
    z = torch.from_numpy(np.random.RandomState(0).uniform(-1, 1, (4, 62)).astype(np.float32)).to(d)
    f0 = gan.input_to_feature(z)

    class _Refused:
        def __init__(self, *a, **k):
            raise RuntimeError("hipGraph capture refused (test)")
    monkeypatch.setattr(torch.cuda, "graph", _Refused)
    r = build(3, 0.1)
    r.build_refiner(f0, None, "deterministic")
    assert r.use_graph is False and "refused" in r.graph_fallback
    assert [k for k in gan._engines] == [(4, False, "f32", 1)]
    monkeypatch.undo()
    # a failure that is NOT the capture: surfaces as it is
    from cgs_amd.engine import RefineEngine
    r2 = build(3, 0.1)

    def boom(self, *a, **k):
        raise ValueError("warm-up failed (test)")
    monkeypatch.setattr(RefineEngine, "_program", boom)
    gan._engines.clear()
    with pytest.raises(ValueError):
        r2.build_refiner(f0, None, "deterministic")
    assert r2.use_graph is True and r2.graph_fallback is None
    assert issubclass(lib.GraphCaptureError, RuntimeError)
    ops.reset_variables()


def test_refresh_weights_repacks_in_the_engines_own_contraction():
    """engines of both contraction modes on one GAN; after an in-place weight update, the split-bf16 engine's
    refresh_weights() must re-pack ITS workspaces even though the exact-fp32 engine ran last -- a graph replay then equals a fresh
    engine on the new weights."""
    from cgs_amd import kernels as K
    from cgs_amd.engine import RefineEngine
    from cgs_amd.nets import to_device
    d = dev()
    arch, B, Ks = "dcgan32", 8, 2
    P = to_device(N.init_params(arch, 2019, True), d)
    z = torch.from_numpy(np.random.RandomState(4).uniform(-1, 1, (B, 100)).astype(np.float32)).to(d)
    bx = RefineEngine(arch, P, B, d, use_graph=True, contraction="bx6_all")
    f32 = RefineEngine(arch, P, B, d, use_graph=True, contraction="f32")
    bx.refine_from_z(z, Ks, 0.1); bx.refine_from_z(z, Ks, 0.1)           # capture + replay
    old = bx.refine_from_z(z, Ks, 0.1)[0].clone()
    f32.refine_from_z(z, Ks, 0.1)                                         # the other mode ran last
    assert K.L.get_contraction() == "f32"                                 # an engine call leaves the thread's mode as it found it
    with torch.no_grad():
        for k, v in P.items():
            if k.endswith("/w"):
                v.mul_(1.05)
    bx.refresh_weights()
    got = bx.refine_from_z(z, Ks, 0.1)[0].clone()                         # graph replay on the re-packed workspaces
    fresh = RefineEngine(arch, P, B, d, use_graph=False, contraction="bx6_all").refine_from_z(z, Ks, 0.1)[0]
    assert not torch.allclose(got, old, atol=1e-4)
    assert torch.equal(got, fresh)
    got32 = f32.refine_from_z(z, Ks, 0.1)[0]                              # the f32 engine re-syncs by itself at its next call
    fresh32 = RefineEngine(arch, P, B, d, use_graph=False).refine_from_z(z, Ks, 0.1)[0]
    assert torch.equal(got32, fresh32)


def test_engine_generate_and_score_against_the_oracle():
    """fake_images / fake_sigmoids (nsgan/GAN.py:153-155) on the engine, per logical batch under bn_groups."""
    from cgs_amd.engine import RefineEngine
    from cgs_amd.nets import to_device
    d = dev()
    P = N.init_params("mnist", 2019, True)
    b, G = 16, 3
    eng = RefineEngine("mnist", to_device(P, d), b * G, d, bn_groups=G)
    z = torch.from_numpy(np.random.RandomState(3).uniform(-1, 1, (b * G, 62)).astype(np.float32))
    img = eng.generate(z.to(d)).clone()
    sig = eng.score(img)
    assert tuple(sig.shape) == (b * G, 1)
    with torch.no_grad():
        for j in range(G):
            want_img = N.feature_to_data("mnist", P, N.input_to_feature("mnist", P, z[j * b:(j + 1) * b]))
            assert relerr(img[j * b:(j + 1) * b].cpu().numpy(), want_img.numpy()) < 1e-4
            want = torch.sigmoid(N.discriminator("mnist", P, want_img)).numpy()
            np.testing.assert_allclose(sig[j * b:(j + 1) * b].cpu().numpy(), want, rtol=0, atol=3e-5)
    # a PatchGAN-shaped logit map: the score is the mean sigmoid over the map (extension; P = 1 in the reference)
    from cgs_amd import kernels as K
    l = torch.from_numpy(np.random.RandomState(1).randn(5, 9, 9, 1).astype(np.float32) * 4).to(d)
    np.testing.assert_allclose(K.sigmoid_rowmean(l).cpu().numpy()[:, 0], torch.sigmoid(l.double()).reshape(5, -1).mean(1).cpu().numpy(), atol=1e-6)
    l2 = torch.tensor([[-100.0], [-20.0], [0.0], [20.0], [100.0]], device=d)
    np.testing.assert_allclose(K.sigmoid_rowmean(l2).cpu().numpy(), torch.sigmoid(l2.double()).cpu().numpy(), rtol=1e-6, atol=1e-38)


@pytest.mark.parametrize("G,eval_size,min_eff", [(4, 96, 0.2), (6, 50, None)])
def test_collaborate_fused_on_device_equals_one_batch_at_a_time(G, eval_size, min_eff):
    """nsgan/GAN.py:398-426 with G logical batches per device round == the same loop proposing one 16-sample batch at a time
    (collaborate + a b-sample engine, scores handed over as float32 like sess.run): same acceptances, efficiency, and the global
    numpy stream left where the one-batch loop leaves it; samples equal to the engines' fp32 agreement."""
    from cgs_amd.engine import RefineEngine
    from cgs_amd.evaluate import FusedProposer, collaborate, collaborate_fused
    from cgs_amd.sampling import IndependenceSampler
    d = dev()
    P = N.init_params("mnist", 2019, True)
    b, Ks = 16, 3
    gan, _ = _wire("mnist", P, b, d)
    one = RefineEngine("mnist", gan.build_variables(), b, d)
    base_z = torch.from_numpy(np.random.RandomState(8).uniform(-1, 1, (b, 62)).astype(np.float32)).to(d)
    base_img = one.generate(base_z).clone()
    base = (base_img.cpu().numpy(), one.score(base_img).cpu().numpy())

    def propose():
        z = torch.from_numpy(np.random.uniform(-1, 1, [b, 62]).astype(np.float32)).to(d)
        return one.refine_from_z(z, Ks, 0.1)[0].cpu().numpy()

    def score(batch):
        return one.score(torch.from_numpy(batch).to(d)).cpu().numpy()
    np.random.seed(21)
    want, eff = collaborate(propose, score, IndependenceSampler(T=3), eval_size, 0.4, base=base, min_efficiency=min_eff)
    tail = np.random.uniform(size=2)
    prop = FusedProposer(gan, Ks, 0.1, batch=b, groups=G, depth=2)
    np.random.seed(21)
    st = {}
    got, eff2 = collaborate_fused(prop, IndependenceSampler(T=3), eval_size, 0.4, base=base, min_efficiency=min_eff, stats=st)
    assert eff2 == eff and got.shape == want.shape == (eval_size, 28, 28, 1)
    np.testing.assert_array_equal(np.random.uniform(size=2), tail)
    np.testing.assert_allclose(got, want, rtol=0, atol=5e-3)
    assert st["proposed"] % b == 0 and st["proposed"] > 0
    # the real-set scoring helper: batch-size-b statistics like nsgan/GAN.py:388-390
    real = np.random.RandomState(2).uniform(-1, 1, (5 * b, 28, 28, 1)).astype(np.float32)
    s = prop.score_real(real)
    for j in range(5):
        np.testing.assert_allclose(s[j * b:(j + 1) * b], score(real[j * b:(j + 1) * b]), rtol=0, atol=3e-6)
