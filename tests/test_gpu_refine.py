"""End-to-end parity of the refinement loop on the GPU: the fused engine and the generic
(ops + autograd) path against (a) the golden vectors captured from the reference's
collaborator.Refiner and (b) the CPU oracle, plus size-independent properties at larger sizes.

Tolerances (fp32, SURVEY.md section 7 'hard parts'): one forward pass 1e-4 relative on logits;
K-step trajectories 2e-3 on logits / features / images; optimal_step must agree wherever the two
candidate logits differ by more than the trajectory tolerance."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import GOLDEN, golden_feature0
from oracle import nets_ref as N
from oracle import sampling_ref as S

G3 = sorted(glob.glob(os.path.join(GOLDEN, "g3_collab_*.npz")))


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def relerr(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return np.abs(got - want).max() / (np.abs(want).max() + 1e-30)


def grad_close(got, want):
    """Gradient parity per sample.  An fp32 pre-activation that lands on the other side of a (l)relu kink than
    in the reference arithmetic multiplies one gradient entry by 5 (or zeroes it): a measure-zero event that
    perturbs that sample's whole gradient map (and, through batch-norm means, the others at the 1e-4 level).
    So: every sample within 5e-2 of max|ref|, and all but at most one sample within 2e-3."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    B = got.shape[0]
    per = np.abs(got - want).reshape(B, -1).max(1) / (np.abs(want).max() + 1e-30)
    return bool(per.max() < 5e-2 and (per < 2e-3).sum() >= B - 1), per


def check_against_golden(g, img, dl, ol, os_, of, traj_tol=2e-3, render=None, oracle_render=None, image_drift=25.0):
    """Against what the reference's own collaborator.Refiner produced.  optimal_step: EQUAL everywhere except where the two
    arithmetics' best logits tie within the trajectory tolerance (a flipped select is only tolerable on a numerical tie), and
    from batch 64 up equal on >= 99 % of the samples.  Measured on MI355X (tools/golden_diag.py): steps
    100 % equal in all 12 goldens, logits <= 2.8e-4 (1.5e-3 at K = 50), features <= 1e-4.
    Images: the DCGAN G tails amplify feature drift ~300x (inference bn divides by sqrt(moving_var ~ 0.02) per layer: a 7e-5
    feature difference is a 2e-2 image difference), so the K-step image against the golden is only a sanity bound; the image is
    pinned tightly (1e-4) twice instead: the engine's render of the GOLDEN feature against the golden image, and the returned
    image against the ORACLE's render of the engine's own optimal feature."""
    assert relerr(dl.cpu().numpy(), g["default_logit"]) < 1e-4
    assert relerr(ol.cpu().numpy(), g["optimal_logit"]) < traj_tol
    steps_ok = os_.cpu().numpy() == g["optimal_step"]
    if not steps_ok.all():
        bad = ~steps_ok
        assert np.all(np.abs(ol.cpu().numpy()[bad] - g["optimal_logit"][bad]) < traj_tol * np.abs(g["optimal_logit"]).max())
    ok = steps_ok
    if len(ok) >= 64:
        assert ok.mean() >= 0.99
    assert relerr(of.cpu().numpy()[ok], g["optimal_feature"][ok]) < traj_tol
    assert relerr(img.cpu().numpy()[ok], g["images"][ok]) < image_drift * traj_tol
    if oracle_render is not None:
        with torch.no_grad():
            mine = oracle_render(of.cpu())
        assert relerr(img.cpu().numpy(), mine.numpy()) < 1e-4
    if render is not None:
        again = render(torch.from_numpy(g["optimal_feature"]).to(img.device))
        assert relerr(again.cpu().numpy(), g["images"]) < 1e-4
        assert torch.equal(render(of), img)          # returned images ARE G_tail(optimal_feature)


def load_case(path):
    g = np.load(path, allow_pickle=True)
    arch = str(g["arch"][0])
    P = N.init_params(arch, seed=2019, perturb=True)
    chk = float(sum(v.double().abs().sum() for v in P.values()))
    assert abs(chk - float(g["params_checksum"][0])) <= 1e-9 * chk
    c = g["constraints"]
    vmin, vmax = (None, None) if np.isnan(c[0]) else (float(c[0]), float(c[1]))
    return g, arch, P, vmin, vmax


@pytest.mark.parametrize("path", G3, ids=lambda p: os.path.basename(p)[10:-4])
@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "hipgraph"])
@pytest.mark.parametrize("contraction", ["f32", "bx6_all"])
def test_engine_matches_reference_golden(path, use_graph, contraction):
    """Every golden captured from the reference's own collaborator.Refiner, on the exact-fp32 contraction and with every layer the
    split-bf16 kernel can serve running on it ("bx6_all": at these small batches the production mode "bx6" would keep them all on
    fp32) -- the same assertions at the same tolerances, except the K-step image-drift SANITY bound (60x instead of 25x the
    trajectory tolerance in the split-bf16 mode: see the comment at the call)."""
    from cgs_amd.engine import RefineEngine
    from cgs_amd.nets import to_device
    g, arch, P, vmin, vmax = load_case(path)
    if contraction != "f32" and "cyclegan_tiny" in path:
        pytest.skip("no layer of this net has >= 128 output channels")
    d = dev()
    eng = RefineEngine(arch, to_device(P, d), len(g["z"]), d, use_graph=use_graph, contraction=contraction)
    f0 = eng.input_to_feature(torch.from_numpy(g["z"]).to(d))
    feature0 = golden_feature0(g, arch, P)
    assert relerr(f0.cpu().numpy(), feature0) < 1e-4                  # the propose step (G head)
    mode = str(g["mode"][0])
    for _ in range(2 if use_graph else 1):                              # 2nd call = graph replay
        out = eng.refine(torch.from_numpy(feature0).to(d), int(g["K"][0]), float(g["rate"][0]), "momentum", mode,
                         g["indices"] if mode == "probabilistic" else None, vmin, vmax)
        # Every pin is the same in both modes -- logits / features / steps at the trajectory tolerance, the two 1e-4 render checks --
        # except the SANITY bound on the K-step image against the golden, which is ~300x-amplified feature drift (see
        # check_against_golden): the split-bf16 kernel rounds like an ordinary fp32 chain (error / sum|a||b| mean 1.8e-8, the level of
        # the oracle's own torch-CPU arithmetic, 2.1e-8) where the exact-fp32 MFMA kernel is 3.5x finer (5e-9; tools/bx6_accuracy.py),
        # and the amplified drift scales with it (measured: 5.1e-2 against 2.2e-2 on the dcgan32 K = 20 golden).
        check_against_golden(g, *[t.clone() for t in out], render=lambda f: eng.feature_to_data(f).clone(),
                             oracle_render=lambda f: N.feature_to_data(arch, P, f), image_drift=25.0 if contraction == "f32" else 60.0)


@pytest.mark.parametrize("path", [p for p in G3 if "K5" in p], ids=lambda p: os.path.basename(p)[10:-4])
def test_refiner_class_generic_and_engine_paths(path):
    """collaborator.Refiner with (a) opaque callables over cgs_amd.ops (autograd path) and
    (b) the GAN object's own methods (engine path)."""
    from cgs_amd import ops
    from cgs_amd.model import GAN
    from cgs_amd.sampling.collaborator import Refiner
    g, arch, P, vmin, vmax = load_case(path)
    d = dev()
    ops.reset_variables()
    gan = GAN(arch, batch_size=len(g["z"]), device=d, params=P)
    mode = str(g["mode"][0])
    idx = g["indices"] if mode == "probabilistic" else None
    f0 = torch.from_numpy(g["feature0"]).to(d)
    real = torch.from_numpy(g["real"]).to(d)
    # (a) generic: wrap in lambdas so the engine detection cannot trigger
    ref = Refiner(int(g["K"][0]), float(g["rate"][0]))
    ref.set_env(lambda x: gan.discriminator(x, is_training=True, reuse=True), lambda f: gan.feature_to_data(f), gan.loss_refine)
    if vmin is not None:
        ref.set_constraints(vmin, vmax)
    assert ref._engine_for(len(f0)) is None
    img = ref.build_refiner(f0, real, mode, indices=idx)
    check_against_golden(g, img, ref.default_logit, ref.optimal_logit, ref.optimal_step, ref.optimal_feature,
                         render=lambda f: gan.feature_to_data(f))
    assert ref.optimizer.momentum is None                                   # reset (collaborator.py:86)
    # (b) engine via the class surface (nsgan/GAN.py:179-183 wiring)
    ref2 = gan.build_refiner(int(g["K"][0]), float(g["rate"][0]))
    if vmin is not None:
        ref2.set_constraints(vmin, vmax)
    assert ref2._engine_for(len(f0)) is not None
    img2 = ref2.build_refiner(f0, real, mode, indices=idx)
    check_against_golden(g, img2, ref2.default_logit, ref2.optimal_logit, ref2.optimal_step, ref2.optimal_feature,
                         oracle_render=lambda f: N.feature_to_data(arch, P, f))
    # one evaluation: logits + gradient vs the oracle (collaborator.py:26-39)
    lm, grad = ref.compute_forward_logits_and_grad(f0)
    lm_o, grad_o = S.forward_logits_and_grad(torch.from_numpy(g["feature0"]), lambda f: N.feature_to_data(arch, P, f),
                                             lambda x: N.discriminator(arch, P, x))
    assert relerr(lm.cpu().numpy(), lm_o.numpy()) < 1e-4
    ok, per = grad_close(grad.cpu().numpy(), grad_o.numpy())
    assert ok, per
    lm_e, grad_e = gan.engine(len(f0)).compute_forward_logits_and_grad(f0)
    assert relerr(lm_e.cpu().numpy(), lm_o.numpy()) < 1e-4
    ok, per = grad_close(grad_e.cpu().numpy(), grad_o.numpy())
    assert ok, per
    ops.reset_variables()


@pytest.mark.parametrize("path", [p for p in G3 if "K5" in p or "mnist_B64_K50" in p], ids=lambda p: os.path.basename(p)[10:-4])
def test_generic_path_replays_a_captured_hipgraph(path):
    """wirings the engine detection cannot see through (lambdas, another D, another loss) ran ~1000 launches of Python +
    autograd per call.  The reference's build_refiner is a graph BUILDER (its callables run once, sess.run replays the graph); here the
    generic loop of a call signature is captured into a hipGraph at its second call and replayed afterwards.  Held to: the first
    (eager), second (capturing) and third (replayed) call all reproduce the reference golden; replays are bit-equal to each other
    and to the eager launch of the same kernels; ops.bn's moving-average update is part of the recorded program (it advances once per
    call, eager or replayed); an in-place weight update drops the graph (the packed copies it reads would be stale); a refused
    capture falls back to eager launches and says why."""
    from cgs_amd import kernels as Kn
    from cgs_amd import ops
    from cgs_amd.model import GAN
    from cgs_amd.sampling.collaborator import Refiner
    g, arch, P, vmin, vmax = load_case(path)
    d = dev()
    ops.reset_variables()
    gan = GAN(arch, batch_size=len(g["z"]), device=d, params=P)
    mode = str(g["mode"][0])
    idx = g["indices"] if mode == "probabilistic" else None
    f0 = torch.from_numpy(golden_feature0(g, arch, P)).to(d)
    real = torch.from_numpy(g["real"]).to(d)

    def make():
        r = Refiner(int(g["K"][0]), float(g["rate"][0]))
        r.set_env(lambda x: gan.discriminator(x, is_training=True, reuse=True), lambda f: gan.feature_to_data(f), gan.loss_refine)
        if vmin is not None:
            r.set_constraints(vmin, vmax)
        return r
    ref = make()
    mms = [v for k, v in ops.variables().items() if k.startswith("discriminator") and k.endswith("moving_mean")]
    mm = mms[0] if mms else None                                                  # (a batch norm of D: the refiner runs it in training mode)
    outs = []
    for call in range(4):
        before = None if mm is None else mm.clone()
        img = ref.build_refiner(f0, real, mode, indices=idx)
        assert ref.path == "generic" and ref.graph_fallback is None
        check_against_golden(g, img, ref.default_logit, ref.optimal_logit, ref.optimal_step, ref.optimal_feature,
                             render=lambda f: gan.feature_to_data(f))
        outs.append([t.clone() for t in (img, ref.default_logit, ref.optimal_logit, ref.optimal_step, ref.optimal_feature)])
        assert ref.optimizer.momentum is None
        assert len(ref._generic_graphs) == (0 if call == 0 else 1)               # call 0 eager, call 1 captures, 2.. replay
        if mm is not None:
            assert not torch.equal(mm, before)                                    # the moving averages move at EVERY call (ops.py:19-26, decay 0.9)
    for a, b in zip(outs[0], outs[2]):
        assert torch.equal(a, b)                                                  # replay == eager launch of the same kernels
    for a, b in zip(outs[2], outs[3]):
        assert torch.equal(a, b)
    # another input through the same graph == a fresh eager refiner on it
    f1 = (f0 * 0.9 + 0.05).contiguous()
    img_g = ref.build_refiner(f1, real, mode, indices=idx)
    eager = make(); eager.use_graph = False
    img_e = eager.build_refiner(f1, real, mode, indices=idx)
    assert len(eager._generic_graphs) == 0 and torch.equal(img_g, img_e) and torch.equal(ref.optimal_step, eager.optimal_step)
    assert torch.equal(ref.optimal_logit, eager.optimal_logit)
    # a weight moves on in place: the recorded program would read the stale packed copy -> the graph is dropped and re-captured
    key = [k for k in ops.variables() if k.startswith("discriminator") and k.endswith("/w")][0]
    (gg,) = ref._generic_graphs.values()
    assert gg.valid()
    with torch.no_grad():
        ops.variables()[key].mul_(1.01)
    assert not gg.valid()
    img_w = ref.build_refiner(f1, real, mode, indices=idx)                        # (eager warm-up on the capture stream + new capture)
    (gg2,) = ref._generic_graphs.values()
    assert gg2 is not gg and gg2.valid()
    img_w2 = eager.build_refiner(f1, real, mode, indices=idx)
    assert torch.equal(img_w, img_w2) and not torch.equal(img_w, img_g)
    with torch.no_grad():
        ops.variables()[key].div_(1.01)
    # set_env with OTHER callables drops the recorded programs (they belong to the old ones); the same objects again keep them
    other = make()
    for _ in range(3):
        other.build_refiner(f0, real, mode, indices=idx)
    assert len(other._generic_graphs) == 1
    other.set_env(other.discriminator, other.feature_to_data, other.func_loss)
    assert len(other._generic_graphs) == 1
    scale = 0.5
    other.set_env(lambda x: gan.discriminator(x, is_training=True, reuse=True) * scale, other.feature_to_data, other.func_loss)
    assert len(other._generic_graphs) == 0 and not other._generic_seen
    img_s = [other.build_refiner(f0, real, mode, indices=idx) for _ in range(3)]
    assert len(other._generic_graphs) == 1 and torch.equal(img_s[0], img_s[2]) and not torch.equal(other.default_logit, ref.default_logit)
    # a callable that synchronises with the host cannot be captured: eager launches in the same process, and the refiner says why
    bad = Refiner(int(g["K"][0]), float(g["rate"][0]))

    def syncing_d(x):
        y = gan.discriminator(x, is_training=True, reuse=True)
        float(y.sum().item())                                                     # a host read-back: legal eagerly, refused inside a capture
        return y
    bad.set_env(syncing_d, lambda f: gan.feature_to_data(f), gan.loss_refine)
    if vmin is not None:
        bad.set_constraints(vmin, vmax)
    for call in range(3):
        img_b = bad.build_refiner(f0, real, mode, indices=idx)
        check_against_golden(g, img_b, bad.default_logit, bad.optimal_logit, bad.optimal_step, bad.optimal_feature,
                             render=lambda f: gan.feature_to_data(f))
    assert bad.graph_fallback and bad.use_graph is False and len(bad._generic_graphs) == 0
    ops.reset_variables()


@pytest.mark.parametrize("path", G3, ids=lambda p: os.path.basename(p)[10:-4])
def test_reference_verbatim_wiring_takes_the_engine(path):
    """The four lines of nsgan/GAN.py:174-181 written against cgs_amd exactly as the reference writes them -- a
    ``functools.partial`` of the discriminator and a LOCAL loss closure -- must reach the fused engine, replay its
    hipGraph from the second call on, and reproduce every reference golden."""
    from functools import partial
    from cgs_amd import ops
    from cgs_amd.model import GAN
    from cgs_amd.sampling.collaborator import Refiner
    g, arch, P, vmin, vmax = load_case(path)
    d = dev()
    ops.reset_variables()
    self = GAN(arch, batch_size=len(g["z"]), device=d, params=P)
    rollout_steps, rollout_rate = int(g["K"][0]), float(g["rate"][0])
    # ---- nsgan/GAN.py:174-181 ------------------------------------------------------------------
    discriminator_refine = partial(self.discriminator, is_training=True, reuse=True)
    def loss_refine(logits):
        return ops.sigmoid_cross_entropy_with_logits(logits=logits, labels=ops.ones_like(logits))
    refiner = Refiner(rollout_steps=rollout_steps, rollout_rate=rollout_rate)
    refiner.set_env(discriminator_refine, self.feature_to_data, loss_refine)
    # ---------------------------------------------------------------------------------------------
    if vmin is not None:
        refiner.set_constraints(vmin, vmax)
    mode = str(g["mode"][0])
    idx = g["indices"] if mode == "probabilistic" else None
    f0 = torch.from_numpy(golden_feature0(g, arch, P)).to(d)
    real = torch.from_numpy(g["real"]).to(d)
    for call in range(2):                                                   # the second call replays the captured graph
        img = refiner.build_refiner(f0, real, mode, indices=idx)
        assert refiner.path == "engine" and refiner.use_graph and refiner.graph_fallback is None
        check_against_golden(g, img, refiner.default_logit, refiner.optimal_logit, refiner.optimal_step, refiner.optimal_feature,
                             oracle_render=lambda f: N.feature_to_data(arch, P, f))
    assert len(self.engine(len(f0), use_graph=True)._graphs) == 1
    if "dcgan64_B64" in path:      # the class surface's opt-in: the same wiring with refiner.contraction set (every eligible layer: "bx6_all")
        refiner.contraction = "bx6_all"
        img = refiner.build_refiner(f0, real, mode, indices=idx)
        assert refiner.path == "engine" and self.engine(len(f0), use_graph=True, contraction="bx6_all").contraction == "bx6_all"
        check_against_golden(g, img, refiner.default_logit, refiner.optimal_logit, refiner.optimal_step, refiner.optimal_feature,
                             oracle_render=lambda f: N.feature_to_data(arch, P, f), image_drift=60.0)
    ops.reset_variables()


def test_engine_detection_accepts_only_what_the_engine_computes():
    """Which set_env arguments select the fused engine: the GAN's own methods in every spelling of the reference's wiring, a loss
    that IS softplus(-logit); everything else stays on the generic path (and ``path`` says so)."""
    from functools import partial
    from cgs_amd import ops
    from cgs_amd.model import GAN
    from cgs_amd.sampling.collaborator import Refiner
    d = dev()
    ops.reset_variables()
    gan = GAN("mnist", batch_size=4, device=d, params=N.init_params("mnist", 2019, True))
    other = GAN("mnist", batch_size=4, device=d)
    bce = lambda l: ops.sigmoid_cross_entropy_with_logits(labels=ops.ones_like(l), logits=l)

    def owner(dfn, ffn, loss, method="momentum"):
        r = Refiner(2, 0.1, method)
        r.set_env(dfn, ffn, loss)
        return r._engine_owner()
    D = partial(gan.discriminator, is_training=True, reuse=True)
    assert owner(D, gan.feature_to_data, bce) is gan
    assert owner(gan.discriminator_refine, gan.feature_to_data, GAN.loss_refine) is gan
    assert owner(partial(gan.discriminator, reuse=True), partial(gan.feature_to_data, is_training=False), bce) is gan
    assert owner(partial(D, reuse=True), gan.feature_to_data, bce) is gan                     # a partial of a partial
    assert owner(D, gan.feature_to_data, lambda l: torch.nn.functional.softplus(-l)) is gan    # any spelling of the same loss
    # not the engine's computation:
    assert owner(partial(gan.discriminator, is_training=False, reuse=True), gan.feature_to_data, bce) is None   # inference-mode bn in D
    assert owner(D, partial(gan.feature_to_data, is_training=True), bce) is None              # batch statistics in the G tail
    assert owner(D, other.feature_to_data, bce) is None                                       # two different models
    assert owner(partial(gan.discriminator, gan), gan.feature_to_data, bce) is None           # positional arguments bound
    assert owner(lambda x: gan.discriminator(x, True, True), gan.feature_to_data, bce) is None
    assert owner(D, gan.feature_to_data, lambda l: ops.sigmoid_cross_entropy_with_logits(labels=ops.zeros_like(l), logits=l)) is None
    assert owner(D, gan.feature_to_data, lambda l: bce(l).mean()) is None                     # a reduction
    assert owner(D, gan.feature_to_data, lambda l: 2.0 * bce(l)) is None
    assert owner(D, gan.feature_to_data, lambda l: l.nonexistent_attribute) is None           # a loss the probe cannot evaluate
    assert owner(D, gan.feature_to_data, bce, "ladam") is None                                # (needs a loss the map refiner never passes)
    # the generic path reports itself, and a rejected loss is still differentiated as given
    z = torch.from_numpy(np.random.RandomState(0).uniform(-1, 1, (4, 62)).astype(np.float32)).to(d)
    f0 = gan.input_to_feature(z)
    r = Refiner(2, 0.1)
    r.set_env(D, gan.feature_to_data, lambda l: 2.0 * bce(l))
    img2 = r.build_refiner(f0, None, "deterministic")
    assert r.path == "generic"
    r1 = Refiner(2, 0.2)                                                                        # 2 x loss at rate 0.1 == loss at rate 0.2
    r1.set_env(D, gan.feature_to_data, bce)
    img1 = r1.build_refiner(f0, None, "deterministic")
    assert r1.path == "engine"
    assert relerr(img2.cpu().numpy(), img1.cpu().numpy()) < 2e-3
    # the TF-shaped loss entry point itself: ones / zeros / a label tensor, keyword-only
    l = torch.linspace(-9, 9, 24, device=d).reshape(24, 1)
    want1 = torch.nn.functional.softplus(-l.double().cpu())
    assert relerr(ops.sigmoid_cross_entropy_with_logits(labels=ops.ones_like(l), logits=l).cpu().numpy(), want1.numpy()) < 1e-6
    assert relerr(ops.sigmoid_cross_entropy_with_logits(labels=ops.zeros_like(l), logits=l).cpu().numpy(),
                  torch.nn.functional.softplus(l.double().cpu()).numpy()) < 1e-6
    zl = torch.rand(24, 1, device=d)
    want = torch.nn.functional.binary_cross_entropy_with_logits(l.double().cpu(), zl.double().cpu(), reduction="none")
    assert relerr(ops.sigmoid_cross_entropy_with_logits(labels=zl, logits=l).cpu().numpy(), want.numpy()) < 1e-6
    with pytest.raises(ValueError):
        ops.sigmoid_cross_entropy_with_logits(l, l)
    ops.reset_variables()


def test_refused_graph_capture_falls_back_to_eager_in_process(monkeypatch):
    """Refiner.use_graph defaults to True; a capture that fails must not fail the call: same engine code launched eagerly, and
    the refiner records why."""
    from cgs_amd import ops
    from cgs_amd.model import GAN
    d = dev()
    ops.reset_variables()
    P = N.init_params("mnist", 2019, True)
    gan = GAN("mnist", batch_size=4, device=d, params=P)
    z = torch.from_numpy(np.random.RandomState(0).uniform(-1, 1, (4, 62)).astype(np.float32)).to(d)
    f0 = gan.input_to_feature(z)
    good = gan.build_refiner(3, 0.1)
    img_g = good.build_refiner(f0, None, "deterministic")
    assert good.path == "engine" and good.use_graph and good.graph_fallback is None

    class _Refused:
        def __init__(self, *a, **k):
            raise RuntimeError("hipGraph capture refused (test)")
    gan._engines.clear()
    monkeypatch.setattr(torch.cuda, "graph", _Refused)
    ref = gan.build_refiner(3, 0.1)
    img_e = ref.build_refiner(f0, None, "deterministic")
    assert ref.path == "engine" and ref.use_graph is False and "refused" in ref.graph_fallback
    assert torch.equal(img_e, img_g) and torch.equal(ref.optimal_step, good.optimal_step)   # the same kernels in the same order
    ops.reset_variables()


def test_probabilistic_draw_and_ladam_guard():
    from cgs_amd import ops, lib
    from cgs_amd.model import GAN
    d = dev()
    ops.reset_variables()
    gan = GAN("mnist", batch_size=4, device=d, params=N.init_params("mnist", 2019, True))
    ref = gan.build_refiner(3, 0.1)
    z = torch.from_numpy(np.random.RandomState(0).uniform(-1, 1, (4, 62)).astype(np.float32)).to(d)
    f0 = gan.input_to_feature(z)
    np.random.seed(5)
    want = np.random.randint(3 + 1, size=4)
    np.random.seed(5)
    ref.build_refiner(f0, None, "probabilistic")
    np.testing.assert_array_equal(ref.indices_batch, want)               # drawn from the global RNG like collaborator.py:56
    never = want == 3
    assert np.all(ref.optimal_step.cpu().numpy()[never] == 1)
    with pytest.raises(NotImplementedError):
        ref.build_refiner(f0, None, "greedy")
    with pytest.raises(lib.CgsError):
        gan.engine(4).refine(f0, 2, 0.1, method="ladam")
    ops.reset_variables()


@pytest.mark.parametrize("arch,B,K", [("dcgan32", 256, 3), ("dcgan64", 128, 2)])
def test_full_size_properties(arch, B, K):
    """BASELINE-shaped nets at sizes the CPU oracle would not finish quickly: properties that hold at any size."""
    from cgs_amd.engine import RefineEngine
    from cgs_amd.nets import to_device, ARCHS
    d = dev()
    P = to_device(N.init_params(arch, 2019, True), d)
    eng = RefineEngine(arch, P, B, d)
    z = torch.from_numpy(np.random.RandomState(2019).uniform(-1, 1, (B, ARCHS[arch]["z_dim"])).astype(np.float32)).to(d)
    f0 = eng.input_to_feature(z).clone()
    img, dl, ol, os_, of = [t.clone() for t in eng.refine(f0, K, 0.1)]
    assert torch.isfinite(img).all() and img.abs().max() <= 1.0           # tanh range
    assert (ol >= dl).all()                                                # the best logit never decreases
    assert ((os_ >= 1) & (os_ <= K)).all()
    unchanged = ol == dl
    assert torch.equal(of[unchanged], f0[unchanged])                       # never improved -> theta0 kept (step stays 1, Q1)
    img2, dl2, ol2, os2, of2 = eng.refine(f0, K, 0.1)
    assert torch.equal(img, img2) and torch.equal(ol, ol2) and torch.equal(os_, os2)   # deterministic, bit-identical
    img0 = eng.refine(f0, 0, 0.1)[0].clone()                               # K = 0: plain rendering of the proposal
    assert torch.equal(img0, eng.feature_to_data(f0))
    idx = np.full(B, K)                                                    # probabilistic index K => never selected (Q2)
    _, _, _, osp, ofp = eng.refine(f0, K, 0.1, mode="probabilistic", indices=idx)
    assert torch.equal(ofp, f0) and (osp == 1).all()
    idx0 = np.zeros(B, dtype=np.int64)                                     # index 0 => exactly one update applied
    _, _, _, osp, ofp = eng.refine(f0, K, 0.1, mode="probabilistic", indices=idx0)
    lm, grad = eng.compute_forward_logits_and_grad(f0)
    assert (osp == 1).all() and torch.allclose(ofp, f0 - 0.1 * grad, rtol=0, atol=1e-6)


@pytest.mark.parametrize("arch,B,K", [("dcgan64", 1024, 2), ("dcgan32", 256, 4)])
def test_two_batches_in_flight_match_sequential(arch, B, K):
    """bench.py's default mode at BASELINE's full batch: two engines on two HIP streams refine concurrently (shared frozen
    weights, per-stream packed-weight caches and workspaces).  Each must reproduce, bit for bit, what one engine computes
    alone -- any cross-stream hazard (shared scratch, weight re-packing under a running kernel) would show up here."""
    from cgs_amd.engine import RefineEngine
    from cgs_amd.nets import to_device, ARCHS
    d = dev()
    P = to_device(N.init_params(arch, 2019, True), d)
    rs = np.random.RandomState(5)
    z = [torch.from_numpy(rs.uniform(-1, 1, (B, ARCHS[arch]["z_dim"])).astype(np.float32)).to(d) for _ in range(2)]
    solo = RefineEngine(arch, P, B, d)
    want = [[t.clone() for t in solo.refine_from_z(zi, K, 0.1)[:4]] for zi in z]
    engines = [RefineEngine(arch, P, B, d) for _ in range(2)]
    streams = [torch.cuda.Stream(d) for _ in range(2)]
    torch.cuda.synchronize(d)
    for rep in range(2):                      # second round: caches warm, both streams busy from the first launch
        got = []
        for e, st, zi in zip(engines, streams, z):
            with torch.cuda.stream(st):
                got.append(e.refine_from_z(zi, K, 0.1)[:4])
        torch.cuda.synchronize(d)
        for g, w in zip(got, want):
            for a, b in zip(g, w):
                assert torch.equal(a, b)


@pytest.mark.parametrize("arch,B,G,K", [("mnist", 8, 3, 3), ("dcgan32", 16, 2, 2)])
def test_fused_logical_batches_keep_their_own_statistics(arch, B, G, K):
    """RefineEngine(bn_groups=G): G logical batches in one engine batch, D's batch norm per logical batch -> the same
    result as G separate calls (the convolutions only see a larger batch)."""
    from cgs_amd.engine import RefineEngine
    from cgs_amd.nets import to_device, ARCHS
    d = dev()
    P = to_device(N.init_params(arch, 2019, True), d)
    z = torch.from_numpy(np.random.RandomState(9).uniform(-1, 1, (G * B, ARCHS[arch]["z_dim"])).astype(np.float32)).to(d)
    solo = RefineEngine(arch, P, B, d)
    want = [[t.clone() for t in solo.refine_from_z(z[i * B:(i + 1) * B], K, 0.1)] for i in range(G)]
    fused = RefineEngine(arch, P, G * B, d, bn_groups=G)
    img, dl, ol, st, of = fused.refine_from_z(z, K, 0.1)
    cat = lambda j: torch.cat([w[j] for w in want])
    assert torch.allclose(dl, cat(1), rtol=1e-4, atol=1e-5)
    assert torch.allclose(ol, cat(2), rtol=2e-3, atol=2e-4)
    assert (st == cat(3)).float().mean().item() >= 0.9
    assert torch.allclose(of, cat(4), rtol=0, atol=2e-3 * cat(4).abs().max().item())
    assert torch.allclose(img, cat(0), rtol=0, atol=5e-3)
    plain = RefineEngine(arch, P, G * B, d)                          # one batch of G*B: different statistics, different logits
    assert (plain.refine_from_z(z, K, 0.1)[1] - cat(1)).abs().max().item() > 10 * (dl - cat(1)).abs().max().item()


def test_collaborative_fill_loop_on_device():
    """refine -> D-score -> MH accept -> fill (nsgan/GAN.py:398-426) driven by the device engine."""
    from cgs_amd.engine import RefineEngine
    from cgs_amd.evaluate import collaborate, engine_proposer, MIN_EFFICIENCY
    from cgs_amd.nets import to_device
    from cgs_amd.sampling import IndependenceSampler
    d = dev()
    P = N.init_params("mnist", 2019, True)
    eng = RefineEngine("mnist", to_device(P, d), 32, d)
    propose, score = engine_proposer(eng, 3, 0.1, rng=np.random.RandomState(5))
    batch = propose()
    assert batch.shape == (32, 28, 28, 1) and np.abs(batch).max() <= 1.0
    sig = score(batch)
    with torch.no_grad():
        want = torch.sigmoid(N.discriminator("mnist", P, torch.from_numpy(batch))).numpy()
    np.testing.assert_allclose(sig, want, rtol=0, atol=2e-5)                     # scoring = D on batch statistics
    np.random.seed(3)
    out, eff = collaborate(propose, score, IndependenceSampler(T=20), 40, float(sig.mean()), min_efficiency=MIN_EFFICIENCY)
    assert out.shape == (40, 28, 28, 1) and 0.0 < eff <= 1.0 and np.isfinite(out).all()
