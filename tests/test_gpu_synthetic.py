"""BASELINE config 1 on the GPU: the MLP discriminator kernel and the fused 2-D refiner against the golden vectors
captured from the reference's refiner_cpu.Refiner (tests/golden/g1_refiner_cpu.npz) and the oracle."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden
from oracle import sampling_ref as S


def traj_close(out, want):
    """K ladam steps divide by sqrt(v)+1e-8 with tiny saliencies: fp32 differences of the D evaluation are amplified on a
    few samples.  Bulk tight, tail bounded."""
    err = np.abs(np.asarray(out, dtype=np.float64) - want)
    assert np.percentile(err, 99) < 1e-4 and err.max() < 5e-3, (np.percentile(err, 99), err.max())


def _D():
    from cgs_amd.synthetic import MLPDiscriminator
    g = load_golden("g1_refiner_cpu.npz")
    return g, MLPDiscriminator.from_lists(list(g["W"]), list(g["b"]), "cuda:0")


def test_mlp_sigmoid_and_saliency_vs_oracle():
    g, D = _D()
    Ws, bs = [torch.from_numpy(w) for w in g["W"]], [torch.from_numpy(b) for b in g["b"]]
    for x in (g["fake"], S.toy_next_batch("Imbal-8Gaussians", 10.0, 0.9, 333, np.random.RandomState(1)).astype(np.float32)):
        sig, sal = D.sigmoid_and_saliency(x)
        want_sig, want_sal = S.mlp_sigmoid_and_saliency(Ws, bs, x)
        np.testing.assert_allclose(sig.cpu().numpy(), want_sig, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(sal.cpu().numpy(), want_sal, rtol=1e-3, atol=1e-8)


def test_reference_host_loop_drives_gpu_discriminator():
    """sampling.refiner_cpu.Refiner (the reference-shaped host loop) with the GPU Session adaptor reproduces the
    reference's own output for config 1 (12 sess.run calls)."""
    from cgs_amd.datasets import ToyDataset
    from cgs_amd.sampling import refiner_cpu
    from cgs_amd.synthetic import Gan, Session
    g, D = _D()
    gan = Gan(D)
    args = types.SimpleNamespace(rollout_steps=10, rollout_rate=0.1, rollout_method="ladam")
    for mode in ("deterministic", "probabilistic"):
        sess = Session(gan)
        ref = refiner_cpu.Refiner(args)
        ref.set_env(gan, sess, ToyDataset("Imbal-8Gaussians", 10.0, 0.9))
        np.random.seed(2019)
        out = ref.manipulate_sample(g["fake"].copy(), mode)
        assert sess.n_runs == 12 and str(out.dtype) == str(g[mode + "_dtype"][0])
        traj_close(out, g[mode])


def test_fused_refiner_matches_reference_golden():
    """synthetic.Refiner: the whole K-step ladam loop in one launch, against the reference's output."""
    from cgs_amd.datasets import ToyDataset
    from cgs_amd.synthetic import Gan, Refiner
    g, D = _D()
    args = types.SimpleNamespace(rollout_steps=10, rollout_rate=0.1, rollout_method="ladam")
    for mode in ("deterministic", "probabilistic"):
        ref = Refiner(args)
        ref.set_env(Gan(D), None, ToyDataset("Imbal-8Gaussians", 10.0, 0.9))
        np.random.seed(2019)
        fake = g["fake"].copy()
        out = ref.manipulate_sample(fake, mode)
        np.testing.assert_array_equal(fake, g["fake"])
        assert str(out.dtype) == str(g[mode + "_dtype"][0])
        traj_close(out, g[mode])
    with pytest.raises(NotImplementedError):
        ref.manipulate_sample(g["fake"].copy(), "greedy")


@pytest.mark.parametrize("method", ["sgd", "momentum", "ladam"])
def test_fused_refiner_methods_vs_oracle_large_batch(method):
    g, D = _D()
    Ws, bs = [torch.from_numpy(w) for w in g["W"]], [torch.from_numpy(b) for b in g["b"]]
    rs = np.random.RandomState(3)
    fake = (3.0 * rs.randn(4096, 2)).astype(np.float32)
    real = S.toy_next_batch("Imbal-8Gaussians", 10.0, 0.9, 4096, rs)
    d_fn = lambda x: S.mlp_sigmoid_and_saliency(Ws, bs, x)
    # the saliency carries 1/B (quirk Q8): plain sgd / momentum need a rate ~B to move at all, ladam normalises it away
    rate = {"sgd": 200.0, "momentum": 60.0, "ladam": 0.05}[method]
    want, want_step, _ = S.refine_2d(fake, real, d_fn, 5, rate, method, "deterministic")
    base = float(np.mean(d_fn(real)[0]))
    best, step, traj = D.refine(fake, base, 5, rate, method, want_traj=True)
    agree = step.cpu().numpy() == want_step
    assert agree.mean() > 0.98                                       # best-step flips only on numerical ties
    traj_close(best.cpu().numpy()[agree], want[agree])
    np.testing.assert_allclose(traj.cpu().numpy()[:, 0, :], fake, rtol=0, atol=0)


def test_device_baseline_equals_host_baseline():
    """cgs_refine2d_devbase: the real batch's mean sigmoid passed as a device scalar gives the bits of the host-float call."""
    import numpy as np
    import torch
    from cgs_amd.synthetic import MLPDiscriminator
    from oracle import sampling_ref as S
    d = torch.device("cuda:0")
    Ws, bs = S.mlp_init(64, 6, seed=2019, scale=2.0)
    D = MLPDiscriminator.from_lists([w.numpy() for w in Ws], [b.numpy() for b in bs], d)
    x = torch.from_numpy((3.0 * np.random.RandomState(3).randn(512, 2)).astype(np.float32)).to(d)
    real = torch.from_numpy(np.random.RandomState(4).randn(512, 2).astype(np.float32)).to(d)
    base = D.sigmoid_and_saliency(real, want_saliency=False)[0].mean()
    a = D.refine(x, base, 10, 0.1, "ladam")
    b = D.refine(x, float(base.item()), 10, 0.1, "ladam")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_mlp_d_step_gradients_and_sgd_vs_autograd():
    """cgs_mlp2d_d_step: every D-variable gradient of d_loss (synthetic/GAN.py:69-74) against torch autograd on the oracle MLP,
    the two loss terms, the in-place GradientDescentOptimizer update, determinism, ragged batch sizes."""
    from cgs_amd.synthetic import DShaper, MLPDiscriminator
    rs = np.random.RandomState(5)
    for nl, nh, Br, Bf in ((6, 64, 1000, 1000), (3, 33, 77, 130), (2, 64, 5, 3)):
        Ws, bs = S.mlp_init(nh, nl, seed=3, scale=2.0)
        real = S.toy_next_batch("Imbal-8Gaussians", 10.0, 0.9, max(Br, 64), rs)[:Br].astype(np.float32)
        fake = (3.0 * rs.randn(Bf, 2)).astype(np.float32)
        D = MLPDiscriminator.from_lists([w.numpy() for w in Ws], [b.numpy() for b in bs], "cuda:0")
        sh = DShaper(D, lrd=8e-3)
        losses, gW, gb = S.mlp_d_loss_and_grads(Ws, bs, real, fake)
        loss, dW, db = sh.loss_and_grads(real, fake)
        np.testing.assert_allclose(loss.cpu().numpy(), losses, rtol=2e-5)
        for got, want in zip(dW + db, gW + gb):
            scale = float(want.abs().max()) + 1e-12
            assert float((got.cpu() - want).abs().max()) <= 2e-4 * scale + 1e-9, (nl, nh, float((got.cpu() - want).abs().max()), scale)
        for t, w in zip(D.w + D.b, Ws + bs):
            assert torch.equal(t.cpu(), w)                                           # lr = 0 left the weights alone
        g_first = [t.clone() for t in dW + db]
        before = [t.clone() for t in D.w + D.b]
        sh.step(real, fake)
        for t, w0, g in zip(D.w + D.b, before, g_first):
            assert torch.equal(t, w0 - torch.tensor(8e-3, dtype=torch.float32) * g)  # var -= lr*grad, two roundings, bit-exact
        D2 = MLPDiscriminator.from_lists([w.numpy() for w in Ws], [b.numpy() for b in bs], "cuda:0")
        sh2 = DShaper(D2, lrd=8e-3)
        sh2.step(real, fake)
        for a, b in zip(D.w + D.b, D2.w + D2.b):
            assert torch.equal(a, b)                                                 # deterministic


def test_shaping_iteration_matches_reference_golden():
    """synthetic/main.py:366-370 on the device: fused probabilistic refine -> D SGD step -> the refiner on the shaped D,
    against the iteration captured with the reference's refiner_cpu.Refiner (tests/golden/g10_shape2d.npz)."""
    from cgs_amd.datasets import ToyDataset
    from cgs_amd.synthetic import DShaper, Gan, MLPDiscriminator, Refiner, shape_step
    g = load_golden("g10_shape2d.npz")
    K, lrd = int(g["K"][0]), float(g["lrd"][0])
    D = MLPDiscriminator.from_lists(list(g["W0"]), list(g["b0"]), "cuda:0")
    data = ToyDataset("Imbal-8Gaussians", 10.0, 0.9)
    ref = Refiner(types.SimpleNamespace(rollout_steps=K, rollout_rate=0.1, rollout_method="ladam"))
    ref.set_env(Gan(D), None, data)
    sh = DShaper(D, lrd=lrd)
    np.random.seed(2019)
    real_batch = data.next_batch(len(g["noise_sample"]))
    np.testing.assert_array_equal(real_batch, g["real_batch"])
    loss, refined = shape_step(ref, sh, g["noise_sample"], real_batch)
    assert refined.dtype == np.float64
    traj_close(refined, g["refined"])
    # the D step itself on exactly the reference's refined batch (so trajectory noise does not enter the weight comparison)
    D2 = MLPDiscriminator.from_lists(list(g["W0"]), list(g["b0"]), "cuda:0")
    loss2 = DShaper(D2, lrd=lrd).step(real_batch, g["refined"])
    np.testing.assert_allclose(loss2.cpu().numpy(), g["d_loss"], rtol=2e-5)
    for t, want, w0 in zip(D2.w + D2.b, list(g["W1"]) + list(g["b1"]), list(g["W0"]) + list(g["b0"])):
        step = np.abs(want - w0).max()
        assert np.abs(t.cpu().numpy() - want).max() <= 1e-3 * step + 1e-9           # the update agrees to 0.1 % of its own size
    # ... and the refiner then sees the shaped D (synthetic/main.py:217 after :370)
    ref2 = Refiner(types.SimpleNamespace(rollout_steps=K, rollout_rate=0.1, rollout_method="ladam"))
    ref2.set_env(Gan(D2), None, data)
    np.random.seed(7)
    traj_close(ref2.manipulate_sample(g["eval_batch"].copy()), g["refined_after"])
    np.testing.assert_allclose(loss.cpu().numpy(), g["d_loss"], rtol=1e-2)          # same iteration end to end (device-refined batch)


def test_config1_end_to_end_collaborative_evaluation():
    """synthetic/main.py:215-263 on the device refiner: refine -> D-score -> MH fill -> good-rate / KL / JS.  D is first
    trained on the device (3800 "calibrate" iterations against a fixed poor generator -- the modes blurred by a 0.8-sigma
    Gaussian -- then 200 "shape" iterations on refined batches, synthetic/main.py:350-370, lrd 1e-2) so that its gradient
    points at the modes; then refinement must raise the good-sample rate, and the collaborative sample must beat both
    (the ordering of README.md:26-28; the same scenario on the CPU oracle gives good 0.22 -> 0.31 -> 0.44, JS 0.51 -> 0.45 -> 0.26)."""
    from cgs_amd.datasets import ToyDataset
    from cgs_amd.synthetic import DShaper, Gan, MLPDiscriminator, Refiner, evaluate_collaborative, shape_step
    Ws, bs = S.mlp_init(64, 6, seed=2019, scale=1.0)
    D = MLPDiscriminator.from_lists([w.numpy() for w in Ws], [b.numpy() for b in bs], "cuda:0")
    data = ToyDataset("Imbal-8Gaussians", 10.0, 0.9)
    gen_rs = np.random.RandomState(3)
    generate = lambda n=1000: (data.centeroids[gen_rs.randint(8, size=n)] + 0.8 * gen_rs.randn(n, 2)).astype(np.float32)
    args = types.SimpleNamespace(rollout_steps=20, rollout_rate=0.1, rollout_method="ladam")
    ref = Refiner(args)
    ref.set_env(Gan(D), None, data)
    sh = DShaper(D, lrd=1e-2)
    np.random.seed(2019)
    for it in range(4000):
        real = data.next_batch(1000)
        fake = generate()
        if it < 3800:
            sh.step(real, fake)                         # mode "calibrate": D on (real, generated), main.py:361-364
        else:
            shape_step(ref, sh, fake, real)             # mode "shape": D on (real, refined), main.py:366-370
    out = evaluate_collaborative(ref, D, lambda: generate(2000), generate(2000), data.next_batch(2000), data.centeroids, data.std)
    print("config-1 end to end:", out)
    assert out["refinement"]["good"] > out["standard"]["good"] + 0.03
    assert out["collaborate"]["good"] > out["refinement"]["good"] and 0.0 < out["collaborate"]["eff"] <= 1.0
    assert out["collaborate"]["js"] < out["refinement"]["js"] < out["standard"]["js"]
