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
