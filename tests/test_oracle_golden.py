"""Pin the oracle (oracle/) against the golden vectors captured from the
reference's own classes (tests/golden/make_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_feature0, load_golden
from oracle import nets_ref as N
from oracle import sampling_ref as S


def test_g1_refiner_cpu_matches_reference():
    g = load_golden("g1_refiner_cpu.npz")
    Ws = [torch.from_numpy(w) for w in g["W"]]
    bs = [torch.from_numpy(b) for b in g["b"]]
    Ws2, _ = S.mlp_init(64, 6, seed=int(g["mlp_seed"][0]), scale=float(g["mlp_scale"][0]))
    assert all(torch.equal(a, b) for a, b in zip(Ws, Ws2))          # seeded init is reproducible
    d_fn = lambda x: S.mlp_sigmoid_and_saliency(Ws, bs, x)
    fake = g["fake"]
    for mode in ("deterministic", "probabilistic"):
        np.random.seed(2019)
        real = S.toy_next_batch("Imbal-8Gaussians", 10.0, 0.9, len(fake))
        idx = None
        if mode == "probabilistic":
            # the reference draws the indices AFTER the loop (refiner_cpu.py:72); nothing else
            # touches the global RNG in between.
            idx = np.random.randint(10 + 1, size=len(fake))
        out, step, calls = S.refine_2d(fake, real, d_fn, 10, 0.1, "ladam", mode, idx)
        assert calls == int(g[mode + "_calls"][0]) == 12
        assert str(out.dtype) == str(g[mode + "_dtype"][0])
        np.testing.assert_allclose(out, g[mode], rtol=0, atol=1e-6)
    assert str(g["deterministic_dtype"][0]) == "float32" and str(g["probabilistic_dtype"][0]) == "float64"


def test_g2_policy_traces():
    g = load_golden("g2_policy.npz")
    for method in ("sgd", "momentum", "ladam"):
        p = S.Policy(0.1, method)
        th = g["theta0"].copy()
        for i in range(3):
            th = p.step(th, g["grads"][i], g["losses"][i]).astype(np.float32)
            np.testing.assert_allclose(th, g[method][i], rtol=1e-6, atol=1e-7)
    p = S.Policy(0.5, "momentum")
    th = torch.from_numpy(g["theta0_map"])
    for i in range(3):
        th = p.step(th, torch.from_numpy(g["grads_map"][i]))
        np.testing.assert_allclose(th.numpy(), g["momentum_map"][i], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "g3_collab_*.npz"))),
                         ids=lambda p: os.path.basename(p)[3:-4])
def test_g3_collaborator_matches_reference(path):
    g = np.load(path, allow_pickle=True)
    arch, K, mode = str(g["arch"][0]), int(g["K"][0]), str(g["mode"][0])
    P = N.init_params(arch, seed=2019, perturb=True)
    chk = float(sum(v.double().abs().sum() for v in P.values()))
    assert abs(chk - float(g["params_checksum"][0])) <= 1e-9 * chk   # same weights as at capture time
    with torch.no_grad():
        f0 = N.input_to_feature(arch, P, torch.from_numpy(g["z"]))
    feature0 = golden_feature0(g, arch, P)
    np.testing.assert_allclose(f0.numpy(), feature0, rtol=1e-5, atol=1e-6)
    c = g["constraints"]
    vmin, vmax = (None, None) if np.isnan(c[0]) else (float(c[0]), float(c[1]))
    img, dl, ol, os_, of = S.collaborative_refine(
        torch.from_numpy(feature0), lambda f: N.feature_to_data(arch, P, f),
        lambda x: N.discriminator(arch, P, x), K, float(g["rate"][0]), "momentum", mode,
        indices=g["indices"] if mode == "probabilistic" else None, vmin=vmin, vmax=vmax)
    np.testing.assert_allclose(dl.numpy(), g["default_logit"], rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(os_.numpy(), g["optimal_step"])
    np.testing.assert_allclose(ol.numpy(), g["optimal_logit"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(of.numpy(), g["optimal_feature"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(img.numpy(), g["images"], rtol=1e-3, atol=1e-4)
    if mode == "probabilistic":          # quirk Q2: index K => never selected => step stays 1, feature stays theta0
        never = g["indices"] == K
        assert np.all(g["optimal_step"][never] == 1)
        np.testing.assert_array_equal(of.numpy()[never], feature0[never])


def test_g6_rejector_mask_bit_exact():
    g = load_golden("g6_rejector.npz")
    for tag, pct in (("p60", 60.0), ("p100", 100.0), ("none", None)):
        rej = S.RejectorRef()
        np.random.seed(2019)
        for c in range(3):
            mask, _ = rej.accept_mask(g[f"{tag}_sig{c}"], shift_percent=pct)
            np.testing.assert_array_equal(mask, g[f"{tag}_mask{c}"])
            assert rej.D_tilde_M == g[f"{tag}_M{c}"][0]           # bit-exact float64 state
    rej = S.RejectorRef(); rej.set_score_max(np.array(0.93, dtype=np.float32))
    assert rej.D_tilde_M == g["set_score_max_M"][0]


def test_g7_mh_chain_exact():
    g = load_golden("g7_mh.npz")
    mh = S.IndependenceSamplerRef(T=20)
    mh.d_curr = 0.4
    np.random.seed(2019)
    for c in range(2):
        idx = mh.accepted_indices(g[f"sig{c}"])
        np.testing.assert_array_equal(np.array(idx, dtype=np.int64), g[f"accepted{c}"])
        assert str(g[f"dtype{c}"][0]) == "float32"


def test_g8_toy_dataset_exact():
    g = load_golden("g8_toy.npz")
    for distr, ratio, B in (("Imbal-8Gaussians", 0.9, 512), ("8Gaussians", 0.5, 100), ("25Gaussians", 0.5, 60)):
        np.random.seed(2019)
        np.testing.assert_array_equal(S.toy_next_batch(distr, 10.0, ratio, B), g[distr])


def test_g10_shape2d_iteration():
    """The 2-D D-shaping iteration (synthetic/main.py:366-370): the oracle's refine_2d reproduces what the reference's Refiner
    produced before and after the D step, and its restated GradientDescentOptimizer step reproduces the stored weights."""
    g = load_golden("g10_shape2d.npz")
    B, K, lrd = len(g["noise_sample"]), int(g["K"][0]), float(g["lrd"][0])
    W0, b0 = [torch.from_numpy(w) for w in g["W0"]], [torch.from_numpy(b) for b in g["b0"]]
    np.random.seed(2019)
    real_batch = S.toy_next_batch("Imbal-8Gaussians", 10.0, 0.9, B)
    np.testing.assert_array_equal(real_batch, g["real_batch"])
    inner_real = S.toy_next_batch("Imbal-8Gaussians", 10.0, 0.9, B)               # refiner_cpu.py:22
    idx = np.random.randint(K + 1, size=B)                                        # refiner_cpu.py:72
    refined, _, _ = S.refine_2d(g["noise_sample"], inner_real, lambda x: S.mlp_sigmoid_and_saliency(W0, b0, x), K, 0.1, "ladam",
                                "probabilistic", idx)
    np.testing.assert_allclose(refined, g["refined"], rtol=0, atol=1e-6)
    W1, b1, losses = S.mlp_d_sgd_step(W0, b0, real_batch.astype(np.float32), g["refined"].astype(np.float32), lrd)
    np.testing.assert_allclose(losses, g["d_loss"], rtol=1e-6)
    for a, b in zip(W1 + b1, list(g["W1"]) + list(g["b1"])):
        np.testing.assert_allclose(a.numpy(), b, rtol=1e-6, atol=1e-8)
    assert max(float((a - torch.from_numpy(b)).abs().max()) for a, b in zip(W0, g["W1"])) > 1e-5      # the step moved D
    np.random.seed(7)
    inner_real = S.toy_next_batch("Imbal-8Gaussians", 10.0, 0.9, len(g["eval_batch"]))
    after, _, _ = S.refine_2d(g["eval_batch"], inner_real, lambda x: S.mlp_sigmoid_and_saliency(W1, b1, x), K, 0.1, "ladam")
    np.testing.assert_allclose(after, g["refined_after"], rtol=0, atol=1e-6)
