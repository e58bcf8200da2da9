"""Distribution-level equivalence of the refined sample POOL: the metric's second half is "FID vs reference".
No Inception features / datasets / checkpoints exist here, so the stand-in is the same construction one level down: the Frechet
distance (cgs_amd.metrics.frechet_distance, the formula behind the reference's mnist_frechet_distance, nsgan/utils_mnist.py:85-116)
between pools of refined samples in a FIXED feature space -- the oracle discriminator's penultimate activations -- plus the two
acceptance statistics the reference's samplers derive from a pool (Rejector accept rate, MH-chain efficiency).

Pools on the same z: unrefined G(z); the CPU oracle's refinement (the pinned restatement of the reference); the engine on the exact-fp32
contraction; the engine on the opt-in split-bf16 contraction (every eligible layer: "bx6_all").  Per-sample parity already bounds the
f32 pool; the bx6 pool is the one whose K-step trajectories may differ on ties -- what must hold is that the POOL does not move:
FD(engine, oracle) << FD(refined, unrefined), and the acceptance statistics agree."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nets_ref as N
from oracle import sampling_ref as S

RESULTS = {}


def _features(arch, P, images, b):
    """Oracle D up to (not including) its last linear layer, batch statistics per logical batch of b; conv maps mean-pooled over space."""
    layers = N.ARCHS[arch]["d"]
    cut = max(i for i, L in enumerate(layers) if L[0] == "linear")
    body = [L for L in layers[:cut] if L[0] != "flatten"] if arch != "mnist" else layers[:cut]
    feats, sig = [], []
    with torch.no_grad():
        for i in range(0, len(images), b):
            x = images[i:i + b]
            f = N.run_layers(body, x, P, "discriminator", bn_training=True)
            feats.append(f.mean(dim=(1, 2)) if f.dim() == 4 else f[:, :256])      # (a fixed 256-coordinate marginal of the 1024-d fc features: n >> d)
            sig.append(torch.sigmoid(N.discriminator(arch, P, x)).reshape(len(x), -1).mean(1, keepdim=True))
    return torch.cat(feats).double().numpy(), torch.cat(sig).numpy()


def _acceptance(images, sig, b):
    """(Rejector accept rate, MH efficiency) of a pool walked batch by batch with seeded global RNG (rejector.py:16-38, idpsampler.py:17-53)."""
    from cgs_amd.sampling import IndependenceSampler, Rejector
    np.random.seed(99)
    rej, kept = Rejector(), 0
    for i in range(0, len(images), b):
        kept += len(rej.sampling(images[i:i + b], sig[i:i + b]))
    np.random.seed(99)
    mh = IndependenceSampler(T=20)
    mh.set_score_curr(float(sig.mean()))
    acc = sum(len(mh.walk(sig[i:i + b])) for i in range(0, len(images), b))
    return kept / len(images), acc / len(images)


@pytest.mark.parametrize("arch,b,G,K", [("dcgan32", 256, 4, 20), ("mnist", 64, 16, 50)])      # (1024-sample pools: the oracle's refinement is the cost)
def test_refined_pool_is_the_oracles_pool_in_feature_space(arch, b, G, K):
    from cgs_amd.engine import RefineEngine
    from cgs_amd.metrics import frechet_distance
    from cgs_amd.nets import to_device
    d = torch.device("cuda:0")
    torch.set_num_threads(min(16, torch.get_num_threads()))             # (the oracle on the host: 16 threads beat 128 on these layer sizes, bench.cpu_baseline)
    P = N.init_params(arch, 2019, True)
    n = b * G
    z = torch.from_numpy(np.random.RandomState(2019).uniform(-1, 1, (n, N.ARCHS[arch]["z_dim"])).astype(np.float32))
    gt, dd = (lambda f: N.feature_to_data(arch, P, f)), (lambda x: N.discriminator(arch, P, x))
    pools = {}
    with torch.no_grad():
        f0 = N.input_to_feature(arch, P, z)
        pools["unrefined"] = N.feature_to_data(arch, P, f0)
    pools["oracle"] = torch.cat([S.collaborative_refine(f0[i:i + b], gt, dd, K, 0.1)[0] for i in range(0, n, b)])
    Pd = to_device(P, d)
    for mode in ("f32", "bx6_all"):
        eng = RefineEngine(arch, Pd, n, d, use_graph=True, bn_groups=G, contraction=mode)
        pools[mode] = eng.refine_from_z(z.to(d), K, 0.1)[0].cpu()
        del eng
    feat, sig, acc = {}, {}, {}
    for k, img in pools.items():
        feat[k], sig[k] = _features(arch, P, img, b)
        acc[k] = _acceptance(img.numpy(), sig[k], b)
    fd_move = frechet_distance(feat["oracle"], feat["unrefined"])
    fd = {k: frechet_distance(feat["oracle"], feat[k]) for k in ("f32", "bx6_all")}
    fd_modes = frechet_distance(feat["f32"], feat["bx6_all"])
    RESULTS[arch] = dict(fd_refined_vs_unrefined=fd_move, fd_f32_vs_oracle=fd["f32"], fd_bx6_vs_oracle=fd["bx6_all"], fd_f32_vs_bx6=fd_modes,
                         acceptance=acc, mean_sigmoid={k: float(v.mean()) for k, v in sig.items()})
    print(f"\n[distribution {arch} {G}x{b} K={K}] FD(refined, unrefined) = {fd_move:.6g}; FD(f32, oracle) = {fd['f32']:.3g}; "
          f"FD(bx6, oracle) = {fd['bx6_all']:.3g}; FD(f32, bx6) = {fd_modes:.3g}; (Rejector rate, MH efficiency): "
          + ", ".join(f"{k} ({a:.4f}, {e:.4f})" for k, (a, e) in acc.items()))
    assert fd_move > 0
    for k in ("f32", "bx6_all"):
        # the pool does not move: at most 1 % of what refinement itself moves it (measured: 1e-5 ... 1e-3 of it)
        assert abs(fd[k]) < 1e-2 * fd_move, (k, fd[k], fd_move)
        # acceptance statistics of the pool: the Rejector's rate and the MH efficiency within half a percent of the pool size
        assert abs(acc[k][0] - acc["oracle"][0]) <= 5e-3 and abs(acc[k][1] - acc["oracle"][1]) <= 5e-3, (k, acc)
        assert abs(float(sig[k].mean()) - float(sig["oracle"].mean())) < 1e-3
    assert abs(fd_modes) < 1e-2 * fd_move
    # and refinement did something the samplers can see: the refined pool scores higher than the proposals
    assert float(sig["oracle"].mean()) > float(sig["unrefined"].mean())
