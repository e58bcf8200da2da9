#!/usr/bin/env python3
"""Time the REFERENCE's own refiners (imported from /root/reference behind the test-only facade of make_golden.py)
on this container's host cores, next to the oracle port.  Build container only - it needs /root/reference.

    python tests/golden/time_reference.py          # prints one JSON object; numbers quoted in DESIGN.md section 6
"""
import json
import os
import sys
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G          # noqa: E402  (sets up the facade and imports the reference modules)
import numpy as np               # noqa: E402
import torch                     # noqa: E402

N, S, T = G.N, G.S, G.T


def time_collab(arch, B, K, reps=2):
    P = N.init_params(arch, seed=2019, perturb=True)
    A = N.ARCHS[arch]
    rs = np.random.RandomState(1)
    z = rs.uniform(-1, 1, (B, A["z_dim"])).astype(np.float32)
    real = torch.zeros((B,) + tuple(A["img"]))
    disc = lambda x: T(N.discriminator(arch, P, T.un(x)))
    g_tail = lambda f: T(N.feature_to_data(arch, P, T.un(f)))
    loss = lambda l: T(torch.nn.functional.softplus(-T.un(l)))
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        with torch.no_grad():
            feat0 = N.input_to_feature(arch, P, torch.from_numpy(z))
        ref = G.collaborator.Refiner(rollout_steps=K, rollout_rate=0.1)
        ref.set_env(disc, g_tail, loss)
        ref.build_refiner(T(feat0), T(real), "deterministic")
        best = min(best, time.perf_counter() - t0)
    return B / best


def time_refiner_cpu(B=512, K=10, reps=5):
    Ws, bs = S.mlp_init(64, 6, seed=2019, scale=2.0)

    class Gan:
        fake_samples, fake_sigmoid, fake_saliency = "fake_samples", "fake_sigmoid", "fake_saliency"

    class Sess:
        def run(self, fetches, feed_dict):
            sig, sal = S.mlp_sigmoid_and_saliency(Ws, bs, feed_dict[Gan.fake_samples])
            return [{"fake_sigmoid": sig, "fake_saliency": sal}[f] for f in fetches]

    args = types.SimpleNamespace(rollout_steps=K, rollout_rate=0.1, rollout_method="ladam")
    data = G.Datasets.ToyDataset(distr="Imbal-8Gaussians", scale=10.0, ratio=0.9)
    fake = (3.0 * np.random.RandomState(7).randn(B, 2)).astype(np.float32)
    best = 1e30
    for _ in range(reps):
        ref = G.refiner_cpu.Refiner(args)
        ref.set_env(Gan, Sess(), data)
        t0 = time.perf_counter()
        ref.manipulate_sample(fake.copy(), "deterministic")
        best = min(best, time.perf_counter() - t0)
    return B / best


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    out = {"cores": os.cpu_count(), "unit": "refined samples/s",
           "refiner_cpu cfg1 B=512 K=10 ladam": round(time_refiner_cpu(), 1),
           "collaborator mnist B=64 K=50": round(time_collab("mnist", 64, 50), 2),
           "collaborator dcgan32 B=32 K=20": round(time_collab("dcgan32", 32, 20), 2),
           "collaborator dcgan64 B=16 K=20": round(time_collab("dcgan64", 16, 20, reps=1), 3)}
    print(json.dumps(out))
