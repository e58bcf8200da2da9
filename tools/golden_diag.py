#!/usr/bin/env python3
"""Measured distances of the fused engine from every g3 golden (the reference's own collaborator.Refiner): what the
tolerances of tests/test_gpu_refine.py::check_against_golden stand on.  GPU box:  python tools/golden_diag.py"""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN, golden_feature0        # noqa: E402
from oracle import nets_ref as N                    # noqa: E402
from cgs_amd.engine import RefineEngine             # noqa: E402
from cgs_amd.nets import to_device                  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


d = torch.device("cuda:0")
for path in sorted(glob.glob(os.path.join(GOLDEN, "g3_collab_*.npz"))):
    g = np.load(path, allow_pickle=True)
    arch, mode = str(g["arch"][0]), str(g["mode"][0])
    P = N.init_params(arch, seed=2019, perturb=True)
    c = g["constraints"]
    vmin, vmax = (None, None) if np.isnan(c[0]) else (float(c[0]), float(c[1]))
    for graph in (False, True):
        eng = RefineEngine(arch, to_device(P, d), len(g["z"]), d, use_graph=graph)
        f0 = torch.from_numpy(golden_feature0(g, arch, P)).to(d)
        for _ in range(2 if graph else 1):
            img, dl, ol, st, of = [t.cpu().numpy() for t in eng.refine(f0, int(g["K"][0]), float(g["rate"][0]), "momentum", mode,
                                                                         g["indices"] if mode == "probabilistic" else None, vmin, vmax)]
        ok = st == g["optimal_step"]
        print(f"{os.path.basename(path)[10:-4]:42s} {'graph' if graph else 'eager'} B={len(ok):3d} step agree {ok.mean():.4f} "
              f"default {rel(dl, g['default_logit']):.1e} logit {rel(ol, g['optimal_logit']):.1e} feature {rel(of[ok], g['optimal_feature'][ok]):.1e} "
              f"image {rel(img[ok], g['images'][ok]):.1e}", flush=True)
