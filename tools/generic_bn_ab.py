#!/usr/bin/env python3
"""A/B of the moving-average update inside ops.bn on the generic (ops + autograd) path, in one process (development aid, GPU box):
the nine-launch tensor-expression form against the four-launch foreach form.   python tools/generic_bn_ab.py [arch]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cgs_amd import kernels as K, nets, ops
from cgs_amd.model import GAN
from cgs_amd.sampling.collaborator import Refiner

arch = sys.argv[1] if len(sys.argv) > 1 else "mnist"
dev = torch.device("cuda:0")
B, Ksteps = 64, 50 if arch == "mnist" else 20
new_bn = ops.bn


def old_bn(x, is_training, scope, leak=1.0):
    if not is_training:
        return new_bn(x, is_training, scope, leak)
    C = x.shape[-1]
    with ops.variable_scope(scope):
        beta = ops.get_variable("beta", [C], ops.constant_initializer(0.0))
        gamma = ops.get_variable("gamma", [C], ops.constant_initializer(1.0))
        mm = ops.get_variable("moving_mean", [C], ops.constant_initializer(0.0))
        mv = ops.get_variable("moving_variance", [C], ops.constant_initializer(1.0))
    y, mean, invstd = ops._BnTrain.apply(x, gamma, beta, float(leak))
    with torch.no_grad():
        mm.mul_(0.9).add_(0.1 * mean)
        mv.mul_(0.9).add_(0.1 * (1.0 / (invstd * invstd) - K.BN_EPS))
    return y


ops.reset_variables()
gan = GAN(arch, batch_size=B, device=dev, params=nets.init_params(arch, dev, seed=2019))
z = torch.from_numpy(np.random.RandomState(0).uniform(-1, 1, (B,) + nets.g_input_shape(nets.ARCHS[arch])).astype(np.float32)).to(dev)
warnings.simplefilter("ignore", RuntimeWarning)
r = Refiner(Ksteps, 0.1)
r.set_env(lambda x: gan.discriminator(x, is_training=True, reuse=True), gan.feature_to_data,
          lambda l: ops.sigmoid_cross_entropy_with_logits(logits=l, labels=ops.ones_like(l)))
with torch.no_grad():
    f = gan.input_to_feature(z)
    for rep in range(3):
        for label, fn in (("nine launches", old_bn), ("four launches", new_bn)):
            ops.bn = fn
            r.build_refiner(f, None, "deterministic")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                r.build_refiner(f, None, "deterministic")
            torch.cuda.synchronize()
            print(f"{arch} B={B} K={Ksteps} generic path, moving-average update in {label}: {(time.perf_counter() - t0) / 4 * 1e3:7.2f} ms per call ({r.path})")
ops.bn = new_bn
