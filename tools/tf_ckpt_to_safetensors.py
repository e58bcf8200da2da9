#!/usr/bin/env python3
"""Convert a TF1 ``tf.train.Saver`` checkpoint of the reference (nsgan/GAN.py:465-471) into the flat
name -> tensor file ``cgs_amd.checkpoint.load`` reads.  Needs TensorFlow, so it runs wherever the reference does
(TF is not installable in the build image of this repo).

    python tools/tf_ckpt_to_safetensors.py checkpoint/GAN_mnist_64_62/GAN/model-5000 mnist_5000.safetensors [--arch mnist]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("ckpt")
    ap.add_argument("out")
    ap.add_argument("--arch", default=None, help="validate names/shapes against a cgs_amd arch (mnist, dcgan32, dcgan64)")
    a = ap.parse_args()
    try:
        import tensorflow as tf
    except ImportError:
        raise SystemExit("TensorFlow is required to read a TF checkpoint; run this where the reference runs")
    reader = tf.train.load_checkpoint(a.ckpt)
    from cgs_amd import checkpoint as C
    params = C.clean_tf_names({n: reader.get_tensor(n) for n in reader.get_variable_to_shape_map()})
    if a.arch:
        C.check_against_arch(params, a.arch)
    C.save(a.out, params)
    print(f"wrote {len(params)} variables to {a.out}")


if __name__ == "__main__":
    main()
