"""Checkpoint import / export in the reference's variable-name space.

The reference saves with ``tf.train.Saver`` (nsgan/GAN.py:149,465-491: ``checkpoint/GAN_mnist_64_62/GAN/model-<step>``;
synthetic/main.py:294-295,394-395).  Variable names come from the scopes in nsgan/ops.py:38-43,49-61,75-79 and
nsgan/GAN.py:62-99 -- ``discriminator/d_conv1/w``, ``generator/g_bn3/moving_variance``, ``discriminator/d_fc3/Matrix``,
... -- and the tensor layouts are the ones this package uses natively (HWIO conv, [kh,kw,Cout,Cin] deconv, [in,out]
linear), so a checkpoint is just a flat ``{name: array}`` map.  Here that map is stored as ``.safetensors`` (default)
or ``.npz``; ``tools/tf_ckpt_to_safetensors.py`` produces it from a TF1 checkpoint wherever TensorFlow is installed
(TF names carry a ``:0`` suffix and optimizer slots such as ``/Adam``; both are dropped).
"""
import os
import re

import numpy as np

_SLOT = re.compile(r"/(Adam(_\d+)?|Momentum|RMSProp(_\d+)?)$|^(beta\d_power|global_step)")


def clean_tf_names(tensors):
    """Drop ``:0`` suffixes and optimizer slot variables from a TF variable dump."""
    out = {}
    for k, v in tensors.items():
        k = k[:-2] if k.endswith(":0") else k
        if _SLOT.search(k):
            continue
        out[k] = np.asarray(v, dtype=np.float32)
    return out


def save(path, params):
    """``params``: {tf_variable_name: tensor/ndarray}.  Format by extension (.safetensors / .npz)."""
    arrs = {k: np.ascontiguousarray(v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v), dtype=np.float32)
            for k, v in params.items()}
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    if path.endswith(".npz"):
        np.savez(path, **arrs)
    else:
        from safetensors.numpy import save_file
        save_file(arrs, path, metadata={"format": "cgs_amd/tf-variable-names", "layouts": "conv HWIO; deconv [kh,kw,Cout,Cin]; linear [in,out]"})


def load(path):
    """-> {name: float32 ndarray}"""
    if path.endswith(".npz"):
        with np.load(path) as z:
            return {k: z[k].astype(np.float32) for k in z.files}
    from safetensors.numpy import load_file
    return {k: v.astype(np.float32) for k, v in load_file(path).items()}


def check_against_arch(params, arch):
    """Raise with a precise message if ``params`` does not fit the layer lists of ``arch`` (names and shapes)."""
    from .nets import param_shapes
    want = param_shapes(arch)
    missing = sorted(set(want) - set(params))
    if missing:
        raise KeyError(f"checkpoint lacks {len(missing)} variables of arch {arch!r}, e.g. {missing[:4]}")
    for k, shp in want.items():
        if tuple(params[k].shape) != tuple(shp):
            raise ValueError(f"{k}: checkpoint shape {tuple(params[k].shape)} != {tuple(shp)} expected by arch {arch!r}")
    return True
