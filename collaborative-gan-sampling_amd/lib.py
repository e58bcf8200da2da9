"""ctypes binding of libcgs_hip.so -- the C ABI declared in include/cgs_hip.h.

The library is built in-tree by ``__graft_entry__.build()`` (or ``make -C csrc``).  Loading is
lazy; if the shared object is missing every device op fails loudly (no fallback path exists).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CGS_LIB") or os.path.join(_HERE, "libcgs_hip.so")      # CGS_LIB: an experimental build (development aid)

OK, EINVAL, EWORKSPACE, ELAUNCH = 0, -1, -2, -3
EPI_NONE, EPI_LRELU, EPI_AFFINE_RELU, EPI_TANH = 0, 1, 2, 3
EPI_RELU_BWD_AFFINE, EPI_LRELU_BWD, EPI_TANH_BWD = 4, 5, 6
CONV_FWD, CONV_BWD_DATA, DECONV_FWD, DECONV_BWD_DATA = 0, 1, 2, 3
FAMILY_IGEMM, FAMILY_QUAD, FAMILY_SMALLN_T, FAMILY_SMALLN_F, FAMILY_PATCH, FAMILY_TAPS, FAMILY_DOT, FAMILY_IGEMM_BX6 = 0, 1, 2, 3, 4, 5, 6, 7
CONTRACTION_F32, CONTRACTION_BX6, CONTRACTION_BX6_ALL = 0, 1, 2
CONTRACTIONS = {"f32": CONTRACTION_F32, "bx6": CONTRACTION_BX6, "bx6_all": CONTRACTION_BX6_ALL}

_p, _i, _f, _z = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_ll = C.c_longlong

# name -> (restype, argtypes); mirrors include/cgs_hip.h one to one
SIGNATURES = {
    "cgs_version": (_i, []),
    "cgs_source_sha": (C.c_char_p, []),
    "cgs_last_error": (C.c_char_p, []),
    "cgs_last_kernel": (C.c_char_p, []),
    "cgs_last_executed_flops": (C.c_double, []),
    "cgs_last_tail_tiles": (_i, []),
    "cgs_last_tail_split": (_i, []),
    "cgs_set_contraction": (_i, [_i]),
    "cgs_get_contraction": (_i, []),
    "cgs_conv_ws_bytes": (_z, [_i] * 7),
    "cgs_conv_ws_bytes_for": (_z, [_i] * 10),
    "cgs_conv_family": (_i, [_i] * 13 + [_z]),
    "cgs_conv2d_nhwc_fwd": (_i, [_p] * 4 + [_i] * 9 + [_i, _p, _p, _p, _z, _i, _p]),
    "cgs_conv_stat_partials": (_i, [_i] * 9 + [_z]),
    "cgs_conv2d_nhwc_fwd_stats": (_i, [_p] * 4 + [_i] * 9 + [_p, _z, _i, _p, _z, _p]),
    "cgs_conv_stat_layout": (_i, [_i] * 13 + [_z, _p, _p, _p]),
    "cgs_deconv2d_nhwc_fwd_stats": (_i, [_p] * 4 + [_i] * 11 + [_p, _z, _i, _p, _z, _p]),
    "cgs_conv2d_nhwc_bwd_data_nstats": (_i, [_p] * 3 + [_i] * 9 + [_p] * 5 + [_f, _i, _p, _z, _i, _p, _z, _p]),
    "cgs_deconv2d_nhwc_bwd_data_nstats": (_i, [_p] * 3 + [_i] * 11 + [_p] * 5 + [_f, _i, _p, _z, _i, _p, _z, _p]),
    "cgs_norm_lrelu_bwd_from_partials": (_i, [_p] * 3 + [_i] * 4 + [_p] * 4 + [_f, _p, _i, _i, _p, _z, _p]),
    "cgs_groupnorm_lrelu_fwd_from_partials": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _f, _f, _p, _p, _p, _i, _i, _p, _z, _p]),
    "cgs_bn_train_lrelu_fwd_from_partials": (_i, [_p, _p, _i, _p, _p, _f, _f, _p, _p, _p, _i, _i, _p, _z, _p]),
    "cgs_conv2d_nhwc_bwd_data": (_i, [_p] * 3 + [_i] * 9 + [_i, _p, _p, _p, _z, _i, _p]),
    "cgs_deconv2d_nhwc_fwd": (_i, [_p] * 4 + [_i] * 11 + [_i, _p, _p, _p, _z, _i, _p]),
    "cgs_deconv2d_nhwc_bwd_data": (_i, [_p] * 3 + [_i] * 11 + [_i, _p, _p, _p, _z, _i, _p]),
    "cgs_conv_signs_ok": (_i, [_i] * 13 + [_z]),
    "cgs_deconv2d_nhwc_fwd_signs": (_i, [_p] * 4 + [_i] * 11 + [_i, _p, _p, _p, _p, _z, _i, _p]),
    "cgs_deconv2d_nhwc_bwd_data_signs": (_i, [_p] * 3 + [_i] * 11 + [_i, _p, _p, _p, _z, _i, _p]),
    "cgs_linear_fwd": (_i, [_p] * 4 + [_i] * 4 + [_p, _z, _i, _p]),
    "cgs_linear_bwd_data": (_i, [_p] * 3 + [_i] * 3 + [_p, _z, _i, _p]),
    "cgs_bn_ws_bytes": (_z, [_i, _i]),
    "cgs_bn_train_lrelu_fwd": (_i, [_p, _p, _p, _f, _f, _p, _p, _p, _i, _i, _p, _z, _p]),
    "cgs_bn_train_lrelu_bwd_data": (_i, [_p] * 6 + [_f, _p, _i, _i, _p, _z, _p]),
    "cgs_bn_sync_fwd_sums": (_i, [_p, _p, _i, _i, _p, _z, _p]),
    "cgs_bn_sync_fwd_apply": (_i, [_p, _p, _p, _f, _f, _p, _ll, _p, _p, _p, _i, _i, _p, _z, _p]),
    "cgs_bn_sync_bwd_sums": (_i, [_p] * 6 + [_f, _p, _i, _i, _p, _z, _p]),
    "cgs_bn_sync_bwd_apply": (_i, [_p] * 6 + [_f, _p, _ll, _p, _i, _i, _p, _z, _p]),
    "cgs_instnorm_ws_bytes": (_z, [_i, _i, _i]),
    "cgs_instnorm_lrelu_fwd": (_i, [_p, _p, _p, _f, _f, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "cgs_instnorm_lrelu_bwd_data": (_i, [_p] * 6 + [_f, _p, _i, _i, _i, _p, _z, _p]),
    "cgs_add": (_i, [_p, _p, _p, _z, _p]),
    "cgs_bn_fold": (_i, [_p] * 4 + [_f, _p, _p, _i, _p]),
    "cgs_affine_relu_fwd": (_i, [_p] * 4 + [_i, _i, _p]),
    "cgs_affine_relu_bwd": (_i, [_p] * 4 + [_i, _i, _p]),
    "cgs_affine_fwd": (_i, [_p] * 4 + [_i, _i, _p]),
    "cgs_affine_bwd": (_i, [_p] * 3 + [_i, _i, _p]),
    "cgs_lrelu_fwd": (_i, [_p, _f, _p, _z, _p]),
    "cgs_lrelu_bwd": (_i, [_p, _p, _f, _p, _z, _p]),
    "cgs_tanh_fwd": (_i, [_p, _p, _z, _p]),
    "cgs_tanh_bwd": (_i, [_p, _p, _p, _z, _p]),
    "cgs_bce_ones_grad_rowmean": (_i, [_p, _p, _p, _i, _i, _p]),
    "cgs_bce_ones_fwd": (_i, [_p, _p, _z, _p]),
    "cgs_bce_ones_bwd": (_i, [_p, _p, _p, _z, _p]),
    "cgs_sigmoid_rowmean": (_i, [_p, _p, _i, _i, _p]),
    "cgs_clip": (_i, [_p, _f, _f, _p, _z, _p]),
    "cgs_refine_update": (_i, [_p, _p, _p, _f, _f, _i, _i, _f, _f, _z, _p]),
    "cgs_refine_select": (_i, [_p, _p, _p, _i, _p, _p, _p, _i, _i, _p]),
    "cgs_refine_select_rows": (_i, [_p, _p, _p, _i, _p, _p, _i, _i, _p]),
    "cgs_refine_select2": (_i, [_p, _p, _i, _p, _p, _i, _p, _p, _i, _p, _p, _p, _i, _p]),
    "cgs_linear_out1_bce": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "cgs_mlp2d_sigmoid_saliency": (_i, [_p, _p, _i, _i, _p, _p, _p, _i, _f, _p]),
    "cgs_refine2d": (_i, [_p, _p, _i, _i, _p, _f, _f, _i, _f, _i, _p, _p, _p, _i, _p]),
    "cgs_refine2d_devbase": (_i, [_p, _p, _i, _i, _p, _p, _f, _i, _f, _i, _p, _p, _p, _i, _p]),
    "cgs_mlp2d_train_ws_bytes": (_z, [_i, _i]),
    "cgs_mlp2d_d_step": (_i, [_p, _p, _i, _i, _p, _i, _p, _i, _f, _p, _p, _p, _p, _z, _p]),
    "cgs_conv_wgrad_ws_bytes": (_z, [_i] * 9),
    "cgs_conv2d_nhwc_bwd_weight": (_i, [_p] * 3 + [_i] * 9 + [_i, _p, _z, _p]),
    "cgs_linear_bwd_weight": (_i, [_p] * 3 + [_i] * 3 + [_i, _p, _z, _p]),
    "cgs_bias_grad": (_i, [_p, _p, _i, _i, _i, _p, _z, _p]),
    "cgs_bn_train_param_grads": (_i, [_p, _i, _i, _p, _p, _i, _p]),
    "cgs_bce_logits_grad": (_i, [_p, _f, _f, _p, _p, _i, _p]),
    "cgs_adam_step": (_i, [_p, _p, _p, _p, _f, _f, _f, _f, _z, _p]),
}

_lib = None
stale = None          # after load(): False = the library's embedded source stamp equals the sources beside it; True = it differs (only
                      # tolerated under CGS_LIB / CGS_ALLOW_STALE); None = no sources beside the library to compare with (a binary-only install)


class CgsError(RuntimeError):
    pass


class GraphCaptureError(RuntimeError):
    """hipGraph capture of an engine's K-step program was refused (raised around the capture block only: argument errors and
    failures of the eager warm-up stay what they are).  Callers may fall back to eager launches of the same engine."""


def load():
    """Load libcgs_hip.so (once) and type its entry points.  Raises if it is absent."""
    global _lib
    if _lib is None:
        # PyTorch hands this library its device pointers and HIP streams, so both must live on ONE HIP runtime instance: torch's
        # wheel bundles its own libamdhip64, and whichever copy is mapped first serves the SONAME for the whole process.  Loading
        # this library before torch would bind it (and then torch) to the system runtime: "no ROCm-capable device" at the first
        # launch (seen with `python __graft_entry__.py smoke`, which builds / loads before it imports torch).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise CgsError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the device path)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the ABI and this table diverge
            fn.restype, fn.argtypes = res, args
        _check_stamp(lib)
        _lib = lib
    return _lib


class StaleLibraryError(CgsError):
    """The prebuilt library was built from other kernel sources than the ones shipped beside it."""


def built_from():
    """The sha256 of the sources the LOADED library was built from (its embedded stamp, ``cgs_source_sha``)."""
    return load().cgs_source_sha().decode()


def _check_stamp(lib, here=None):
    """Compare the library's embedded source stamp with the sources found beside it.  A mismatch means the git-ignored prebuilt .so
    is not the build of this tree (every number measured on it would be filed under the wrong sources): refused, unless the run
    points at another build on purpose (CGS_LIB: experiment builds; CGS_ALLOW_STALE=1) -- then ``lib.stale`` records it and bench.py
    stamps it into its line."""
    global stale
    embedded = lib.cgs_source_sha().decode()
    have = source_hash(here)
    if have is None:
        stale = None
        return
    stale = embedded != have
    if stale and not (os.environ.get("CGS_LIB") or os.environ.get("CGS_ALLOW_STALE")):
        raise StaleLibraryError(f"{LIB_PATH} was built from sources {embedded[:16]}..., the tree beside it hashes to {have[:16]}...: rebuild it "
                                "(`make -C collaborative-gan-sampling_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`); "
                                "CGS_ALLOW_STALE=1 loads it anyway")


def call(name, *args):
    """Call an int-returning entry point; raise CgsError with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != OK:
        raise CgsError(f"{name} failed ({rc}): {lib.cgs_last_error().decode()}")


def set_contraction(mode):
    """The calling thread's contraction arithmetic (include/cgs_hip.h): "f32" (exact fp32 MFMA, the default), "bx6" (opt-in:
    the big layers through split-bf16 MFMA), "bx6_all" (test coverage); returns the previous mode's name."""
    lib = load()
    code = CONTRACTIONS[mode] if isinstance(mode, str) else int(mode)
    prev = int(lib.cgs_get_contraction())
    call("cgs_set_contraction", code)
    return [k for k, v in CONTRACTIONS.items() if v == prev][0]


def get_contraction():
    code = int(load().cgs_get_contraction())
    return [k for k, v in CONTRACTIONS.items() if v == code][0]


def last_kernel():
    """Name of the kernel the last conv / deconv / linear dispatch launched (as rocprofv3 prints it)."""
    return load().cgs_last_kernel().decode()


def conv_ws_bytes(op, kh, kw, sh, sw, cin, cout):
    return int(load().cgs_conv_ws_bytes(op, kh, kw, sh, sw, cin, cout))


def conv_ws_bytes_for(op, b, h, w, cin, cout, kh, kw, sh, sw):
    return int(load().cgs_conv_ws_bytes_for(op, b, h, w, cin, cout, kh, kw, sh, sw))


def bn_ws_bytes(m, c):
    return int(load().cgs_bn_ws_bytes(m, c))


def source_hash(here=None):
    """sha256 over the kernel sources beside the library (csrc/*.hip, *.h and include/cgs_hip.h; the Makefile embeds the same hash
    into the library it builds: ``built_from``): identifies the code measured numbers such as profiles/traffic.json belong to.
    None when there are no sources (a binary-only install)."""
    import glob
    import hashlib
    here = here or _HERE
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.h")))
    header = os.path.join(os.path.dirname(here), "include", "cgs_hip.h")
    if not files or not os.path.exists(header):
        return None
    for f in files + [header]:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()
