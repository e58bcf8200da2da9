"""Tensor-level wrappers over the C ABI (include/cgs_hip.h): torch CUDA(ROCm) tensors in, tensors out.

PyTorch is plumbing only here -- device memory (``torch.empty``), the current HIP stream, and
nothing else: every FLOP below runs in libcgs_hip.so.  Activations are NHWC fp32 contiguous; weight
layouts are the reference's (nsgan/ops.py:39,51,76).  Packed-weight workspaces are cached per
(weight storage, version, op): frozen weights are packed once.
"""
import math
import weakref

import torch

from . import lib as L

LEAK = 0.2      # nsgan/ops.py:69
BN_EPS = 1e-5   # nsgan/ops.py:23


def _chk(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise L.CgsError(f"{name}: expected a contiguous fp32 tensor on the GPU, got "
                         f"{type(t).__name__} {getattr(t, 'dtype', None)} {getattr(t, 'device', None)}")
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ---- optional per-kernel timing (bench.py): {name: [flops, [(ev_start, ev_end), ...], executed_flops, algorithmic_bytes]} or None.
# Events are recorded on the stream the kernel is launched on (torch's current stream).  Conv-family records are keyed by the
# kernel instantiation the dispatcher launched (cgs_last_kernel); the HBM-bound multi-kernel ops (batch norm passes, the
# momentum update) by their entry point.
PROFILE = None


PROFILE_BY_LAYER = False      # key the records by kernel + layer shape instead of kernel only


class _Prof:
    __slots__ = ("name", "flops", "e0", "nbytes", "op")

    def __init__(self, flops, tag="", nbytes=0.0, op=None):
        self.name, self.flops, self.e0, self.nbytes, self.op = tag, flops, None, nbytes, op
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def done(self):
        if self.e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            if self.op is not None:
                kname, executed = self.op, self.flops
            else:
                kname = L.load().cgs_last_kernel().decode()        # the instantiation the dispatcher actually launched
                executed = float(L.load().cgs_last_executed_flops()) or self.flops     # minus the skipped zero-padding taps
            rec = PROFILE.setdefault(kname + " " + self.name if PROFILE_BY_LAYER and self.name else kname, [0.0, [], 0.0, 0.0])
            rec[0] += self.flops
            rec[1].append((self.e0, e1))
            rec[2] += executed
            rec[3] += self.nbytes


def _nb(*tensors):
    """Algorithmic HBM bytes of a launch: every operand tensor once."""
    return 4.0 * sum(t.numel() for t in tensors if t is not None)


def same_out(size, stride):
    """nsgan/ops.py:28-29."""
    return int(math.ceil(float(size) / float(stride)))


# ---- contraction arithmetic of the implicit-GEMM layers (include/cgs_hip.h, cgs_set_contraction): "f32" = exact fp32 MFMA (default),
# "bx6" = opt-in split-bf16 MFMA for the big layers, "bx6_all" = the same for every eligible call (test coverage).  The library keeps
# the mode per host thread; this module mirrors the mode of the thread that drives it (one host thread per process here).
CONTRACTION = "f32"     # the mode this module last put in force (informational; the thread's real state is asked of the library)


def set_contraction(mode):
    """Switch the CALLING THREAD's contraction mode; returns the previous one.  The library keeps the mode per host thread
    (a second host thread starts in "f32" whatever another thread set), so the compare is against the thread's own state."""
    global CONTRACTION
    prev = L.get_contraction()
    if mode != prev:
        L.set_contraction(mode)
    CONTRACTION = mode
    return prev


class contraction:
    """``with contraction("bx6"): ...`` -- the mode for the block, the previous mode restored after it (generic ops.* calls,
    DShaper steps and other engines of the thread do not inherit an engine's opt-in)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = set_contraction(self.mode)
        return self

    def __exit__(self, *exc):
        set_contraction(self.prev)
        return False


class _WsCache:
    """Packed-weight workspaces, keyed by the weight tensor (identity + version), the op, the call geometry, the HIP
    stream AND the kernel family the library will run (``cgs_conv_family``): every family keeps its own packed layout,
    and which one serves a call depends on the fused epilogue too, so the family is part of the key -- a workspace is
    only ever reported as pre-packed to the family that packed it."""

    def __init__(self):
        self._d = {}
        self._family = {}
        self.epoch = 0        # bumped by invalidate(): engines compare it with the epoch their packed copies / folded affines belong to
        self.clears = 0       # bumped by clear()
        self.record = None    # a list: every weight tensor asked for is appended as a weak reference (the generic path's hipGraph capture
                              # learns which weights its recorded program reads, sampling/collaborator.py::_GenericGraph)

    def plan(self, op, B, H, W, cin, ho, wo, cout, kh, kw, sh, sw, epilogue):
        """(kernel family, workspace bytes) of a call; asked of the library once per distinct call signature."""
        # (the family depends on the calling thread's contraction mode: asked of the library itself, so a caller that switched it
        # through lib.set_contraction / the C ABI directly is keyed correctly too)
        key = (op, B, H, W, cin, ho, wo, cout, kh, kw, sh, sw, epilogue, int(L.load().cgs_get_contraction()))
        hit = self._family.get(key)
        if hit is None:
            nbytes = max(L.conv_ws_bytes_for(op, B, H, W, cin, cout, kh, kw, sh, sw), 16)
            f = int(L.load().cgs_conv_family(op, B, H, W, cin, ho, wo, cout, kh, kw, sh, sw, epilogue, nbytes))
            if f < 0:
                raise L.CgsError(f"cgs_conv_family failed ({f}): {L.load().cgs_last_error().decode()}")
            hit = self._family[key] = (f, nbytes)
        return hit

    def get(self, w, op, kh, kw, sh, sw, cin, cout, bhw=(0, 0, 0), epilogue=0, out_hw=(0, 0), ptrs=()):
        """``bhw`` = (B, H, W) of the call: sizes the optional split-K slab area behind the packed weights.
        ``ptrs``: every device pointer of the call; if one is not 16-byte aligned the library may pick another family than
        the planned one, so such calls get a scratch workspace and always re-pack."""
        fam, nbytes = self.plan(op, bhw[0], bhw[1], bhw[2], cin, out_hw[0], out_hw[1], cout, kh, kw, sh, sw, epilogue)
        if self.record is not None:
            self.record.append(weakref.ref(w))
        aligned = not any(p is not None and (p & 15) for p in ptrs)
        if not aligned:
            fam = -1
        # one packed copy per HIP stream: the pack kernel is ordered only with later work of the stream that ran it
        key = (w.data_ptr(), op, kh, kw, sh, sw, cin, cout, w.device.index, bhw, fam, torch.cuda.current_stream(w.device).cuda_stream)
        hit = self._d.get(key)
        if hit is not None and hit[0]() is w:
            if hit[1] == w._version and aligned:
                return hit[2], 1
            self._d[key] = (hit[0], w._version, hit[2])      # stale: re-pack into the SAME buffer (hipGraphs keep its address)
            return hit[2], 0
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
        if len(self._d) > 512:
            # only entries whose weight tensor is gone: a captured hipGraph holds the raw address of its workspaces
            # (ws_prepacked = 1), so an entry of a live weight is never dropped behind an engine's back
            for k in [k for k, v in self._d.items() if v[0]() is None]:
                del self._d[k]
        self._d[key] = (weakref.ref(w), w._version, ws)
        return ws, 0

    def invalidate(self):
        """Weights were updated in place by a kernel torch does not see (the Adam step, a checkpoint restore): every packed
        copy is stale.  Eager calls re-pack on their next use; every RefineEngine notices the new epoch at its next call and
        re-folds its inference-bn affines and re-packs the workspaces its captured hipGraphs read (engine._resync)."""
        for key, (ref, _, ws) in list(self._d.items()):
            self._d[key] = (ref, -1, ws)
        self.epoch += 1

    def clear(self):
        self._d.clear()
        self.clears += 1      # (the buffers a captured generic-path hipGraph points into are gone: it compares this counter)


WS = _WsCache()
_bn_ws = {}


def _bn_workspace(M, C, device):
    key = (C, device.index, torch.cuda.current_stream(device).cuda_stream)     # one scratch area per stream

    ws = _bn_ws.get(key)
    if ws is None:
        ws = torch.empty(L.bn_ws_bytes(M, C) // 4, dtype=torch.float32, device=device)
        _bn_ws[key] = ws
    return ws


# ----------------------------------------------------------------------------- conv family
def conv2d_fwd(x, w, bias, sh=2, sw=2, epilogue=L.EPI_NONE, ep_a=None, ep_b=None, out=None):
    """tf.nn.conv2d 'SAME' + bias (+ fused epilogue).  nsgan/ops.py:41-44."""
    _chk(x, "x"); _chk(w, "w")
    B, H, W, Cin = x.shape
    kh, kw, Cin2, Cout = w.shape
    if Cin2 != Cin:
        raise L.CgsError(f"conv2d: weight Cin {Cin2} != input channels {Cin}")
    y = out if out is not None else torch.empty((B, same_out(H, sh), same_out(W, sw), Cout), dtype=torch.float32, device=x.device)
    ws, pre = WS.get(w, L.CONV_FWD, kh, kw, sh, sw, Cin, Cout, (B, H, W), epilogue,
                     ptrs=(_ptr(x), _ptr(bias), _ptr(y), _ptr(ep_a), _ptr(ep_b)))
    Ho, Wo = y.shape[1], y.shape[2]
    pr = _Prof(2.0 * B * Ho * Wo * Cout * kh * kw * Cin, f"conv_fwd {H}x{W} {Cin}->{Cout}", _nb(x, w, y)) if PROFILE is not None else None
    L.call("cgs_conv2d_nhwc_fwd", _ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, H, W, Cin, Cout, kh, kw, sh, sw,
           epilogue, _ptr(ep_a), _ptr(ep_b), _ptr(ws), ws.numel() * 4, pre, _stream())
    if pr is not None:
        pr.done()
    return y


def conv_stat_partials(x_shape, w_shape, sh=2, sw=2):
    """Partial rows G that ``conv2d_fwd_stats`` writes for this call (0: the fused statistics are not available for it)."""
    B, H, W, Cin = x_shape
    kh, kw, _, Cout = w_shape
    _, nbytes = WS.plan(L.CONV_FWD, B, H, W, Cin, 0, 0, Cout, kh, kw, sh, sw, L.EPI_NONE)
    return int(L.load().cgs_conv_stat_partials(B, H, W, Cin, Cout, kh, kw, sh, sw, nbytes))


def conv_stat_layout(op, B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, group_images):
    """Where the partial rows of every group of ``group_images`` consecutive images lie in the buffer ``conv2d_fwd_stats`` /
    ``deconv2d_fwd(part=...)`` fill: (rows_total, rows_per_seg, nseg, seg_stride), or None when the fused statistics are not
    available for the call (include/cgs_hip.h, cgs_conv_stat_layout)."""
    import ctypes as C
    _, nbytes = WS.plan(op, B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, L.EPI_NONE)
    a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
    rows = int(L.load().cgs_conv_stat_layout(op, B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, int(group_images), nbytes,
                                             C.addressof(a), C.addressof(b), C.addressof(c)))
    return (rows, a.value, b.value, c.value) if rows > 0 else None


def groupnorm_lrelu_fwd_from_partials(x, part, layout, groups, scale, offset, leak=1.0, eps=BN_EPS, out=None, stats=None):
    """Norm over ``groups`` groups of consecutive rows of x ([groups * M_group, C] flattened; instance norm: one group per sample)
    whose statistics the producing convolution left in ``part`` (``layout`` = conv_stat_layout(...) of that call)."""
    _chk(x, "x"); _chk(part, "part")
    C_ = x.shape[-1]
    Mg = x.numel() // (groups * C_)
    y = out if out is not None else torch.empty_like(x)
    mean, invstd = stats if stats is not None else (torch.empty((groups, C_), dtype=torch.float32, device=x.device),
                                                     torch.empty((groups, C_), dtype=torch.float32, device=x.device))
    ws = _in_workspace(groups, Mg, C_, x.device)
    pr = _Prof(0.0, "", _nb(x, y), op="groupnorm_lrelu_fwd_from_partials") if PROFILE is not None else None
    L.call("cgs_groupnorm_lrelu_fwd_from_partials", _ptr(x), _ptr(part), groups, layout[1], layout[2], layout[3], _ptr(scale), _ptr(offset),
           eps, leak, _ptr(y), _ptr(mean), _ptr(invstd), Mg, C_, _ptr(ws), ws.numel() * 4, _stream())
    if pr is not None:
        pr.done()
    return y, mean, invstd


def conv2d_fwd_stats(x, w, bias, part, sh=2, sw=2, out=None):
    """conv2d_fwd (no epilogue) that also leaves the per-block column sums / sums of squares of its output in ``part``
    [G, 2, Cout] for the batch norm that follows (``bn_train_lrelu_fwd_from_partials``)."""
    _chk(x, "x"); _chk(w, "w"); _chk(part, "part")
    B, H, W, Cin = x.shape
    kh, kw, Cin2, Cout = w.shape
    if Cin2 != Cin:
        raise L.CgsError(f"conv2d: weight Cin {Cin2} != input channels {Cin}")
    y = out if out is not None else torch.empty((B, same_out(H, sh), same_out(W, sw), Cout), dtype=torch.float32, device=x.device)
    ws, pre = WS.get(w, L.CONV_FWD, kh, kw, sh, sw, Cin, Cout, (B, H, W), L.EPI_NONE, ptrs=(_ptr(x), _ptr(bias), _ptr(y), _ptr(part)))
    Ho, Wo = y.shape[1], y.shape[2]
    pr = _Prof(2.0 * B * Ho * Wo * Cout * kh * kw * Cin, f"conv_fwd {H}x{W} {Cin}->{Cout}", _nb(x, w, y)) if PROFILE is not None else None
    L.call("cgs_conv2d_nhwc_fwd_stats", _ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, H, W, Cin, Cout, kh, kw, sh, sw,
           _ptr(ws), ws.numel() * 4, pre, _ptr(part), part.numel() * 4, _stream())
    if pr is not None:
        pr.done()
    return y


def bn_train_lrelu_fwd_from_partials(x, part, gamma, beta, leak=LEAK, eps=BN_EPS, out=None, stats=None):
    """bn_train_lrelu_fwd whose statistics pass is replaced by the partial sums of the producing conv."""
    _chk(x, "x")
    C = x.shape[-1]
    M = x.numel() // C
    y = out if out is not None else torch.empty_like(x)
    mean, invstd = stats if stats is not None else (torch.empty(C, dtype=torch.float32, device=x.device),
                                                    torch.empty(C, dtype=torch.float32, device=x.device))
    ws = _bn_workspace(M, C, x.device)
    pr = _Prof(0.0, "", _nb(x, y), op="bn_train_lrelu_fwd_from_partials") if PROFILE is not None else None     # read x, write y
    L.call("cgs_bn_train_lrelu_fwd_from_partials", _ptr(x), _ptr(part), part.shape[0], _ptr(gamma), _ptr(beta), eps, leak, _ptr(y),
           _ptr(mean), _ptr(invstd), M, C, _ptr(ws), ws.numel() * 4, _stream())
    if pr is not None:
        pr.done()
    return y, mean, invstd


class NormBwdStats:
    """What a backward-data call needs to ALSO leave the two column sums of the norm (+ lrelu) whose output gradient it produces
    (include/cgs_hip.h, cgs_*_bwd_data_nstats): the norm's input ``x`` (same shape as the call's result), its saved ``mean`` / ``invstd``
    ([groups, C]), ``gamma`` / ``beta``, the lrelu ``leak``, the images per statistics group, and the partial-row buffer ``part``
    ([rows, 2, C]) with its ``layout`` = conv_stat_layout(<the *_BWD_DATA op>, ...).  ``norm_lrelu_bwd_from_partials`` consumes it."""
    __slots__ = ("x", "mean", "invstd", "gamma", "beta", "leak", "group_images", "part", "layout")

    def __init__(self, x, mean, invstd, gamma, beta, leak, group_images, part, layout):
        self.x, self.mean, self.invstd, self.gamma, self.beta = x, mean, invstd, gamma, beta
        self.leak, self.group_images, self.part, self.layout = float(leak), int(group_images), part, layout

    def args(self):
        return (_ptr(self.x), _ptr(self.mean), _ptr(self.invstd), _ptr(self.gamma), _ptr(self.beta), self.leak, self.group_images)


def conv2d_bwd_data(dy, w, in_hw, sh=2, sw=2, out=None, epilogue=L.EPI_NONE, ep_a=None, ep_aux=None, nstat=None):
    """Input gradient of conv2d_fwd (Conv2DBackpropInput; sampling/collaborator.py:31).
    ``epilogue`` = one of the *_BWD modes folds the activation gradient of the layer below in.
    ``nstat`` (a NormBwdStats; no epilogue): the result is the gradient at the output of a norm -- leave that norm's backward sums too."""
    _chk(dy, "dy"); _chk(w, "w")
    kh, kw, Cin, Cout = w.shape
    B = dy.shape[0]
    H, W = in_hw
    dx = out if out is not None else torch.empty((B, H, W, Cin), dtype=torch.float32, device=dy.device)
    ws, pre = WS.get(w, L.CONV_BWD_DATA, kh, kw, sh, sw, Cin, Cout, (B, H, W), epilogue,
                     ptrs=(_ptr(dy), _ptr(dx), _ptr(ep_a), _ptr(ep_aux)))
    Ho, Wo = dy.shape[1], dy.shape[2]
    pr = _Prof(2.0 * B * Ho * Wo * Cout * kh * kw * Cin, f"conv_bwd {H}x{W} {Cin}<-{Cout}",
               _nb(dy, w, dx, ep_aux, nstat.x if nstat is not None else None)) if PROFILE is not None else None
    if nstat is not None:
        if epilogue != L.EPI_NONE:
            raise L.CgsError("conv2d_bwd_data: norm-backward statistics come with no epilogue")
        L.call("cgs_conv2d_nhwc_bwd_data_nstats", _ptr(dy), _ptr(w), _ptr(dx), B, H, W, Cin, Cout, kh, kw, sh, sw, *nstat.args(),
               _ptr(ws), ws.numel() * 4, pre, _ptr(nstat.part), nstat.part.numel() * 4, _stream())
    else:
        L.call("cgs_conv2d_nhwc_bwd_data", _ptr(dy), _ptr(w), _ptr(dx), B, H, W, Cin, Cout, kh, kw, sh, sw,
               epilogue, _ptr(ep_a), _ptr(ep_aux), _ptr(ws), ws.numel() * 4, pre, _stream())
    if pr is not None:
        pr.done()
    return dx


def conv_signs_ok(op, B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, epilogue):
    """Can this call leave (forward, relu / lrelu epilogue) or take (backward-data, relu' / lrelu' epilogue) a sign mask
    instead of the fp32 aux tensor?  (include/cgs_hip.h, "sign masks")"""
    _, nbytes = WS.plan(op, B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, epilogue)
    return bool(L.load().cgs_conv_signs_ok(op, B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, epilogue, nbytes))


def deconv2d_fwd(x, w, bias, out_hw, sh=2, sw=2, epilogue=L.EPI_NONE, ep_a=None, ep_b=None, out=None, signs=None, part=None):
    """tf.nn.conv2d_transpose (default 'SAME') + bias (+ fused epilogue).  nsgan/ops.py:55,61-62.
    ``signs``: int32 [B*Ho*Wo*Cout/32] that receives the sign mask of the output (only where ``conv_signs_ok``).
    ``part``: [rows, 2, Cout] that receives the statistics partials of the output (no epilogue; rows = conv_stat_layout(...)[0])."""
    _chk(x, "x"); _chk(w, "w")
    B, H, W, Cin = x.shape
    kh, kw, Cout, Cin2 = w.shape
    if Cin2 != Cin:
        raise L.CgsError(f"deconv2d: weight Cin {Cin2} != input channels {Cin}")
    Ho, Wo = out_hw
    y = out if out is not None else torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    ws, pre = WS.get(w, L.DECONV_FWD, kh, kw, sh, sw, Cin, Cout, (B, H, W), epilogue, (Ho, Wo),
                     ptrs=(_ptr(x), _ptr(bias), _ptr(y), _ptr(ep_a), _ptr(ep_b)))
    pr = _Prof(2.0 * B * H * W * Cin * kh * kw * Cout, f"deconv_fwd {H}x{W} {Cin}->{Cout}", _nb(x, w, y)) if PROFILE is not None else None
    if part is not None:
        if epilogue != L.EPI_NONE or signs is not None:
            raise L.CgsError("deconv2d_fwd: the statistics partials are those of the plain output (no epilogue, no sign mask)")
        L.call("cgs_deconv2d_nhwc_fwd_stats", _ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw,
               _ptr(ws), ws.numel() * 4, pre, _ptr(part), part.numel() * 4, _stream())
    elif signs is not None:
        L.call("cgs_deconv2d_nhwc_fwd_signs", _ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw,
               epilogue, _ptr(ep_a), _ptr(ep_b), signs.data_ptr(), _ptr(ws), ws.numel() * 4, pre, _stream())
    else:
        L.call("cgs_deconv2d_nhwc_fwd", _ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw,
               epilogue, _ptr(ep_a), _ptr(ep_b), _ptr(ws), ws.numel() * 4, pre, _stream())
    if pr is not None:
        pr.done()
    return y


def deconv2d_bwd_data(dy, w, in_hw, sh=2, sw=2, out=None, epilogue=L.EPI_NONE, ep_a=None, ep_aux=None, ep_signs=None, nstat=None):
    """Input gradient of deconv2d_fwd: a strided 'SAME' conv of dy with the deconv weights.
    ``ep_signs``: the sign mask of the saved activation instead of ``ep_aux`` (relu' / lrelu' epilogues, where ``conv_signs_ok``).
    ``nstat``: as in conv2d_bwd_data."""
    _chk(dy, "dy"); _chk(w, "w")
    kh, kw, Cout, Cin = w.shape
    B, Ho, Wo, _ = dy.shape
    H, W = in_hw
    dx = out if out is not None else torch.empty((B, H, W, Cin), dtype=torch.float32, device=dy.device)
    ws, pre = WS.get(w, L.DECONV_BWD_DATA, kh, kw, sh, sw, Cin, Cout, (B, H, W), epilogue, (Ho, Wo),
                     ptrs=(_ptr(dy), _ptr(dx), _ptr(ep_a), _ptr(ep_aux)))
    pr = _Prof(2.0 * B * H * W * Cin * kh * kw * Cout, f"deconv_bwd {H}x{W} {Cin}<-{Cout}",
               _nb(dy, w, dx, ep_aux if ep_signs is None else ep_signs, nstat.x if nstat is not None else None)) if PROFILE is not None else None
    if nstat is not None:
        if epilogue != L.EPI_NONE or ep_signs is not None:
            raise L.CgsError("deconv2d_bwd_data: norm-backward statistics come with no epilogue")
        L.call("cgs_deconv2d_nhwc_bwd_data_nstats", _ptr(dy), _ptr(w), _ptr(dx), B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, *nstat.args(),
               _ptr(ws), ws.numel() * 4, pre, _ptr(nstat.part), nstat.part.numel() * 4, _stream())
    elif ep_signs is not None:
        L.call("cgs_deconv2d_nhwc_bwd_data_signs", _ptr(dy), _ptr(w), _ptr(dx), B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw,
               epilogue, _ptr(ep_a), ep_signs.data_ptr(), _ptr(ws), ws.numel() * 4, pre, _stream())
    else:
        L.call("cgs_deconv2d_nhwc_bwd_data", _ptr(dy), _ptr(w), _ptr(dx), B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw,
               epilogue, _ptr(ep_a), _ptr(ep_aux), _ptr(ws), ws.numel() * 4, pre, _stream())
    if pr is not None:
        pr.done()
    return dx


def linear_fwd(x, w, bias, epilogue=L.EPI_NONE, out=None):
    """tf.matmul(x, Matrix) + bias.  nsgan/ops.py:81-83."""
    _chk(x, "x"); _chk(w, "w")
    B, K = x.shape
    K2, N = w.shape
    if K2 != K:
        raise L.CgsError(f"linear: Matrix rows {K2} != input features {K}")
    y = out if out is not None else torch.empty((B, N), dtype=torch.float32, device=x.device)
    if N == 1:
        pr = _Prof(2.0 * B * K, "", _nb(x, w, y), op="linear_out1_fwd") if PROFILE is not None else None
        L.call("cgs_linear_fwd", _ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, K, N, epilogue, None, 0, 0, _stream())
        if pr is not None:
            pr.done()
        return y
    ws, pre = WS.get(w, L.CONV_FWD, 1, 1, 1, 1, K, N, (B, 1, 1), epilogue, ptrs=(_ptr(x), _ptr(bias), _ptr(y)))
    pr = _Prof(2.0 * B * K * N, f"linear_fwd {K}->{N}", _nb(x, w, y)) if PROFILE is not None else None
    L.call("cgs_linear_fwd", _ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, K, N, epilogue, _ptr(ws), ws.numel() * 4, pre, _stream())
    if pr is not None:
        pr.done()
    return y


def linear_bwd_data(dy, w, out=None):
    _chk(dy, "dy"); _chk(w, "w")
    B, N = dy.shape
    K = w.shape[0]
    dx = out if out is not None else torch.empty((B, K), dtype=torch.float32, device=dy.device)
    if N == 1:
        pr = _Prof(2.0 * B * K, "", _nb(dy, w, dx), op="linear_out1_bwd") if PROFILE is not None else None
        L.call("cgs_linear_bwd_data", _ptr(dy), _ptr(w), _ptr(dx), B, K, N, None, 0, 0, _stream())
        if pr is not None:
            pr.done()
        return dx
    ws, pre = WS.get(w, L.CONV_BWD_DATA, 1, 1, 1, 1, K, N, (B, 1, 1), ptrs=(_ptr(dy), _ptr(dx)))
    pr = _Prof(2.0 * B * K * N, f"linear_bwd {K}<-{N}", _nb(dy, w, dx)) if PROFILE is not None else None
    L.call("cgs_linear_bwd_data", _ptr(dy), _ptr(w), _ptr(dx), B, K, N, _ptr(ws), ws.numel() * 4, pre, _stream())
    if pr is not None:
        pr.done()
    return dx


# ----------------------------------------------------------------------------- batch norm / activations
def bn_train_lrelu_fwd(x, gamma, beta, leak=LEAK, eps=BN_EPS, out=None, stats=None):
    """Batch-statistics bn (+ lrelu; leak=1 -> plain bn).  Returns (y, mean, invstd).  nsgan/ops.py:19-26.
    ``stats`` = optional preallocated (mean, invstd)."""
    _chk(x, "x")
    C = x.shape[-1]
    M = x.numel() // C
    y = out if out is not None else torch.empty_like(x)
    if stats is not None:
        mean, invstd = stats
    else:
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = _bn_workspace(M, C, x.device)
    pr = _Prof(0.0, "", _nb(x, x, y), op="bn_train_lrelu_fwd") if PROFILE is not None else None     # statistics pass + apply pass
    L.call("cgs_bn_train_lrelu_fwd", _ptr(x), _ptr(gamma), _ptr(beta), eps, leak, _ptr(y), _ptr(mean), _ptr(invstd),
           M, C, _ptr(ws), ws.numel() * 4, _stream())
    if pr is not None:
        pr.done()
    return y, mean, invstd


def bn_train_lrelu_bwd_data(dy, x, gamma, beta, mean, invstd, leak=LEAK, out=None):
    _chk(dy, "dy"); _chk(x, "x")
    C = x.shape[-1]
    M = x.numel() // C
    dx = out if out is not None else torch.empty_like(x)
    ws = _bn_workspace(M, C, x.device)
    pr = _Prof(0.0, "", _nb(dy, x, dy, x, dx), op="bn_train_lrelu_bwd_data") if PROFILE is not None else None   # sums pass (dy, x) + apply pass (dy, x -> dx)
    L.call("cgs_bn_train_lrelu_bwd_data", _ptr(dy), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(invstd), leak,
           _ptr(dx), M, C, _ptr(ws), ws.numel() * 4, _stream())
    if pr is not None:
        pr.done()
    return dx


# synchronised batch statistics: the reductions stop at per-channel sums (double) that the caller all-reduces over the ranks
def bn_sync_fwd_sums(x, sums):
    """sums[2, C] (float64, device) <- (sum x, sum x^2) over the local rows."""
    _chk(x, "x")
    C = x.shape[-1]
    M = x.numel() // C
    ws = _bn_workspace(M, C, x.device)
    L.call("cgs_bn_sync_fwd_sums", _ptr(x), sums.data_ptr(), M, C, _ptr(ws), ws.numel() * 4, _stream())
    return sums


def bn_sync_fwd_apply(x, gamma, beta, sums, m_total, leak=LEAK, eps=BN_EPS, out=None, stats=None):
    _chk(x, "x")
    C = x.shape[-1]
    M = x.numel() // C
    y = out if out is not None else torch.empty_like(x)
    mean, invstd = stats if stats is not None else (torch.empty(C, dtype=torch.float32, device=x.device),
                                                    torch.empty(C, dtype=torch.float32, device=x.device))
    ws = _bn_workspace(M, C, x.device)
    L.call("cgs_bn_sync_fwd_apply", _ptr(x), _ptr(gamma), _ptr(beta), eps, leak, sums.data_ptr(), int(m_total), _ptr(y),
           _ptr(mean), _ptr(invstd), M, C, _ptr(ws), ws.numel() * 4, _stream())
    return y, mean, invstd


def bn_sync_bwd_sums(dy, x, gamma, beta, mean, invstd, sums, leak=LEAK):
    _chk(dy, "dy"); _chk(x, "x")
    C = x.shape[-1]
    M = x.numel() // C
    ws = _bn_workspace(M, C, x.device)
    L.call("cgs_bn_sync_bwd_sums", _ptr(dy), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(invstd), leak,
           sums.data_ptr(), M, C, _ptr(ws), ws.numel() * 4, _stream())
    return sums


def bn_sync_bwd_apply(dy, x, gamma, beta, mean, invstd, sums, m_total, leak=LEAK, out=None):
    _chk(dy, "dy"); _chk(x, "x")
    C = x.shape[-1]
    M = x.numel() // C
    dx = out if out is not None else torch.empty_like(x)
    ws = _bn_workspace(M, C, x.device)
    L.call("cgs_bn_sync_bwd_apply", _ptr(dy), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(invstd), leak,
           sums.data_ptr(), int(m_total), _ptr(dx), M, C, _ptr(ws), ws.numel() * 4, _stream())
    return dx


_in_ws = {}


def _in_workspace(B, HW, C, device):
    key = (B, HW, C, device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _in_ws.get(key)
    if ws is None:
        ws = torch.empty(int(L.load().cgs_instnorm_ws_bytes(B, HW, C)) // 4 + 4, dtype=torch.float32, device=device)
        _in_ws[key] = ws
    return ws


def instnorm_lrelu_fwd(x, scale, offset, leak=1.0, eps=BN_EPS, out=None, stats=None):
    """Instance norm over the H*W pixels of every (sample, channel) (+ lrelu; leak 0 = relu, 1 = none).
    x: [B,H,W,C].  Returns (y, mean[B,C], invstd[B,C])."""
    _chk(x, "x")
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    y = out if out is not None else torch.empty_like(x)
    mean, invstd = stats if stats is not None else (torch.empty((B, C), dtype=torch.float32, device=x.device),
                                                     torch.empty((B, C), dtype=torch.float32, device=x.device))
    ws = _in_workspace(B, HW, C, x.device)
    pr = _Prof(0.0, "", _nb(x, x, y), op="instnorm_lrelu_fwd") if PROFILE is not None else None
    L.call("cgs_instnorm_lrelu_fwd", _ptr(x), _ptr(scale), _ptr(offset), eps, leak, _ptr(y), _ptr(mean), _ptr(invstd), B, HW, C,
           _ptr(ws), ws.numel() * 4, _stream())
    if pr is not None:
        pr.done()
    return y, mean, invstd


def instnorm_lrelu_bwd_data(dy, x, scale, offset, mean, invstd, leak=1.0, out=None):
    _chk(dy, "dy"); _chk(x, "x")
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    dx = out if out is not None else torch.empty_like(x)
    ws = _in_workspace(B, HW, C, x.device)
    pr = _Prof(0.0, "", _nb(dy, x, dy, x, dx), op="instnorm_lrelu_bwd_data") if PROFILE is not None else None
    L.call("cgs_instnorm_lrelu_bwd_data", _ptr(dy), _ptr(x), _ptr(scale), _ptr(offset), _ptr(mean), _ptr(invstd), leak, _ptr(dx),
           B, HW, C, _ptr(ws), ws.numel() * 4, _stream())
    if pr is not None:
        pr.done()
    return dx


def norm_lrelu_bwd_from_partials(dy, x, nstat, groups, out=None):
    """Backward-data of batch / instance norm (+ lrelu) from the column sums the producing backward-data call left in ``nstat.part``
    (cgs_norm_lrelu_bwd_from_partials): finalize + one pass over dy and x.  dy / x: [groups * M_group, ..., C] (groups = 1: batch norm)."""
    _chk(dy, "dy"); _chk(x, "x")
    C_ = x.shape[-1]
    Mg = x.numel() // (groups * C_)
    dx = out if out is not None else torch.empty_like(x)
    ws = _in_workspace(groups, Mg, C_, x.device) if groups > 1 else _bn_workspace(Mg, C_, x.device)
    lay = nstat.layout
    pr = _Prof(0.0, "", _nb(dy, x, dx), op="norm_lrelu_bwd_from_partials") if PROFILE is not None else None
    L.call("cgs_norm_lrelu_bwd_from_partials", _ptr(dy), _ptr(x), _ptr(nstat.part), groups, lay[1], lay[2], lay[3], _ptr(nstat.gamma), _ptr(nstat.beta),
           _ptr(nstat.mean), _ptr(nstat.invstd), nstat.leak, _ptr(dx), Mg, C_, _ptr(ws), ws.numel() * 4, _stream())
    if pr is not None:
        pr.done()
    return dx


def add(a, b, out=None):
    _chk(a, "a"); _chk(b, "b")
    o = out if out is not None else torch.empty_like(a)
    L.call("cgs_add", _ptr(a), _ptr(b), _ptr(o), a.numel(), _stream())
    return o


def bn_fold(gamma, beta, moving_mean, moving_var, eps=BN_EPS, out=None):
    """Inference-mode bn as a per-channel affine (a, b).  nsgan/GAN.py:87,94.  ``out`` = (a, b) to refresh in place."""
    C = gamma.numel()
    a, b = out if out is not None else (torch.empty(C, dtype=torch.float32, device=gamma.device),
                                        torch.empty(C, dtype=torch.float32, device=gamma.device))
    L.call("cgs_bn_fold", _ptr(gamma), _ptr(beta), _ptr(moving_mean), _ptr(moving_var), eps, _ptr(a), _ptr(b), C, _stream())
    return a, b


def affine_relu_fwd(x, a, b, out=None):
    _chk(x, "x")
    C = x.shape[-1]
    y = out if out is not None else torch.empty_like(x)
    L.call("cgs_affine_relu_fwd", _ptr(x), _ptr(a), _ptr(b), _ptr(y), x.numel() // C, C, _stream())
    return y


def affine_relu_bwd(dy, y, a, out=None):
    _chk(dy, "dy"); _chk(y, "y")
    C = y.shape[-1]
    dx = out if out is not None else torch.empty_like(y)
    L.call("cgs_affine_relu_bwd", _ptr(dy), _ptr(y), _ptr(a), _ptr(dx), y.numel() // C, C, _stream())
    return dx


def affine_fwd(x, a, b, out=None):
    _chk(x, "x")
    C = x.shape[-1]
    y = out if out is not None else torch.empty_like(x)
    L.call("cgs_affine_fwd", _ptr(x), _ptr(a), _ptr(b), _ptr(y), x.numel() // C, C, _stream())
    return y


def affine_bwd(dy, a, out=None):
    _chk(dy, "dy")
    C = dy.shape[-1]
    dx = out if out is not None else torch.empty_like(dy)
    L.call("cgs_affine_bwd", _ptr(dy), _ptr(a), _ptr(dx), dy.numel() // C, C, _stream())
    return dx


def lrelu_fwd(x, leak=LEAK, out=None):
    _chk(x, "x")
    y = out if out is not None else torch.empty_like(x)
    L.call("cgs_lrelu_fwd", _ptr(x), leak, _ptr(y), x.numel(), _stream())
    return y


def lrelu_bwd(dy, y, leak=LEAK, out=None):
    _chk(dy, "dy"); _chk(y, "y")
    dx = out if out is not None else torch.empty_like(y)
    L.call("cgs_lrelu_bwd", _ptr(dy), _ptr(y), leak, _ptr(dx), y.numel(), _stream())
    return dx


def tanh_fwd(x, out=None):
    _chk(x, "x")
    y = out if out is not None else torch.empty_like(x)
    L.call("cgs_tanh_fwd", _ptr(x), _ptr(y), x.numel(), _stream())
    return y


def tanh_bwd(dy, y, out=None):
    _chk(dy, "dy"); _chk(y, "y")
    dx = out if out is not None else torch.empty_like(y)
    L.call("cgs_tanh_bwd", _ptr(dy), _ptr(y), _ptr(dx), y.numel(), _stream())
    return dx


# ----------------------------------------------------------------------------- refinement-loop scalars
def bce_ones_grad_rowmean(logits, dlogits=None, logit_mean=None):
    """d softplus(-l)/dl = sigmoid(l) - 1 and the per-sample mean logit (collaborator.py:31-37)."""
    _chk(logits, "logits")
    B = logits.shape[0]
    P = logits.numel() // B
    dl = dlogits if dlogits is not None else torch.empty_like(logits)
    lm = logit_mean if logit_mean is not None else torch.empty(B, dtype=torch.float32, device=logits.device)
    L.call("cgs_bce_ones_grad_rowmean", _ptr(logits), _ptr(dl), _ptr(lm), B, P, _stream())
    return dl, lm


def bce_ones_fwd(logits, out=None):
    """softplus(-logits): sigmoid_cross_entropy_with_logits(labels=1), unreduced (nsgan/GAN.py:176-177)."""
    _chk(logits, "logits")
    o = out if out is not None else torch.empty_like(logits)
    L.call("cgs_bce_ones_fwd", _ptr(logits), _ptr(o), logits.numel(), _stream())
    return o


def bce_ones_bwd(dloss, logits, out=None):
    """dloss * (sigmoid(logits) - 1)."""
    _chk(dloss, "dloss"); _chk(logits, "logits")
    o = out if out is not None else torch.empty_like(logits)
    L.call("cgs_bce_ones_bwd", _ptr(dloss), _ptr(logits), _ptr(o), logits.numel(), _stream())
    return o


def sigmoid_rowmean(logits, out=None):
    """Per-sample discriminator score: the mean over a sample's logits of sigmoid(logit), [B, 1]   (fake_sigmoids, nsgan/GAN.py:154-155)."""
    _chk(logits, "logits")
    B = logits.shape[0]
    o = out if out is not None else torch.empty((B, 1), dtype=torch.float32, device=logits.device)
    L.call("cgs_sigmoid_rowmean", _ptr(logits), _ptr(o), B, logits.numel() // B, _stream())
    return o


def clip(x, vmin, vmax, out=None):
    """tf.clip_by_value (sampling/collaborator.py:69-70)."""
    _chk(x, "x")
    o = out if out is not None else torch.empty_like(x)
    L.call("cgs_clip", _ptr(x), float(vmin), float(vmax), _ptr(o), x.numel(), _stream())
    return o


def refine_update(theta, m, g, rate, alpha, first, vmin=None, vmax=None):
    """In-place momentum / sgd step on theta (+ optional clip).  policy.py:27-37, collaborator.py:66-70."""
    use_clip = 1 if (vmin and vmax) else 0          # the reference's truthiness test (quirk Q5)
    pr = _Prof(0.0, "", _nb(theta, theta, m, g) + (0.0 if first else _nb(m)), op="refine_update") if PROFILE is not None else None
    L.call("cgs_refine_update", _ptr(theta), _ptr(m), _ptr(g), rate, alpha, 1 if first else 0, use_clip,
           float(vmin or 0.0), float(vmax or 0.0), theta.numel(), _stream())
    if pr is not None:
        pr.done()


def refine_select(theta, logit, forced, step_index, best_theta, best_logit, best_step):
    """Row-wise best-sample select.  collaborator.py:76-83."""
    B = theta.shape[0]
    L.call("cgs_refine_select", _ptr(theta), _ptr(logit), _ptr(forced), step_index, _ptr(best_theta), _ptr(best_logit),
           _ptr(best_step), B, theta.numel() // B, _stream())


def refine_select2(rows, best_rows, theta, best_theta, logit, forced, step_index, best_logit, best_step, tickets=None):
    """refine_select_rows(rows) + refine_select(theta) with the two row copies in one launch; with ``tickets`` (int32 [B], zero: the call leaves
    it zero) the scalars too -- otherwise they follow in a second launch."""
    B = theta.shape[0]
    if tickets is not None and (tickets.dtype != torch.int32 or tickets.numel() < B or not tickets.is_contiguous() or tickets.device != theta.device):
        raise L.CgsError("refine_select2: tickets must be a contiguous int32 device tensor of at least B elements")
    L.call("cgs_refine_select2", _ptr(rows), _ptr(best_rows), rows.numel() // B, _ptr(theta), _ptr(best_theta), theta.numel() // B, _ptr(logit),
           _ptr(forced), step_index, _ptr(best_logit), _ptr(best_step), _ptr(tickets), B, _stream())


def linear_out1_bce(x, w, bias, logits, dlogits, logit_mean):
    """The one-logit head of D + the loss seed (linear_fwd(N = 1) then bce_ones_grad_rowmean) in one launch."""
    _chk(x, "x"); _chk(w, "w")
    B, K = x.shape
    pr = _Prof(2.0 * B * K, "", _nb(x, w, logits), op="linear_out1_fwd") if PROFILE is not None else None
    L.call("cgs_linear_out1_bce", _ptr(x), _ptr(w), _ptr(bias), _ptr(logits), _ptr(dlogits), _ptr(logit_mean), B, K, _stream())
    if pr is not None:
        pr.done()
    return logits


def refine_select_rows(src, logit, forced, step_index, dst, best_logit):
    """Copy the rows of ``src`` whose sample is selected at this step into ``dst`` (predicate of refine_select;
    call before it, which updates best_logit)."""
    B = src.shape[0]
    L.call("cgs_refine_select_rows", _ptr(src), _ptr(logit), _ptr(forced), step_index, _ptr(dst), _ptr(best_logit),
           B, src.numel() // B, _stream())


# ----------------------------------------------------------------------------- D shaping step (weight gradients, Adam)
_wgrad_ws = {}


def _wgrad_workspace(nbytes, device):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _wgrad_ws.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(max(nbytes, 16) // 4 + 4, dtype=torch.float32, device=device)
        _wgrad_ws[key] = ws
    return ws


def conv2d_bwd_weight(x, dy, kh, kw, sh=2, sw=2, out=None, accumulate=False):
    """dw[kh,kw,Cin,Cout] (+)= weight gradient of conv2d_fwd (Conv2DBackpropFilter)."""
    _chk(x, "x"); _chk(dy, "dy")
    B, H, W, Cin = x.shape
    Cout = dy.shape[3]
    dw = out if out is not None else torch.empty((kh, kw, Cin, Cout), dtype=torch.float32, device=x.device)
    ws = _wgrad_workspace(int(L.load().cgs_conv_wgrad_ws_bytes(B, H, W, Cin, Cout, kh, kw, sh, sw)), x.device)
    # (the weight-gradient GEMM multiplies the zero-padding taps too: issued = nominal flops; the slab reduce rides in the same record)
    pr = _Prof(2.0 * B * dy.shape[1] * dy.shape[2] * Cout * kh * kw * Cin, "", _nb(x, dy, dw), op="wgrad_kernel") if PROFILE is not None else None
    L.call("cgs_conv2d_nhwc_bwd_weight", _ptr(x), _ptr(dy), _ptr(dw), B, H, W, Cin, Cout, kh, kw, sh, sw,
           1 if accumulate else 0, _ptr(ws), ws.numel() * 4, _stream())
    if pr is not None:
        pr.done()
    return dw


def linear_bwd_weight(x, dy, out=None, accumulate=False):
    """dw[in,out] (+)= x^T dy."""
    _chk(x, "x"); _chk(dy, "dy")
    B, K = x.shape
    N = dy.shape[1]
    dw = out if out is not None else torch.empty((K, N), dtype=torch.float32, device=x.device)
    ws = _wgrad_workspace(int(L.load().cgs_conv_wgrad_ws_bytes(B, 1, 1, K, N, 1, 1, 1, 1)), x.device)
    pr = _Prof(2.0 * B * K * N, "", _nb(x, dy, dw), op="wgrad_kernel") if PROFILE is not None else None
    L.call("cgs_linear_bwd_weight", _ptr(x), _ptr(dy), _ptr(dw), B, K, N, 1 if accumulate else 0, _ptr(ws), ws.numel() * 4, _stream())
    if pr is not None:
        pr.done()
    return dw


def bias_grad(dy, out=None, accumulate=False):
    """db[C] (+)= sum of dy over every axis but the last."""
    _chk(dy, "dy")
    C = dy.shape[-1]
    M = dy.numel() // C
    db = out if out is not None else torch.empty(C, dtype=torch.float32, device=dy.device)
    if C % 4 != 0:                                   # the single-logit head: a B-element sum
        s = dy.reshape(M, C).sum(0)
        db.copy_(db + s if accumulate else s)
        return db
    ws = _bn_workspace(M, C, dy.device)
    L.call("cgs_bias_grad", _ptr(dy), _ptr(db), M, C, 1 if accumulate else 0, _ptr(ws), ws.numel() * 4, _stream())
    return db


def bn_train_param_grads(x, dgamma, dbeta, accumulate=False):
    """(dgamma, dbeta) (+)= from the statistics the bn_train_lrelu_bwd_data call JUST made on ``x`` left in its workspace."""
    C = x.shape[-1]
    M = x.numel() // C
    ws = _bn_workspace(M, C, x.device)
    L.call("cgs_bn_train_param_grads", _ptr(ws), M, C, _ptr(dgamma), _ptr(dbeta), 1 if accumulate else 0, _stream())


def bce_logits_grad(logits, target, scale, dlogits=None, loss=None):
    """dlogits = scale*(sigmoid(l) - target); loss[0] = scale * sum BCE(l, target)."""
    _chk(logits, "logits")
    dl = dlogits if dlogits is not None else torch.empty_like(logits)
    L.call("cgs_bce_logits_grad", _ptr(logits), float(target), float(scale), _ptr(dl), _ptr(loss), logits.numel(), _stream())
    return dl


def adam_step(w, g, m, v, lr_t, beta1, beta2, eps):
    L.call("cgs_adam_step", _ptr(w), _ptr(g), _ptr(m), _ptr(v), float(lr_t), float(beta1), float(beta2), float(eps), w.numel(), _stream())
