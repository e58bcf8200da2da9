"""The reference's operator API (nsgan/ops.py:12-83) on torch GPU tensors over libcgs_hip.so.

Same function names, argument names, defaults and layouts -- ``bn``, ``conv2d``, ``deconv2d``,
``lrelu``, ``linear``, ``conv_out_size_same``, ``conv_cond_concat``, ``concat`` -- so that network
code written against nsgan/ops.py (nsgan/GAN.py:59-101) reads the same here.  TF's
``variable_scope`` / ``get_variable`` are restated as a small name -> tensor store with the same
variable names (``discriminator/d_conv1/w`` ...), which is also the checkpoint key space.

Each op is a ``torch.autograd.Function`` whose forward and backward-DATA run hand-written HIP
kernels (no weight gradients: the sampler only differentiates w.r.t. activations with frozen
weights, sampling/collaborator.py:31).  ``tf.gradients(loss, feature)`` becomes
``torch.autograd.grad(loss.sum(), feature)``.
"""
import contextlib
import math

import torch

from . import kernels as K
from . import lib as L

# ----------------------------------------------------------------------------- variable store
_VARS = {}
_SCOPE = []
_REUSE = [False]
_DEVICE = [None]
_GENERATION = 0


def set_device(device):
    """Device new variables are created on (default: cuda:0)."""
    _DEVICE[0] = torch.device(device)


def _device():
    return _DEVICE[0] if _DEVICE[0] is not None else torch.device("cuda:0")


@contextlib.contextmanager
def variable_scope(name, reuse=None):
    """tf.variable_scope(name, reuse=...)."""
    _SCOPE.append(name)
    _REUSE.append(_REUSE[-1] if reuse is None else bool(reuse))
    try:
        yield "/".join(_SCOPE)
    finally:
        _SCOPE.pop()
        _REUSE.pop()


def get_variable(name, shape, initializer):
    """tf.get_variable: look up ``scope/name``; create it with ``initializer(shape)`` unless reusing."""
    full = "/".join(_SCOPE + [name])
    v = _VARS.get(full)
    if v is None:
        if _REUSE[-1]:
            raise ValueError(f"Variable {full} does not exist, or was not created with get_variable() (reuse=True)")
        v = initializer(tuple(int(s) for s in shape)).float().contiguous().to(_device())
        _VARS[full] = v
    elif tuple(v.shape) != tuple(int(s) for s in shape):
        raise ValueError(f"Variable {full} has shape {tuple(v.shape)}, requested {tuple(shape)}")
    return v


def variables():
    return _VARS


def set_variables(params, device=None):
    """Load a name -> array checkpoint (TF variable names) into the store."""
    dev = torch.device(device) if device is not None else _device()
    for k, v in params.items():
        new = torch.as_tensor(v).float().contiguous()
        old = _VARS.get(k)
        if old is not None and old.device == dev and tuple(old.shape) == tuple(new.shape):
            # restore INTO the existing tensor (tf.train.Saver.restore assigns to the variables, nsgan/GAN.py:473-491):
            # engines and refiners built before the load keep pointing at live storage
            with torch.no_grad():
                old.copy_(new)
        else:
            _VARS[k] = new.to(dev)
    K.WS.invalidate()                     # packed copies of the old values are stale
    global _GENERATION
    _GENERATION += 1


def generation():
    """Bumped by every set_variables(): consumers that derived state from the values (folded inference-bn affines in a
    compiled engine) rebuild when it moves."""
    return _GENERATION


def reset_variables():
    _VARS.clear()
    K.WS.clear()


def truncated_normal_initializer(stddev):
    def init(shape):
        t = torch.empty(shape)
        torch.nn.init.trunc_normal_(t, 0.0, stddev, -2 * stddev, 2 * stddev)
        return t
    return init


def random_normal_initializer(stddev):
    return lambda shape: torch.randn(shape) * stddev


def constant_initializer(value):
    return lambda shape: torch.full(shape, float(value))


# ----------------------------------------------------------------------------- autograd glue
# Input gradients are the refinement path (frozen weights, sampling/collaborator.py:31).  Parameter gradients are
# produced only for parameters that require grad (the D shaping step / training callers, nsgan/GAN.py:141-146): the layer
# input is kept for them only then, so the refinement path saves nothing extra.
def _keep(x, *params):
    return x if any(p.requires_grad for p in params) else None


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, sh, sw):
        x = x.contiguous()
        ctx.save_for_backward(w, _keep(x, w))
        ctx.hw, ctx.s = (x.shape[1], x.shape[2]), (sh, sw)
        return K.conv2d_fwd(x, w, b, sh, sw)

    @staticmethod
    def backward(ctx, dy):
        w, x = ctx.saved_tensors
        dy = dy.contiguous()
        nx, nw, nb = ctx.needs_input_grad[:3]
        dw = K.conv2d_bwd_weight(x, dy, w.shape[0], w.shape[1], *ctx.s) if nw else None
        db = K.bias_grad(dy) if nb else None
        dx = K.conv2d_bwd_data(dy, w, ctx.hw, *ctx.s) if nx else None
        return dx, dw, db, None, None


class _Deconv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, ho, wo, sh, sw):
        x = x.contiguous()
        ctx.save_for_backward(w, _keep(x, w))
        ctx.hw, ctx.s = (x.shape[1], x.shape[2]), (sh, sw)
        return K.deconv2d_fwd(x, w, b, (ho, wo), sh, sw)

    @staticmethod
    def backward(ctx, dy):
        w, x = ctx.saved_tensors
        dy = dy.contiguous()
        nx, nw, nb = ctx.needs_input_grad[:3]
        # w is the HWIO filter of the conv this op is the adjoint of (big = output side): its gradient is that conv's
        # filter gradient with the roles swapped -- the conv's input is dy, its output gradient is x
        dw = K.conv2d_bwd_weight(dy, x, w.shape[0], w.shape[1], *ctx.s) if nw else None
        db = K.bias_grad(dy) if nb else None
        dx = K.deconv2d_bwd_data(dy, w, ctx.hw, *ctx.s) if nx else None
        return dx, dw, db, None, None, None, None


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        ctx.save_for_backward(w, _keep(x, w))
        return K.linear_fwd(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        w, x = ctx.saved_tensors
        dy = dy.contiguous()
        nx, nw, nb = ctx.needs_input_grad[:3]
        dw = K.linear_bwd_weight(x, dy) if nw else None
        db = K.bias_grad(dy) if nb else None
        dx = K.linear_bwd_data(dy, w) if nx else None
        return dx, dw, db


class _BnTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, leak):
        x = x.contiguous()
        y, mean, invstd = K.bn_train_lrelu_fwd(x, gamma, beta, leak)
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.leak = leak
        ctx.mark_non_differentiable(mean, invstd)
        return y, mean, invstd

    @staticmethod
    def backward(ctx, dy, _dm, _di):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        dx = K.bn_train_lrelu_bwd_data(dy.contiguous(), x, gamma, beta, mean, invstd, ctx.leak)
        dgamma = dbeta = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
            K.bn_train_param_grads(x, dgamma, dbeta)          # from the sums the backward-data call just left in its workspace
        return dx, dgamma, dbeta, None


class _Affine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, a, b):
        ctx.save_for_backward(a)
        return K.affine_fwd(x.contiguous(), a, b)

    @staticmethod
    def backward(ctx, dy):
        (a,) = ctx.saved_tensors
        return K.affine_bwd(dy.contiguous(), a), None, None


class _Lrelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, leak):
        y = K.lrelu_fwd(x.contiguous(), leak)
        ctx.save_for_backward(y)
        ctx.leak = leak
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return K.lrelu_bwd(dy.contiguous(), y, ctx.leak), None


class _Tanh(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = K.tanh_fwd(x.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return K.tanh_bwd(dy.contiguous(), y)


class _InstNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, offset, leak):
        x = x.contiguous()
        y, mean, invstd = K.instnorm_lrelu_fwd(x, scale, offset, leak)
        ctx.save_for_backward(x, scale, offset, mean, invstd)
        ctx.leak = leak
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, offset, mean, invstd = ctx.saved_tensors
        return K.instnorm_lrelu_bwd_data(dy.contiguous(), x, scale, offset, mean, invstd, ctx.leak), None, None, None


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return K.add(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class _BceOnes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits):
        logits = logits.contiguous()
        ctx.save_for_backward(logits)
        return K.bce_ones_fwd(logits)

    @staticmethod
    def backward(ctx, dloss):
        (logits,) = ctx.saved_tensors
        return K.bce_ones_bwd(dloss.contiguous(), logits)


# ----------------------------------------------------------------------------- the operator API
def concat(tensors, axis, *args, **kwargs):
    """nsgan/ops.py:12-17."""
    return torch.cat(tensors, dim=axis)


def bn(x, is_training, scope, leak=1.0):
    """nsgan/ops.py:19-26: contrib batch_norm(decay=0.9, epsilon=1e-5, scale=True, updates_collections=None).
    is_training=True: batch statistics (and the moving averages are updated in place);
    is_training=False: moving averages.  ``leak`` (extension) fuses a following lrelu into the kernel."""
    C = x.shape[-1]
    with variable_scope(scope):
        beta = get_variable("beta", [C], constant_initializer(0.0))
        gamma = get_variable("gamma", [C], constant_initializer(1.0))
        mm = get_variable("moving_mean", [C], constant_initializer(0.0))
        mv = get_variable("moving_variance", [C], constant_initializer(1.0))
    if is_training:
        y, mean, invstd = _BnTrain.apply(x, gamma, beta, float(leak))
        with torch.no_grad():                      # decay 0.9 moving averages (nothing on the sampling path reads D's)
            # four launches for both statistics (nine as separate tensor expressions: on the generic path, where every launch is host time, D's
            # two norms made a third of all launches of a batch-64 MNIST call)
            var = invstd.pow(-2).sub_(K.BN_EPS)     # the (biased) batch variance behind invstd = 1 / sqrt(var + eps)
            torch._foreach_mul_([mm, mv], 0.9)
            torch._foreach_add_([mm, mv], [mean, var], alpha=0.1)
        return y
    a, b = K.bn_fold(gamma, beta, mm, mv)
    y = _Affine.apply(x, a, b)
    return y if leak == 1.0 else _Lrelu.apply(y, float(leak))


def instance_norm(x, scope, leak=1.0):
    """Instance norm over the pixels of every (sample, channel), variables ``scale`` / ``offset`` (extension: the CycleGAN /
    PatchGAN nets of BASELINE config 5; the reference has no such op).  Frozen parameters: input gradient only."""
    C = x.shape[-1]
    with variable_scope(scope):
        scale = get_variable("scale", [C], constant_initializer(1.0))
        offset = get_variable("offset", [C], constant_initializer(0.0))
    return _InstNorm.apply(x, scale, offset, float(leak))


def add(a, b):
    """a + b (residual connections) on the HIP path."""
    return _Add.apply(a, b)


def conv_out_size_same(size, stride):
    """nsgan/ops.py:28-29."""
    return int(math.ceil(float(size) / float(stride)))


def conv_cond_concat(x, y):
    """nsgan/ops.py:31-35: concatenate a conditioning vector on the feature-map axis."""
    return concat([x, y * torch.ones([x.shape[0], x.shape[1], x.shape[2], y.shape[3]], device=x.device)], 3)


def conv2d(input_, output_dim, k_h=5, k_w=5, d_h=2, d_w=2, stddev=0.02, name="conv2d"):
    """nsgan/ops.py:37-46."""
    with variable_scope(name):
        w = get_variable("w", [k_h, k_w, input_.shape[-1], output_dim], truncated_normal_initializer(stddev))
        biases = get_variable("biases", [output_dim], constant_initializer(0.0))
    return _Conv2d.apply(input_, w, biases, d_h, d_w)


def deconv2d(input_, output_shape, k_h=5, k_w=5, d_h=2, d_w=2, name="deconv2d", stddev=0.02, with_w=False):
    """nsgan/ops.py:48-67.  ``output_shape`` = [batch, height, width, channels] as in the reference."""
    if int(output_shape[0]) != int(input_.shape[0]):
        raise ValueError(f"deconv2d: output_shape batch {output_shape[0]} != input batch {input_.shape[0]}")
    with variable_scope(name):
        w = get_variable("w", [k_h, k_w, output_shape[-1], input_.shape[-1]], random_normal_initializer(stddev))
        biases = get_variable("biases", [output_shape[-1]], constant_initializer(0.0))
    deconv = _Deconv2d.apply(input_, w, biases, int(output_shape[1]), int(output_shape[2]), d_h, d_w)
    if with_w:
        return deconv, w, biases
    return deconv


def lrelu(x, leak=0.2, name="lrelu"):
    """nsgan/ops.py:69-70."""
    return _Lrelu.apply(x, float(leak))


def relu(x):
    """tf.nn.relu (nsgan/GAN.py:89-96)."""
    return _Lrelu.apply(x, 0.0)


def tanh(x):
    """tf.nn.tanh (nsgan/GAN.py:100)."""
    return _Tanh.apply(x)


def linear(input_, output_size, scope=None, stddev=0.02, bias_start=0.0, with_w=False):
    """nsgan/ops.py:72-83."""
    with variable_scope(scope or "Linear"):
        matrix = get_variable("Matrix", [input_.shape[1], output_size], random_normal_initializer(stddev))
        bias = get_variable("bias", [output_size], constant_initializer(bias_start))
    y = _Linear.apply(input_, matrix, bias)
    if with_w:
        return y, matrix, bias
    return y


def sigmoid_cross_entropy_with_logits_ones(logits):
    """tf.nn.sigmoid_cross_entropy_with_logits(logits, labels=ones), unreduced (nsgan/GAN.py:176-177)."""
    return _BceOnes.apply(logits)


class _Constant(object):
    """``ones_like`` / ``zeros_like`` of the label argument below: a constant that is never materialised."""

    def __init__(self, value, like):
        self.value, self.shape = float(value), tuple(like.shape)


def ones_like(x):
    """tf.ones_like(x) as a label constant (nsgan/GAN.py:177)."""
    return _Constant(1.0, x)


def zeros_like(x):
    """tf.zeros_like(x) as a label constant (nsgan/GAN.py:129)."""
    return _Constant(0.0, x)


def sigmoid_cross_entropy_with_logits(_sentinel=None, labels=None, logits=None, name=None):
    """tf.nn.sigmoid_cross_entropy_with_logits(labels=, logits=), unreduced: max(x, 0) - x z + log(1 + exp(-|x|)).
    Keyword arguments only, as in TF.  ``labels`` = ``ones_like(logits)`` (the refiner's loss, nsgan/GAN.py:176-177) is one
    kernel forward and one backward (``cgs_bce_ones_fwd / _bwd``); zeros is the same kernel on -x; a label TENSOR z adds
    (1 - z) x to the all-ones form (elementwise torch arithmetic: not on the refinement path)."""
    if _sentinel is not None or labels is None or logits is None:
        raise ValueError("Only call `sigmoid_cross_entropy_with_logits` with named arguments (labels=..., logits=...)")
    if isinstance(labels, _Constant):
        if labels.shape != tuple(logits.shape):
            raise ValueError(f"logits and labels must have the same shape ({tuple(logits.shape)} vs {labels.shape})")
        if labels.value == 1.0:
            return _BceOnes.apply(logits)
        if labels.value == 0.0:
            return _BceOnes.apply(-logits)
        labels = torch.full_like(logits, labels.value)
    if tuple(labels.shape) != tuple(logits.shape):
        raise ValueError(f"logits and labels must have the same shape ({tuple(logits.shape)} vs {tuple(labels.shape)})")
    return _BceOnes.apply(logits) + (1.0 - labels) * logits
