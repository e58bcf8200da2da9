"""PolicyAdaptive: the refinement update rule (reference sampling/policy.py:5-64).

``sgd``      theta -= rate*g                                             (:27-29)
``momentum`` m = rate*g (first call) | 0.9*m + rate*g ; theta -= m       (:31-37)
``ladam``    Adam-like step rescaled by a running loss                   (:39-61)

Host arrays (numpy, the 2-D path of refiner_cpu) are updated IN PLACE like the reference's ``-=``
on an ndarray; torch tensors get a new tensor back (TF value semantics).  On GPU tensors the
momentum / sgd update is one fused kernel (cgs_refine_update).
"""
import numpy as np

try:
    import torch
except ImportError:  # pragma: no cover
    torch = None


class PolicyAdaptive(object):
    def __init__(self, step_size, method):
        self.method = method
        self.lambda_ = step_size
        self.alpha_ = 0.9                     # momentum decay
        self.beta1_, self.beta2_, self.beta3_ = 0.9, 0.5, 0.5
        self.degree_ = 2
        self.eps_ = 1e-8
        self.reset_moving_average()

    def reset_moving_average(self):
        self.momentum = None
        self.mean_square = None
        self.loss = None

    # -- helpers -------------------------------------------------------------------------------
    @staticmethod
    def _is_torch(x):
        return torch is not None and isinstance(x, torch.Tensor)

    def _commit(self, theta, new_value):
        """ndarray: write back in place (and return it); tensor: return the new value."""
        if isinstance(theta, np.ndarray):
            theta[...] = new_value
            return theta
        return new_value

    def apply_gradient(self, theta, grad, loss=None):
        if self.method == "sgd":
            if self._is_torch(theta) and theta.is_cuda:
                return self._device_step(theta, grad, alpha=0.0, first=True)
            return self._commit(theta, theta - self.lambda_ * grad)

        if self.method == "momentum":
            if self._is_torch(theta) and theta.is_cuda:
                first = self.momentum is None
                return self._device_step(theta, grad, alpha=self.alpha_, first=first)
            step = self.lambda_ * grad
            self.momentum = step if self.momentum is None else self.alpha_ * self.momentum + step
            return self._commit(theta, theta - self.momentum)

        if self.method == "ladam":
            if loss is None:
                raise ValueError("ladam needs the per-sample loss (the reference's map-space refiner never passes "
                                 "one: sampling/collaborator.py:66)")
            one = 1.0
            self.momentum = grad if self.momentum is None else self.beta1_ * self.momentum + (one - self.beta1_) * grad
            sq = grad ** 2
            self.mean_square = sq if self.mean_square is None else self.beta2_ * self.mean_square + (one - self.beta2_) * sq
            self.loss = loss if self.loss is None else self.beta3_ * self.loss + (one - self.beta3_) * loss
            if theta.shape[1] > 2:            # activation maps (tensor branch, upper clip 1e4)
                n = theta.shape[0]
                direction = (self.lambda_ * self.momentum / (torch.sqrt(self.mean_square) + self.eps_)).reshape(n, -1)
                gain = torch.clamp(self.loss + 0.5, 0.0, 10000.0).reshape(n, 1) ** self.degree_
                return (theta.reshape(n, -1) - direction * gain).reshape(theta.shape)
            gain = np.expand_dims((self.loss + 0.5).clip(min=0.0), axis=1) ** self.degree_
            return self._commit(theta, theta - self.lambda_ * self.momentum / (np.sqrt(self.mean_square) + self.eps_) * gain)

        raise NotImplementedError

    def _device_step(self, theta, grad, alpha, first):
        from .. import kernels as K
        if self.momentum is None:
            self.momentum = torch.empty_like(theta)
        new_theta = theta.clone()
        K.refine_update(new_theta, self.momentum, grad.contiguous(), self.lambda_, alpha, first)
        return new_theta
