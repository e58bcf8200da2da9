"""Rejector: Discriminator Rejection Sampling accept/reject (reference sampling/rejector.py:7-38).

Host float64 math with the global numpy RNG, exactly the reference's operation order, so the
accept mask is bit-identical for identical sigmoids and RNG state."""
import numpy as np
from scipy.special import expit, logit

_LO, _HI = 1e-14, 1 - 1e-14


class Rejector(object):
    def __init__(self):
        self.D_tilde_M = 0.0                          # running max logit; 0.0 = logit(0.5)
        self.last_accept = None                       # mask of the latest sampling() call (extension)

    def set_score_max(self, score_max):
        self.D_tilde_M = logit(np.clip(np.asarray(score_max).astype(float), _LO, _HI))

    def acceptance_probability(self, sigmoids, epsilon=1e-8, shift_percent=60.0):
        """P(accept) per sample; updates the running bound D_tilde_M (rejector.py:18-30)."""
        d_tilde = logit(np.clip(np.asarray(sigmoids).astype(float), _LO, _HI))
        self.D_tilde_M = np.maximum(self.D_tilde_M, np.amax(d_tilde))
        delta = d_tilde - self.D_tilde_M
        F = delta - np.log(1 - np.exp(delta - epsilon))
        if shift_percent is not None:
            F = F - np.percentile(F, shift_percent)
        return np.squeeze(expit(F)), len(delta)

    def sampling(self, samples, sigmoids, epsilon=1e-8, shift_percent=60.0, ranking=None):
        if ranking is not None:
            raise NotImplementedError
        P, n = self.acceptance_probability(sigmoids, epsilon, shift_percent)
        self.last_accept = np.random.rand(n) < P      # one uniform per sample from the global RNG (:33)
        return samples[self.last_accept]
