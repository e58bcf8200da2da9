"""IndependenceSampler: the MH-GAN independence chain with thinning
(reference sampling/idpsampler.py:4-53).  Serial by nature; host code."""
import numpy as np


class IndependenceSampler():
    def __init__(self, T=5, B=0):
        self.d_curr = None        # D score of the chain's current state
        self.cnt_chain = 1        # thinning counter, persists across calls
        self.thin_period = T
        self.burn_in = B

    def set_score_curr(self, d_curr):
        """Seed the chain (burn-in score)."""
        self.d_curr = d_curr

    def next(self, d_next):
        """One MH proposal with score ``d_next``; True if the chain moves."""
        if self.d_curr is not None:
            ratio = d_next * (1.0 - self.d_curr) / (self.d_curr * (1.0 - d_next))
            if np.random.uniform(0, 1) > min(1.0, ratio):
                return False
        self.d_curr = d_next
        return True

    def sampling(self, samples, sigmoids):
        assert samples.shape[0] == sigmoids.shape[0]
        assert np.min(sigmoids) >= 0.0
        assert np.max(sigmoids) <= 1.0
        kept, state, moves = [], None, 0
        for sample, score in zip(samples, sigmoids):
            if self.next(score):
                moves += 1
                if moves > self.burn_in:
                    state = sample
            if state is None:
                continue
            if self.cnt_chain > self.thin_period:
                kept.append(state)
                self.cnt_chain = 1
            else:
                self.cnt_chain += 1
        return np.asarray(kept, dtype=np.float32)
