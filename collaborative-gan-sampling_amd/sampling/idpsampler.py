"""IndependenceSampler: Metropolis-Hastings independence chain over discriminator scores with thinning
(behaviour of the reference's sampling/idpsampler.py:4-53; the MH-GAN acceptance rule).

The chain state is the D score of the last accepted proposal.  A proposal with score d' replaces a state with score d with
probability min(1, d'(1-d) / (d(1-d'))); the first proposal of a fresh chain is always taken.  ``sampling`` walks one batch of
proposals through the chain and emits the current state every ``T``-th visit (thinning) once ``B`` moves have happened
(burn-in); both counters live on the instance, so consecutive batches continue one chain.  Inherently serial: host code.
Bit-compatibility with the reference (tests/golden g7) needs the global numpy RNG to be consumed exactly once per proposal
that has a predecessor, which is what ``next`` does."""
import numpy as np


class IndependenceSampler:
    def __init__(self, T=5, B=0):
        self.thin_period = T
        self.burn_in = B
        self.cnt_chain = 1           # visits since the last emitted sample (kept across calls)
        self.d_curr = None           # score of the chain's current state; None = chain not started

    def set_score_curr(self, d_curr):
        """Start (or restart) the chain from a state with score ``d_curr``."""
        self.d_curr = d_curr

    def _accept(self, d_next):
        d = self.d_curr
        if d is None:
            return True
        odds = d_next * (1.0 - d) / (d * (1.0 - d_next))
        return not (np.random.uniform(0, 1) > min(1.0, odds))

    def next(self, d_next):
        """Offer one proposal; returns True (and moves the chain) if it is accepted."""
        moved = self._accept(d_next)
        if moved:
            self.d_curr = d_next
        return moved

    def _emit_due(self):
        """Thinning: every (T+1)-th visit emits; the counter restarts at 1 after an emission."""
        due = self.cnt_chain > self.thin_period
        self.cnt_chain = 1 if due else self.cnt_chain + 1
        return due

    def sampling(self, samples, sigmoids):
        if samples.shape[0] != sigmoids.shape[0]:
            raise AssertionError("one score per sample")
        if np.min(sigmoids) < 0.0 or np.max(sigmoids) > 1.0:
            raise AssertionError("scores must be sigmoids in [0, 1]")
        out = []
        current = None               # sample held by the chain within this call
        n_moves = 0
        for idx in range(samples.shape[0]):
            if self.next(sigmoids[idx]):
                n_moves += 1
                if n_moves > self.burn_in:
                    current = samples[idx]
            if current is not None and self._emit_due():
                out.append(current)
        return np.asarray(out, dtype=np.float32)
