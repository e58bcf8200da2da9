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

    def _accept(self, d_next, u=None):
        d = self.d_curr
        if d is None:
            return True
        odds = d_next * (1.0 - d) / (d * (1.0 - d_next))
        if u is None:
            u = np.random.uniform(0, 1)
        return not (u > min(1.0, odds))

    def next(self, d_next, u=None):
        """Offer one proposal; returns True (and moves the chain) if it is accepted.  ``u`` (extension): the uniform to compare
        against instead of one drawn from the global stream now."""
        moved = self._accept(d_next, u)
        if moved:
            self.d_curr = d_next
        return moved

    def _emit_due(self):
        """Thinning: every (T+1)-th visit emits; the counter restarts at 1 after an emission."""
        due = self.cnt_chain > self.thin_period
        self.cnt_chain = 1 if due else self.cnt_chain + 1
        return due

    def walk(self, sigmoids, uniforms=None):
        """One batch of proposals through the chain -> the row index emitted at every thinning point (a row may repeat).
        ``uniforms`` (extension): one pre-drawn U(0,1) per proposal -- ``np.random.uniform(0, 1, size=len(sigmoids))`` drawn where
        the reference's loop would have reached this batch consumes the global stream exactly as the per-proposal draws do, so a
        caller that runs the chain LATER (evaluate.collaborate_fused: while the GPU refines the next round) reproduces the
        reference's results bit for bit.  Needs a started chain (``set_score_curr``): an unstarted one takes its first proposal
        without a draw."""
        n = len(sigmoids)
        if n and (np.min(sigmoids) < 0.0 or np.max(sigmoids) > 1.0):
            raise AssertionError("scores must be sigmoids in [0, 1]")
        if uniforms is not None:
            if self.d_curr is None:
                raise ValueError("pre-drawn uniforms need a started chain (set_score_curr): an unstarted chain consumes one draw less")
            if len(uniforms) != n:
                raise AssertionError("one uniform per proposal")
        out = []
        current = -1                 # row held by the chain within this call
        n_moves = 0
        flat = np.asarray(sigmoids).reshape(n, -1) if n else None
        if n and flat.shape[1] == 1:
            # one score per proposal (the reference's [B, 1] sigmoids): the same arithmetic as ``next`` on numpy SCALARS of the
            # scores' dtype instead of 1-element arrays (same IEEE operations and promotions, a tenth of the host time -- the chain
            # is the serial part of the fill loop), the chain state handed back in the caller's form at the end
            col, d, last = flat[:, 0], self.d_curr, -1
            if isinstance(d, np.ndarray) and d.size == 1:
                d = d.reshape(-1)[0]
            T, B0, cnt = self.thin_period, self.burn_in, self.cnt_chain
            for idx in range(n):
                dn = col[idx]
                if d is None:
                    moved = True
                else:
                    odds = dn * (1.0 - d) / (d * (1.0 - dn))
                    u = np.random.uniform(0, 1) if uniforms is None else uniforms[idx]
                    moved = not (u > min(1.0, odds))
                if moved:
                    d, last = dn, idx
                    n_moves += 1
                    if n_moves > B0:
                        current = idx
                if current >= 0:
                    if cnt > T:
                        out.append(current)
                        cnt = 1
                    else:
                        cnt += 1
            self.cnt_chain = cnt
            if last >= 0:
                self.d_curr = sigmoids[last]
            return out
        for idx in range(n):
            if self.next(sigmoids[idx], None if uniforms is None else uniforms[idx]):
                n_moves += 1
                if n_moves > self.burn_in:
                    current = idx
            if current >= 0 and self._emit_due():
                out.append(current)
        return out

    def sampling(self, samples, sigmoids, uniforms=None):
        if samples.shape[0] != sigmoids.shape[0]:
            raise AssertionError("one score per sample")
        return np.asarray([samples[i] for i in self.walk(sigmoids, uniforms)], dtype=np.float32)
