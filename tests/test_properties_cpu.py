"""Property tests (hypothesis) for the host-side accept/reject samplers and the update rule: invariants that hold for
any input, complementing the golden-vector tests."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from cgs_amd.sampling import IndependenceSampler, PolicyAdaptive, Rejector
from oracle import sampling_ref as S

sigm = st.lists(st.floats(min_value=1e-6, max_value=1 - 1e-6), min_size=1, max_size=200)


@settings(max_examples=60, deadline=None)
@given(sig=sigm, seed=st.integers(0, 2 ** 31 - 1), pct=st.sampled_from([None, 0.0, 60.0, 100.0]))
def test_rejector_mask_equals_oracle_and_is_monotone_in_state(sig, seed, pct):
    s = np.asarray(sig, dtype=np.float32).reshape(-1, 1)
    samples = np.arange(len(sig), dtype=np.float32).reshape(-1, 1)
    r, o = Rejector(), S.RejectorRef()
    for _ in range(2):                                   # two consecutive calls: the running bound persists
        np.random.seed(seed)
        good = r.sampling(samples, s, shift_percent=pct)
        np.random.seed(seed)
        mask, P = o.accept_mask(s, shift_percent=pct)
        assert np.array_equal(r.last_accept, mask)       # bit-exact with the pinned oracle
        assert np.array_equal(good[:, 0], samples[mask, 0])
        assert r.D_tilde_M == o.D_tilde_M and r.D_tilde_M >= 0.0          # bound never decreases (starts at logit(.5) = 0)
        assert np.all((P >= 0) & (P <= 1))
    if pct == 100.0:                                     # shift by the max => max P is exactly 1/2
        assert abs(np.max(P) - 0.5) < 1e-12


@settings(max_examples=60, deadline=None)
@given(sig=sigm, seed=st.integers(0, 2 ** 31 - 1), T=st.integers(0, 25), d0=st.floats(min_value=0.01, max_value=0.99))
def test_independence_sampler_matches_oracle_and_thinning(sig, seed, T, d0):
    s = np.asarray(sig, dtype=np.float64).reshape(-1, 1)
    samples = np.arange(len(sig), dtype=np.float32).reshape(-1, 1)
    mh, o = IndependenceSampler(T=T), S.IndependenceSamplerRef(T=T)
    mh.set_score_curr(d0); o.d_curr = d0
    np.random.seed(seed)
    got = mh.sampling(samples, s)
    np.random.seed(seed)
    want = o.accepted_indices(s)
    assert list(got[:, 0].astype(int)) == want if len(want) else got.shape[0] == 0
    assert len(want) <= len(sig) // (T + 1) + 1          # at most one sample per thinning period
    assert all(a <= b for a, b in zip(want, want[1:]))   # the chain only moves forward


@settings(max_examples=40, deadline=None)
@given(n=st.integers(1, 40), steps=st.integers(1, 6), rate=st.floats(min_value=1e-3, max_value=1.0),
       seed=st.integers(0, 10 ** 6), method=st.sampled_from(["sgd", "momentum", "ladam"]))
def test_policy_matches_oracle_state_machine(n, steps, rate, seed, method):
    rs = np.random.RandomState(seed)
    th = rs.randn(n, 2).astype(np.float32)
    p, o = PolicyAdaptive(rate, method), S.Policy(rate, method)
    ref = th.copy()
    for _ in range(steps):
        g, l = rs.randn(n, 2).astype(np.float32), rs.randn(n).astype(np.float32)
        p.apply_gradient(th, g, l)
        ref = o.step(ref, g, l).astype(np.float32)
        np.testing.assert_array_equal(th, ref)
    p.reset_moving_average()
    assert p.momentum is None and p.mean_square is None and p.loss is None
