"""Mask-level parity of the accept / reject step on DEVICE-produced scores (`north_star`: "bit-exact for the
rejector's accept mask").

The sampler classes are bit-exact against the reference on GIVEN sigmoids (tests/golden g6 / g7, CPU suite).  What the reference
actually runs is `fake_sigmoids` from the device into `Rejector.sampling` / `IndependenceSampler.sampling`
(/root/reference/nsgan/GAN.py:314-324, :409-412; sampling/rejector.py:16-38; sampling/idpsampler.py:17-53).  Here that path, level by
level, on one refined pool per net:

(a) `engine.score` (sigmoid(D(images)), D on batch statistics per logical batch, on the GPU: csrc/elementwise.hip's expf) against
    the oracle D's torch-CPU sigmoids of the SAME images: max |delta| stated and bounded;
(b) `Rejector.sampling` walked batch by batch under one seed from both score sets: the accept masks are equal, EXCEPT at samples
    whose uniform draw lies between the two acceptance probabilities -- a tie that only a score difference can decide.  Every
    flip is held to exactly that criterion (u within [min(P_dev, P_orc), max(...)]), the probabilities themselves to a bound, and
    the count is printed (expected n x mean|dP|: none in a 1024-sample pool);
(c) `IndependenceSampler` under one seed from both score sets: the emitted indices are equal, except after a chain decision whose
    uniform lies between the two acceptance ratios while both chains hold the same row; every such primary flip is held to that
    criterion and counted; with none the index lists (and the classes' own `sampling` output under the global RNG) are identical.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nets_ref as N

RESULTS = {}


def _mh_lockstep(sig_a, sig_b, d0, uniforms, T, batch):
    """Two MH independence chains (idpsampler.py:27-53 arithmetic: odds = d'(1-d) / (d(1-d')), accept unless u > min(1, odds)) fed
    the SAME proposals in the same order with the SAME uniforms but scores from two sources -> (indices emitted by a, by b, primary
    flips [(row, u, alpha_a, alpha_b)]): a primary flip is a differing decision while both chains hold the same row."""
    out = {"a": [], "b": []}
    st = {"a": [type(sig_a[0, 0])(d0), -1, -1, 1], "b": [type(sig_b[0, 0])(d0), -1, -1, 1]}      # score, held row, current (emitted) row, thinning counter
    flips = []
    for i in range(len(sig_a)):
        if i % batch == 0:
            st["a"][2] = st["b"][2] = -1            # (``walk`` emits only rows of the batch it is called with: idpsampler.py:41-52 per call)
        u, dec, al = uniforms[i], {}, {}
        for k, sig in (("a", sig_a), ("b", sig_b)):
            d, dn = st[k][0], sig[i, 0]
            al[k] = min(1.0, dn * (1.0 - d) / (d * (1.0 - dn)))
            dec[k] = not (u > al[k])
        if dec["a"] != dec["b"] and st["a"][1] == st["b"][1]:
            flips.append((i, float(u), float(al["a"]), float(al["b"])))
        for k, sig in (("a", sig_a), ("b", sig_b)):
            if dec[k]:
                st[k][0], st[k][1], st[k][2] = sig[i, 0], i, i
            if st[k][2] >= 0:
                if st[k][3] > T:
                    out[k].append(st[k][2]); st[k][3] = 1
                else:
                    st[k][3] += 1
    return out["a"], out["b"], flips


@pytest.mark.parametrize("arch,b,G,K", [("mnist", 64, 16, 50), ("dcgan32", 256, 4, 20)])
def test_accept_masks_from_device_scores_equal_those_from_oracle_scores(arch, b, G, K):
    from cgs_amd.engine import RefineEngine
    from cgs_amd.nets import to_device
    from cgs_amd.sampling import IndependenceSampler, Rejector
    d = torch.device("cuda:0")
    P = N.init_params(arch, 2019, True)
    n = b * G
    z = torch.from_numpy(np.random.RandomState(4711).uniform(-1, 1, (n, N.ARCHS[arch]["z_dim"])).astype(np.float32))
    eng = RefineEngine(arch, to_device(P, d), n, d, use_graph=True, bn_groups=G)
    images_dev = eng.refine_from_z(z.to(d), K, 0.1)[0].clone()
    sig_dev = eng.score(images_dev).cpu().numpy()                       # fake_sigmoids, nsgan/GAN.py:154-155,409 -- on the GPU
    images = images_dev.cpu()
    with torch.no_grad():                                               # the oracle's D on the SAME images, batch statistics per logical batch
        sig_orc = torch.cat([torch.sigmoid(N.discriminator(arch, P, images[i:i + b])).reshape(b, -1).mean(1, keepdim=True)
                             for i in range(0, n, b)]).numpy()
    assert sig_dev.shape == sig_orc.shape == (n, 1) and sig_dev.dtype == sig_orc.dtype == np.float32
    # (a) the scores: one D forward (logits within 1e-4 of max|logit|) through a sigmoid whose slope is <= 1/4
    dmax = float(np.abs(sig_dev - sig_orc).max())
    assert dmax < 2e-5, dmax
    assert 0.0 < sig_dev.min() and sig_dev.max() < 1.0 and np.ptp(sig_orc) > 1e-3     # (a pool with spread: the masks below are not trivial)

    imgs = images.numpy()
    # (b) Rejector: rejector.py:16-38, called batch by batch as nsgan/GAN.py:314-324 does (shift_percent 100 there; the class default 60 too)
    rej_flips, rej_dp = 0, 0.0
    for shift in (100.0, 60.0):
        ra, rb = Rejector(), Rejector()
        ra.set_score_max(np.amax(sig_dev)); rb.set_score_max(np.amax(sig_orc))            # :313 (the real set's maximum there; any common start does)
        for j, i in enumerate(range(0, n, b)):
            pa, _ = Rejector.acceptance_probability(_clone(ra), sig_dev[i:i + b], 1e-8, shift)
            pb, _ = Rejector.acceptance_probability(_clone(rb), sig_orc[i:i + b], 1e-8, shift)
            np.random.seed(1000 + j); u = np.random.rand(b)
            np.random.seed(1000 + j); kept_a = ra.sampling(imgs[i:i + b], sig_dev[i:i + b], shift_percent=shift)
            np.random.seed(1000 + j); kept_b = rb.sampling(imgs[i:i + b], sig_orc[i:i + b], shift_percent=shift)
            ma, mb = ra.last_accept, rb.last_accept
            assert np.array_equal(ma, u < pa) and np.array_equal(mb, u < pb)              # (the class's own draw is the u replayed here)
            assert len(kept_a) == ma.sum() and len(kept_b) == mb.sum()
            flip = ma != mb
            lo, hi = np.minimum(pa, pb), np.maximum(pa, pb)
            assert np.all((u[flip] >= lo[flip]) & (u[flip] < hi[flip])), "a mask difference that is not a tie between the two probabilities"
            assert np.array_equal(ma[~flip], mb[~flip])
            rej_flips += int(flip.sum())
            rej_dp = max(rej_dp, float(np.abs(pa - pb).max()))
            assert abs(float(ra.D_tilde_M) - float(rb.D_tilde_M)) < 1e-3
    # the acceptance probabilities move by the score difference through logit (slope 1/(s(1-s))) and the shift by a percentile: bounded
    assert rej_dp < 2e-3, rej_dp
    assert rej_flips <= 2, rej_flips                                    # expected 2 x n x mean|dP| ~ 0.01

    # (c) MH independence chain: idpsampler.py:17-53, started as nsgan/GAN.py:398-399 does, T = 20 (nsgan/main.py), batch by batch
    T = 20
    d0 = np.mean(sig_orc)                                               # (np.mean(sigmoid_real) there: one common starting score)
    rs = np.random.RandomState(77)
    uni = rs.uniform(0, 1, size=n)
    ea, eb, flips = _mh_lockstep(sig_dev, sig_orc, d0, uni, T, b)
    for (i, u, a1, a2) in flips:
        assert min(a1, a2) <= u <= max(a1, a2), (i, u, a1, a2)          # a decision only the score difference can make
    # the classes themselves, pre-drawn uniforms (the fused evaluate loop's form) and the global stream (the reference's form)
    sa, sb = IndependenceSampler(T=T), IndependenceSampler(T=T)
    sa.set_score_curr(d0); sb.set_score_curr(d0)
    wa = [i0 + k for i0 in range(0, n, b) for k in sa.walk(sig_dev[i0:i0 + b], uni[i0:i0 + b])]
    wb = [i0 + k for i0 in range(0, n, b) for k in sb.walk(sig_orc[i0:i0 + b], uni[i0:i0 + b])]
    assert wa == ea and wb == eb                                        # (the lockstep replay is the classes' arithmetic)
    if not flips:
        assert wa == wb
        ga, gb = IndependenceSampler(T=T), IndependenceSampler(T=T)
        ga.set_score_curr(d0); gb.set_score_curr(d0)
        np.random.seed(5); xa = np.concatenate([ga.sampling(imgs[i:i + b], sig_dev[i:i + b]).reshape(-1, *imgs.shape[1:]) for i in range(0, n, b)])
        np.random.seed(5); xb = np.concatenate([gb.sampling(imgs[i:i + b], sig_orc[i:i + b]).reshape(-1, *imgs.shape[1:]) for i in range(0, n, b)])
        assert xa.shape == xb.shape and np.array_equal(xa, xb) and len(xa) > 0
    assert len(flips) <= 2, flips
    RESULTS[arch] = dict(max_sigmoid_delta=dmax, rejector_flips=rej_flips, rejector_max_dP=rej_dp, mh_primary_flips=len(flips), mh_emitted=len(wa))
    print(f"\n[accept mask {arch} {G}x{b} K={K}] max|sigmoid_dev - sigmoid_oracle| = {dmax:.3g}; Rejector: {rej_flips} flips in {2 * n} draws "
          f"(max|dP| = {rej_dp:.3g}); MH chain: {len(flips)} primary flips in {n} proposals, {len(wa)} emitted indices "
          f"{'identical' if wa == wb else 'differ after a tie'}")


def _clone(r):
    """A Rejector with the same running bound (acceptance_probability moves D_tilde_M: the probe must not move the instance under test)."""
    from cgs_amd.sampling import Rejector
    c = Rejector()
    c.D_tilde_M = np.copy(r.D_tilde_M)
    return c
