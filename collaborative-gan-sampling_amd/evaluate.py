"""Collaborative sampling = refine, then accept/reject: the step right after the hot path
(nsgan/GAN.py:398-426 "collaborate", synthetic/main.py:232-253).

``collaborate`` is the fill loop both reference drivers run: proposals come from the refiner, their discriminator
scores drive the MH independence chain (``IndependenceSampler``, thinning T=20), and accepted samples fill an
evaluation set of ``eval_size``.  The proposals here come from the device engine (one refine call per batch, images
and sigmoids read back once per batch); the chain is host code exactly like the reference's.
"""
import numpy as np

MIN_EFFICIENCY = 0.2     # nsgan/GAN.py:18


def collaborate(propose, score, mh_sampler, eval_size, real_sigmoid_mean, base=None, min_efficiency=None,
                count_only_productive=False, max_rounds=100000):
    """Fill ``eval_size`` accepted samples.

    propose()            -> refined batch as ndarray [B, ...]           (g_refine_detem / manipulate_sample)
    score(batch)         -> discriminator sigmoids [B, 1]               (fake_sigmoids)
    base                 -> optional (samples, sigmoids) for the first MH pass (nsgan uses the *standard* samples
                            there, nsgan/GAN.py:402, quirk Q10; synthetic uses the refined evaluation batch)
    min_efficiency       -> nsgan/GAN.py:283,410: once proposals exceed eval_size / min_efficiency, stop rejecting
                            and take whole batches
    count_only_productive-> synthetic/main.py:251: the proposal counter only advances for batches that yielded samples
    Returns (samples [eval_size, ...], efficiency = accepted / proposed)."""
    out, cnt, cnt_propose = None, 0, eval_size
    mh_sampler.set_score_curr(real_sigmoid_mean)                                   # nsgan/GAN.py:401
    if base is not None:
        acc = mh_sampler.sampling(base[0], base[1])
        if acc.shape[0] > 0:
            out = np.empty((eval_size,) + acc.shape[1:], dtype=acc.dtype)
            k = min(acc.shape[0], eval_size)
            out[:k] = acc[:k]
        cnt = acc.shape[0]
    max_propose = eval_size / min_efficiency if min_efficiency else float("inf")
    rounds = 0
    while cnt < eval_size:
        rounds += 1
        if rounds > max_rounds:
            raise RuntimeError("collaborate: no sample accepted in %d rounds" % max_rounds)
        batch = propose()
        B = batch.shape[0]
        if out is None:
            out = np.empty((eval_size,) + batch.shape[1:], dtype=np.float32)
        if cnt_propose < max_propose:
            acc = mh_sampler.sampling(batch, score(batch))
            n = acc.shape[0]
            if n > 0:
                k = min(n, eval_size - cnt)
                out[cnt:cnt + k] = acc[:k]
            cnt += n
            if n > 0 or not count_only_productive:
                cnt_propose += B
        else:                                                                       # too inefficient: take the batch as is
            k = min(B, eval_size - cnt)
            out[cnt:cnt + k] = batch[:k]
            cnt += B
            cnt_propose += B
    return out, cnt / cnt_propose


def engine_proposer(engine, steps, rate, rng=None, **refine_kw):
    """propose() / score() closures over a ``RefineEngine``: z ~ U(-1,1) (nsgan/GAN.py:408), refine on device,
    read the images back; score = sigmoid(D(images)) with D on batch statistics (nsgan/GAN.py:154-155)."""
    import torch
    rng = rng or np.random
    zdim = engine.A["z_dim"]

    def propose():
        z = torch.from_numpy(rng.uniform(-1, 1, [engine.B, zdim]).astype(np.float32)).to(engine.dev)
        return engine.refine_from_z(z, steps, rate, **refine_kw)[0].cpu().numpy()

    def score(batch):
        x = torch.from_numpy(np.ascontiguousarray(batch, dtype=np.float32)).to(engine.dev)
        logits = engine.discriminator(x)
        return torch.sigmoid(logits).reshape(len(batch), -1).mean(1, keepdim=True).cpu().numpy().astype(np.float64)

    return propose, score
