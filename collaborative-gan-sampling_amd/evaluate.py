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
        return engine.score(x).cpu().numpy()                 # float32 [B, 1]: what sess.run(fake_sigmoids) hands the chain (nsgan/GAN.py:409)

    return propose, score


# ------------------------------------------------------------------------------------------------------------------------------
# The same fill loop with G logical batches per device round (SURVEY.md 8f-1 on the fast path).
#
# nsgan/GAN.py:398-426 runs one batch_size-64 batch per iteration (nsgan/main.py:32): sess.run(g_refine_detem), sess.run(fake_sigmoids),
# the host chain.  At 64 samples every launch is latency-bound.  Here a ROUND is G such iterations: their z batches are drawn in the
# reference's order, refined in ONE engine call (``bn_groups`` = G: D's batch statistics per logical batch, nsgan/GAN.py:175), scored on
# the device the same way, and copied to pinned host memory once; the host chain walks round r while the device refines round r + 1.
# The global numpy stream is consumed exactly as the reference's loop consumes it -- per iteration the z draw, then one uniform per
# proposal for the chain (drawn ahead, handed to ``IndependenceSampler.walk``) -- and is rewound at the end to where the reference's
# loop would have left it, so the accepted set, the efficiency and every later np.random draw are those of the one-batch-at-a-time loop.
class FusedProposer:
    """G logical batches of ``batch`` samples per round on a ``model.GAN``'s engines; ``depth`` rounds in flight (one engine + HIP
    stream + pinned result buffers each)."""

    def __init__(self, gan, steps, rate, batch=None, groups=32, depth=2, method="momentum", mode="deterministic", contraction="f32",
                 use_graph=True, indices=None):
        import torch
        from .engine import RefineEngine
        self.torch = torch
        self.b, self.G = int(batch or gan.batch_size), int(groups)
        self.steps, self.rate, self.method, self.mode = steps, rate, method, mode
        self.dev = gan.device
        A = gan.A
        self.zdim = A["z_dim"]
        n = self.b * self.G
        P = gan.build_variables()
        # (own engines rather than gan.engine()'s cached one: every round in flight needs its own activation buffers)
        self.engines = [RefineEngine(A, P, n, self.dev, use_graph=use_graph, bn_groups=self.G, contraction=contraction) for _ in range(depth)]
        self.streams = [torch.cuda.Stream(self.dev) for _ in range(depth)]
        img = (n,) + tuple(A["img"])
        self.z_host = [torch.empty((n, self.zdim), dtype=torch.float32).pin_memory() for _ in range(depth)]
        self.z_dev = [torch.empty((n, self.zdim), dtype=torch.float32, device=self.dev) for _ in range(depth)]
        self.img_host = [torch.empty(img, dtype=torch.float32).pin_memory() for _ in range(depth)]
        self.sig_host = [torch.empty((n, 1), dtype=torch.float32).pin_memory() for _ in range(depth)]
        self.sig_dev = [torch.empty((n, 1), dtype=torch.float32, device=self.dev) for _ in range(depth)]
        self.done = [torch.cuda.Event() for _ in range(depth)]
        self.depth, self.launched = depth, 0
        self.indices = None
        if mode == "probabilistic":      # the reference bakes ONE draw of size batch_size into the graph, for every batch (collaborator.py:54-56)
            one = np.asarray(indices) if indices is not None else np.random.randint(steps + 1, size=self.b)
            self.indices = np.tile(one, self.G)

    def launch(self, z, refine=True):
        """Queue one round on the next slot: z [G*b, z_dim] (host array) -> images + sigmoids in the slot's pinned buffers.
        ``refine=False``: the standard samples instead (fake_images / fake_sigmoids, nsgan/GAN.py:153-155).  Returns the slot."""
        torch = self.torch
        k = self.launched % self.depth
        self.launched += 1
        self.z_host[k].numpy()[...] = z
        e, st = self.engines[k], self.streams[k]
        with torch.cuda.stream(st):
            self.z_dev[k].copy_(self.z_host[k], non_blocking=True)
            if refine:
                kw = dict(mode="probabilistic", indices=self.indices) if self.mode == "probabilistic" else {}
                img = e.refine_from_z(self.z_dev[k], self.steps, self.rate, method=self.method, **kw)[0]
            else:
                img = e.generate(self.z_dev[k])
            self.img_host[k].copy_(img, non_blocking=True)        # (before score(): D's forward re-uses no G buffer, but keep the order plain)
            e.score(img, out=self.sig_dev[k])
            self.sig_host[k].copy_(self.sig_dev[k], non_blocking=True)
            self.done[k].record(st)
        return k

    def result(self, k):
        """Wait for slot k -> (images [G*b, ...], sigmoids [G*b, 1]) float32 host arrays (views of the pinned buffers: valid until
        the slot is launched again)."""
        self.done[k].synchronize()
        return self.img_host[k].numpy(), self.sig_host[k].numpy()

    def score_real(self, x, chunk=None):
        """np.mean-able sigmoids of a real evaluation set, scored ``batch`` samples at a time like nsgan/GAN.py:388-390
        (``G`` batches per launch; x.shape[0] must be a multiple of ``batch``; a last partial round is padded with repeats
        of whole batches, whose scores are dropped)."""
        torch = self.torch
        n = self.b * self.G
        N = x.shape[0]
        if N % self.b:
            raise ValueError(f"{N} real samples are not whole batches of {self.b}")
        out = np.empty((N, 1), dtype=np.float32)
        e = self.engines[0]
        for i in range(0, N, n):
            xb = np.asarray(x[i:i + n], dtype=np.float32)
            m = xb.shape[0]
            if m < n:
                xb = np.concatenate([xb] + [xb[:self.b]] * ((n - m) // self.b))
            with torch.cuda.stream(self.streams[0]):
                s = e.score(torch.from_numpy(np.ascontiguousarray(xb)).to(self.dev))
                out[i:i + m] = s.cpu().numpy()[:m]
        return out


def collaborate_fused(proposer, mh_sampler, eval_size, real_sigmoid_mean, base=None, min_efficiency=None, max_rounds=100000, stats=None):
    """``collaborate`` (count_only_productive=False: the nsgan form, nsgan/GAN.py:398-426) over a ``FusedProposer``.
    ``stats`` (dict, optional) receives rounds / proposals / host-chain seconds / device-wait seconds.
    Returns (samples [eval_size, ...], efficiency) -- bit-identical to ``collaborate`` fed by one-batch-at-a-time proposals of the
    same engine arithmetic, global numpy stream included."""
    import time
    b, G = proposer.b, proposer.G
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
    t_chain = t_wait = 0.0
    drawn = 0                                          # logical batches drawn so far: batch i meets cnt_propose = eval_size + i*b
    pending = []                                       # rounds in flight: (slot, uniforms per logical batch, RNG state after each)

    def launch_round():
        nonlocal drawn
        zs, us, states = [], [], []
        for _ in range(G):
            zs.append(np.random.uniform(-1, 1, [b, proposer.zdim]).astype(np.float32))          # nsgan/GAN.py:408
            through_chain = eval_size + drawn * b < max_propose                                 # :410, known ahead: +b per iteration (:426)
            us.append(np.random.uniform(0, 1, size=b) if through_chain else None)               # idpsampler.py:50, one per proposal
            states.append(np.random.get_state())
            drawn += 1
        pending.append((proposer.launch(np.concatenate(zs)), us, states))

    rounds = 0
    final_state = None
    while cnt < eval_size:
        rounds += 1
        if rounds > max_rounds:
            raise RuntimeError("collaborate_fused: no sample accepted in %d rounds" % max_rounds)
        while len(pending) < proposer.depth:
            launch_round()
        slot, us, states = pending.pop(0)
        t0 = time.perf_counter()
        imgs, sigs = proposer.result(slot)
        t1 = time.perf_counter()
        t_wait += t1 - t0
        if out is None:
            out = np.empty((eval_size,) + imgs.shape[1:], dtype=np.float32)
        for j in range(G):
            batch = imgs[j * b:(j + 1) * b]
            if cnt_propose < max_propose:
                # (a private copy of the batch's scores: the chain keeps a VIEW of the score it moved to -- idpsampler.py:52 -- and the
                # pinned buffer behind ``sigs`` is overwritten when its slot is launched again)
                rows = mh_sampler.walk(sigs[j * b:(j + 1) * b].copy(), us[j])
                n = len(rows)
                if n > 0:
                    k = min(n, eval_size - cnt)
                    out[cnt:cnt + k] = batch[rows[:k]]
                cnt += n
            else:                                                                   # too inefficient: take the batch as is
                k = min(b, eval_size - cnt)
                out[cnt:cnt + k] = batch[:k]
                cnt += b
            cnt_propose += b
            if cnt >= eval_size:
                final_state = states[j]
                break
        t_chain += time.perf_counter() - t1
    for slot, _, _ in pending:                         # rounds drawn ahead that the reference's loop never reaches: drained, discarded
        proposer.result(slot)
    if final_state is not None:
        np.random.set_state(final_state)               # the global stream where the one-batch-at-a-time loop leaves it
    if stats is not None:
        stats.update(rounds=rounds, proposed=cnt_propose - eval_size, host_chain_s=t_chain, device_wait_s=t_wait,
                     discarded_batches=drawn - (cnt_propose - eval_size) // b)
    return out, cnt / cnt_propose
