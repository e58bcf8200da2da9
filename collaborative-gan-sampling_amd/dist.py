"""Multi-GPU: independent z-batches per rank + one collective for the refined sample pool.

The refinement of a z-batch needs nothing from any other batch (frozen replicated weights, optimizer state
reset per call: sampling/collaborator.py:86), so the path shards at batch granularity with NO data-path
collective: one process per GPU, rank r draws its z from RandomState(2019 + r) and refines batches
r, r+W, r+2W, ... .  The only exchange is the final gather of refined images (and their logits / steps)
into the node-wide pool that the host-side accept/reject samplers consume: one all-gather per pool
(``torch.distributed`` backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests).

Caveat: D normalises with BATCH statistics, so a logical batch of W*B samples split W ways is
W independent B-batches, not one W*B batch.  That is the default semantics here (and of the reference run at
batch B).  ``RefineEngine(..., sync_bn=True)`` makes the W shards ONE logical batch instead: every bn pass of D
all-reduces its 2*C per-channel sums (double) over the group between the two halves of the pass
(cgs_bn_sync_*; tests/test_gpu_sync_bn.py checks two ranks against the unsplit batch).
"""
import numpy as np
import torch

BASE_SEED = 2019          # nsgan/main.py:13-16, synthetic/main.py:22-24


def rank_seed(rank, base=BASE_SEED):
    return base + int(rank)


def z_batches(rank, n_batches, batch_size, z_dim, base=BASE_SEED):
    """The rank's proposal noise, z ~ U(-1,1) (nsgan/GAN.py:290), as float32 [n_batches, B, z_dim]."""
    rs = np.random.RandomState(rank_seed(rank, base))
    return rs.uniform(-1, 1, (n_batches, batch_size, z_dim)).astype(np.float32)


def batch_owner(global_batch_index, world_size):
    """Round-robin ownership of global batch indices: batch i is refined by rank i % W as its local batch i // W."""
    return global_batch_index % world_size, global_batch_index // world_size


def gather_pool(local, group=None):
    """All-gather ``local`` [B, ...] from every rank into [W*B, ...], rank-major (rank r occupies rows r*B:(r+1)*B).
    One collective; with world size 1 (or no process group) it is the identity."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    W = dist.get_world_size(group)
    local = local.contiguous()
    pool = torch.empty((W * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(pool, local, group=group)
    return pool


def refine_pool(refine_fn, z_local, group=None):
    """Refine every local z-batch with ``refine_fn(z) -> (images, logit, step)`` and gather the pool.
    Returns (images [W*n*B, ...], logits [W*n*B], steps [W*n*B]) ordered (rank, local batch, sample)."""
    imgs, logits, steps = [], [], []
    for z in z_local:
        i, l, s = refine_fn(z)
        imgs.append(i.clone()); logits.append(l.clone()); steps.append(s.clone())
    cat = lambda xs: torch.cat(xs, dim=0)
    return gather_pool(cat(imgs), group), gather_pool(cat(logits), group), gather_pool(cat(steps), group)
