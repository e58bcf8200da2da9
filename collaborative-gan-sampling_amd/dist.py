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


def gather_pool(local, group=None, out=None, skip_trivial=False):
    """All-gather ``local`` [B, ...] from every rank into [W*B, ...], rank-major (rank r occupies rows r*B:(r+1)*B).
    One collective per pool.  Without a process group it is the identity; with one it ALWAYS runs the collective, world size
    1 included (the RCCL path a 1-GPU box can exercise is this very function) unless ``skip_trivial``.  "nccl" (= RCCL over
    xGMI) gathers the device tensors in place on the current stream; any other backend (gloo: the debug transport of
    ``bench.py --backend gloo --share-gpu`` and the CPU tests) cannot all-gather device tensors and is staged through the host.
    ``out``: a preallocated pool buffer (bench.py keeps one per batch in flight)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    W = dist.get_world_size(group)
    if W == 1 and skip_trivial:
        return local
    local = local.contiguous()
    shape = (W * local.shape[0],) + tuple(local.shape[1:])
    pool = out if out is not None else torch.empty(shape, dtype=local.dtype, device=local.device)
    if tuple(pool.shape) != shape:
        raise ValueError(f"gather_pool: pool buffer {tuple(pool.shape)} != {shape}")
    if local.is_cuda and dist.get_backend(group) != "nccl":
        host = torch.empty(shape, dtype=local.dtype)
        dist.all_gather_into_tensor(host, local.cpu(), group=group)
        pool.copy_(host)
    else:
        dist.all_gather_into_tensor(pool, local, group=group)
    return pool


def all_gather_floats(values, device=None, group=None):
    """[W, len(values)] float64 numpy array: every rank's ``values`` (timing / bookkeeping scalars of bench.py).  Device
    tensors under "nccl", host tensors otherwise."""
    import torch.distributed as dist
    W = dist.get_world_size(group)
    on_dev = dist.get_backend(group) == "nccl"
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device if on_dev else None)
    out = torch.empty(W * t.numel(), dtype=torch.float64, device=t.device)        # (flat: gloo checks the concatenated shape)
    dist.all_gather_into_tensor(out, t, group=group)
    return out.view(W, t.numel()).cpu().numpy()


def refine_pool(refine_fn, z_local, group=None):
    """Refine every local z-batch with ``refine_fn(z) -> (images, logit, step)`` and gather the pool.
    Returns (images [W*n*B, ...], logits [W*n*B], steps [W*n*B]) ordered (rank, local batch, sample)."""
    imgs, logits, steps = [], [], []
    for z in z_local:
        i, l, s = refine_fn(z)
        imgs.append(i.clone()); logits.append(l.clone()); steps.append(s.clone())
    cat = lambda xs: torch.cat(xs, dim=0)
    return gather_pool(cat(imgs), group), gather_pool(cat(logits), group), gather_pool(cat(steps), group)
