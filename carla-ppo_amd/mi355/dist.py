"""Data parallelism: one process per GPU, gradients all-reduced with torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The reference has no distributed code at all — this is the new
component BASELINE configs 4-5 ask for.

Scheme: every rank draws the SAME host permutation (same legacy-numpy seed), takes rows
[r*B/W, (r+1)*B/W) of each global minibatch, computes gradients of  sum_local(loss_i) / B_global,  and the flat fp32
gradient buffer is summed across ranks.  Parameters stay replicated: broadcast once from rank 0, then identical
fused-Adam steps everywhere.  No other exchange is needed: VAE encode, GAE and per-row advantage normalisation are
row-local (PPO rows shard across ranks with no collective).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(backend=None):
    """Initialise the default process group from torchrun-style env vars; no-op for a single process."""
    world, rank, local = env_world()
    if world <= 1:
        return 1, 0, 0
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_bounds(n, r=None, w=None):
    """Contiguous [lo, hi) slice of n minibatch rows owned by rank r of w (n need not divide evenly)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    base, rem = divmod(n, w)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


def shard(mb_idx, r=None, w=None):
    lo, hi = shard_bounds(len(mb_idx), r, w)
    return mb_idx[lo:hi]


def all_reduce_sum(t, async_op=False):
    """In-place sum over ranks of a flat tensor (gradient bucket / metric accumulators)."""
    if world_size() == 1:
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)


def broadcast(t, src=0):
    if world_size() > 1:
        dist.broadcast(t, src=src)


def barrier():
    if world_size() > 1:
        dist.barrier()
