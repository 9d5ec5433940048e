"""Worker for test_data_parallel_two_ranks_gloo_equals_single_process (launched by torch.distributed.run, gloo, CPU).

Exercises the PRODUCT's data-parallel plumbing (mi355.dist: env init, shard bounds, flat all-reduce, broadcast) with the
oracle standing in for the per-rank gradient engine (the HIP engine needs a GPU): each rank takes its rows of the global
minibatch, computes gradients of sum_local/B_global, the flat buffer is summed over ranks, every rank applies the same
TF-Adam step.  Compared with a single process on the whole minibatch."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from mi355 import dist as midist  # noqa: E402
from oracle import ppo_oracle as po  # noqa: E402
from oracle import vae_oracle as vo  # noqa: E402


def flat(d):
    return torch.from_numpy(np.concatenate([np.asarray(v, np.float32).reshape(-1) for v in d.values()]))


def main(out):
    torch.set_num_threads(2)
    world, rank, _ = midist.init_from_env("gloo")
    assert world == 2 and midist.world_size() == 2 and midist.rank() == rank
    B = 6
    frames = np.random.RandomState(1234).randint(0, 256, (16, 80, 160, 3), dtype=np.uint8).astype(np.float32) / 255.0
    params = vo.init_vae_params(0)
    np.random.seed(0)                                   # same legacy-numpy permutation on every rank
    indices = np.arange(len(frames)); np.random.shuffle(indices)
    mb = indices[:B]
    eps = np.random.RandomState(4321).standard_normal((B, 64)).astype(np.float32)
    # single process, global minibatch
    (recon_s, kl_s, _), g_single, _ = vo.vae_loss_and_grads(params, frames[mb], frames[mb], eps)
    # this rank's rows; loss scaled by local/global so that the SUM over ranks is the global mean
    lo, hi = midist.shard_bounds(B)
    mine = midist.shard(mb)
    assert np.array_equal(mine, mb[lo:hi])
    (recon_l, kl_l, _), g_local, _ = vo.vae_loss_and_grads(params, frames[mine], frames[mine], eps[lo:hi])
    scale = (hi - lo) / B
    gflat = flat(g_local) * scale
    w = midist.all_reduce_sum(gflat, async_op=True)
    w.wait()
    gref = flat(g_single)
    max_rel = float((gflat - gref).abs().max() / gref.abs().max())
    metrics = torch.tensor([recon_l * scale, kl_l * scale, scale], dtype=torch.float64)
    midist.all_reduce_sum(metrics)
    # identical Adam on every rank from the reduced gradient
    names = list(params)
    sizes = [params[k].size for k in names]
    chunks = torch.split(gflat, sizes)
    p_dp = {k: v.copy() for k, v in params.items()}
    adam = vo.AdamTF({k: v.shape for k, v in params.items()})
    adam.step(p_dp, {k: c.numpy().reshape(params[k].shape) for k, c in zip(names, chunks)}, 1e-4)
    p_single = {k: v.copy() for k, v in params.items()}
    vo.AdamTF({k: v.shape for k, v in params.items()}).step(p_single, g_single, 1e-4)
    mine_flat = flat(p_dp)
    other = mine_flat.clone()
    midist.broadcast(other, src=0)
    between = float((other - mine_flat).abs().max())
    upd_ref = flat(p_single) - flat(params)
    upd = mine_flat - flat(params)
    rel_p = float((upd - upd_ref).abs().max() / upd_ref.abs().max())

    # PPO rows shard the same way; the state-independent entropy gradient is shared through grad_scale = local/global
    space = po.ActionSpace()
    m = po.OraclePPO([67], space, seed=1, initial_std=1.0)
    rng = np.random.RandomState(5)
    for k in m.params:
        m.params[k] = m.params[k] + (0.02 * rng.standard_normal(m.params[k].shape)).astype(np.float32)
    M = 10
    s = (0.5 * rng.standard_normal((M, 67))).astype(np.float32); a = rng.uniform(-1, 1, (M, 2)).astype(np.float32)
    R, A = rng.randn(M).astype(np.float32), rng.randn(M).astype(np.float32)
    _, gs = m.loss_and_grads(s, a, R, A)
    lo, hi = midist.shard_bounds(M)
    _, gl = m.loss_and_grads(s[lo:hi], a[lo:hi], R[lo:hi], A[lo:hi])
    gl_flat = flat(gl) * ((hi - lo) / M)
    midist.all_reduce_sum(gl_flat)
    ppo_rel = float((gl_flat - flat(gs)).abs().max() / flat(gs).abs().max())
    midist.barrier()
    if rank == 0:
        json.dump({"world": world, "max_grad_rel_err": max_rel, "max_param_diff_between_ranks": between, "max_param_rel_err_vs_single": rel_p,
                   "metric_recon_dp": float(metrics[0] / metrics[2]), "metric_recon_single": recon_s, "ppo_max_grad_rel_err": ppo_rel},
                  open(os.path.join(out, "result.json"), "w"))
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
