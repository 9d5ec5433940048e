"""Worker for test_data_parallel_two_ranks_on_the_gpu (tests/test_d_c4_dp_gpu.py): launched by torch.distributed.run with 2 ranks that SHARE
cuda:0 over the gloo backend (RCCL refuses two ranks on one device; gloo all-reduces device tensors through the host).  Unlike
tests/dp_worker.py (CPU, oracle gradients) this runs the PRODUCT's whole data-parallel path: the native engine's decoder / encoder
halves of backward, the two asynchronous bucket all-reduces on slices of the flat gradient buffer, fused Adam, the per-epoch metric
all-reduce and the rank-0 parameter broadcast.  Writes what rank r computed to <out>/rank<r>.npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from mi355 import dist as midist  # noqa: E402
from vae.models import ConvVAE  # noqa: E402


def dataset():
    rng = np.random.RandomState(77)
    frames = (rng.randint(0, 256, (16, 80, 160, 3)).astype(np.float32) / 255.0)
    eps = [rng.standard_normal((8, 64)).astype(np.float32) for _ in range(2)]
    return frames, eps


def build(model_dir, params):
    m = ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=model_dir, precision="fp32", learning_rate=1e-4, seed=0)
    m.set_weights(params)
    m.init_session(init_logging=False)
    return m


def run(m, frames, eps):
    """One explicit global minibatch (gradients after the all-reduce), then one epoch of 2 SGD steps.  Same code for 1 and 2 ranks."""
    B = 8
    lo, hi = midist.shard_bounds(B)
    dev = m.dev
    src = m._frames(frames, 38400, "src")
    idx = torch.arange(B, device=dev.device, dtype=torch.int32)[lo:hi].contiguous()
    e = m._eps(hi - lo, eps[0][lo:hi])
    dev.forward(src, src, idx, hi - lo, 1.0 / B, e, 1, 1, accumulate_metrics=False)
    if midist.world_size() > 1:
        pending = []
        for part, lo_f, hi_f in dev.grad_buckets:       # the same bucket loop as VAE._train_minibatch
            dev.backward(src, idx, e, 1.0 / B, part=part)
            pending.append(midist.all_reduce_sum(dev.grads[lo_f:hi_f], async_op=True))
        for w in pending:
            w.wait()
        assert sorted((lo_f, hi_f) for _, lo_f, hi_f in dev.grad_buckets)[0][0] == 0 and sum(hi_f - lo_f for _, lo_f, hi_f in dev.grad_buckets) == dev.n_flat
    else:
        dev.backward(src, idx, e, 1.0 / B, part=0)
    torch.cuda.synchronize()
    grads = {k: v.copy() for k, v in dev.export_grads().items()}
    dev.grads.zero_()
    np.random.seed(0)                                   # the legacy-numpy permutation every rank draws (vae/models.py:209)
    m.train_one_epoch(frames, frames, B, eps=eps)
    losses = m.last_train_metrics
    torch.cuda.synchronize()
    return grads, np.asarray(losses, np.float64), m.dev.export_params()


def ppo_batch():
    rng = np.random.RandomState(5)
    return (rng.standard_normal((32, 67)), np.stack([rng.uniform(-1, 1, 32), rng.uniform(0, 1, 32)], 1), rng.standard_normal(32), rng.standard_normal(32))


def run_ppo(model_dir):
    """PPO.train on a global minibatch of 32: every rank passes ITS 16 rows (ppo.py's data-parallel contract); 3 steps; losses and parameters."""
    from ppo import PPO

    class Box:
        low, high, shape = np.array([-1.0, 0.0], np.float32), np.array([1.0, 1.0], np.float32), (2,)
    m = PPO(np.array([67]), Box(), model_dir=model_dir, seed=3)
    m.init_session(init_logging=False)
    s, a, r, adv = ppo_batch()
    lo, hi = midist.shard_bounds(32)
    out = []
    for _ in range(3):
        L = m.train_step(s[lo:hi], a[lo:hi], r[lo:hi], adv[lo:hi])
        out.append([L["policy_loss"], L["value_loss"], L["entropy_loss"], L["loss"], L["prob_ratio"]])
    torch.cuda.synchronize()
    return np.asarray(out, np.float64), m.dev.params.cpu().numpy().copy()


def main(out, backend="gloo"):
    world, rank, local = midist.init_from_env(backend)
    torch.cuda.set_device(local if backend == "nccl" else 0)     # nccl: one device per rank (RCCL over xGMI); gloo: both ranks share cuda:0
    import vae_gpu_common as T
    frames, eps = dataset()
    m = build(os.path.join(out, "model_rank%d" % rank), T.trained_like_params(2))
    grads, losses, params = run(m, frames, eps)
    ppo_losses, ppo_params = run_ppo(os.path.join(out, "ppo_rank%d" % rank))
    np.savez(os.path.join(out, "rank%d.npz" % rank), losses=losses, world=world, comm=np.array(midist.comm_note), ppo_losses=ppo_losses, ppo_params=ppo_params,
             **{"g|" + k.replace("/", "|"): v for k, v in grads.items()}, **{"p|" + k.replace("/", "|"): v for k, v in params.items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "gloo")
