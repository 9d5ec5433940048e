"""BASELINE configs[4] (SURVEY "C5"): end-to-end synthetic replay -- VAE encode + GAE + PPO minibatch SGD over many recorded trajectories.

    python tools/replay_synthetic.py [--rows 1024] [--steps 128] [--batch 2048] [--epochs 4]            (1 GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/replay_synthetic.py ...

Rows (trajectories) are sharded over the ranks; every rank generates its own synthetic uint8 frames (replay.replay_update, local_rows=True).
Prints one JSON line: frames (= samples) per second of the whole job for the encode stage, the PPO stage and end to end.  Reported next
to bench.py's headline metric, not part of it (the headline configuration is configs[1]: the ConvVAE SGD step).
Not yet run on hardware: written after round 1's GPU minutes were spent."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--batch", type=int, default=2048, help="global PPO minibatch")
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    args = ap.parse_args()
    from mi355 import dist as midist
    world, rank, local = midist.init_from_env("nccl")
    torch.cuda.set_device(local)
    import replay
    from ppo import PPO
    from vae.models import ConvVAE

    class Box:
        low, high, shape = np.array([-1.0, 0.0], np.float32), np.array([1.0, 1.0], np.float32), (2,)
    tmp = tempfile.mkdtemp(prefix="mi355_replay_")
    vae = ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=os.path.join(tmp, "vae"), precision=args.precision, training=False, seed=0)
    vae.init_session(init_logging=False)
    ppo = PPO(np.array([67]), Box(), learning_rate=1e-4, lr_decay=1.0, epsilon=0.2, value_scale=1.0, entropy_scale=0.01, initial_std=1.0,
              model_dir=os.path.join(tmp, "ppo"))
    ppo.init_session(init_logging=False)
    T = args.steps
    lo, hi = midist.shard_bounds(args.rows, rank, world)
    R = hi - lo                                                # every rank generates (and owns) its own trajectories
    rng = np.random.RandomState(1234 + rank)
    frames = rng.randint(0, 256, (R, T + 1, 80, 160, 3), dtype=np.uint8)
    meas = np.stack([rng.uniform(-1, 1, (R, T + 1)), rng.uniform(0, 1, (R, T + 1)), rng.uniform(0, 30, (R, T + 1))], axis=-1).astype(np.float32)
    actions = np.stack([rng.uniform(-1, 1, (R, T)), rng.uniform(0, 1, (R, T))], axis=-1).astype(np.float32)
    rewards, dones = rng.uniform(0, 1, (R, T)), np.zeros((R, T))

    def once():
        midist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = replay.replay_update(vae, ppo, frames, meas, actions, rewards, dones, 0.99, 0.95, args.epochs, args.batch, local_rows=True)
        torch.cuda.synchronize(); midist.barrier()
        return time.perf_counter() - t0, out
    once()                                                     # warm-up (engines sized, RCCL communicator created)
    dt, out = once()
    t = torch.tensor([dt], device="cuda", dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    if rank == 0:
        n = args.rows * T
        print(json.dumps({"config": "synthetic replay: %d trajectories x %d steps, PPO minibatch %d x %d epochs, %s VAE encode" % (args.rows, T, args.batch, args.epochs, args.precision),
                          "n_gpus": world, "seconds": float(t.item()), "samples_per_s": n / float(t.item()), "sgd_steps": len(out["losses"]),
                          "last_loss": out["losses"][-1] if out["losses"] else None, "includes": "host->device upload of the uint8 frames, encode, values, GAE, PPO SGD"}))
    midist.barrier()
    if world > 1:
        midist.shutdown()               # the library's own RCCL communicator first, then the process group it was bootstrapped from
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
