#!/usr/bin/env python3
"""What the DATA-PARALLEL form of the ConvVAE step costs one rank WITHOUT its collectives: mi_vae_train_step (one backward pass) against mi_vae_train_step_dp on a recording communicator
(the three backward parts in bucket order, each bucket handed to the communicator -- which only records it --, the join, Adam), batch 512 bf16, one GPU, interleaved rounds.
    python tools/dp_step_time.py [--steps 200] [--rounds 3]"""
import argparse, ctypes, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=200); ap.add_argument("--rounds", type=int, default=3); ap.add_argument("--batch", type=int, default=512)
args = ap.parse_args()
from vae.models import ConvVAE, adam_alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON
B = args.batch
model = ConvVAE(np.array([80, 160, 3]), z_dim=64, beta=1.0, learning_rate=1e-4, model_dir=os.path.join(tempfile.mkdtemp(prefix="mi355_dp_"), "vae"), precision="bf16", seed=0)
model.init_session(init_logging=False)
dev = model.dev; dev.ensure_batch(B); L = dev.L
g = torch.Generator(device=dev.device); g.manual_seed(1234)
pool = torch.randint(0, 256, (2048, 38400), device=dev.device, generator=g, dtype=torch.int32).to(torch.uint8).contiguous()
idx = torch.stack([torch.randperm(2048, device=dev.device, generator=g)[:B] for _ in range(32)]).to(torch.int32).contiguous()
h = ctypes.c_void_p(); log = np.zeros((64, 4), np.int64)
L.mi_comm_init_recording(ctypes.addressof(h), 0, 1, log.ctypes.data, 64)
alpha = adam_alpha(1e-4, np.float32(0.9), np.float32(0.999))
single = lambda i: dev.train_step(pool, pool, idx[i % 32], B, 1.0 / B, None, alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON)
dp = lambda i: dev.train_step_dp(h, pool, pool, idx[i % 32], B, 1.0 / B, None, alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON)
for i in range(400):
    single(i)
res = {"single": [], "dp": []}
for r in range(args.rounds):
    for name, f in (("single", single), ("dp", dp)):
        for i in range(20):
            f(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(args.steps):
            f(i)
        torch.cuda.synchronize(); res[name].append((time.perf_counter() - t0) / args.steps * 1e3)
L.mi_comm_destroy(h)
for k, v in res.items():
    print("%-8s ms per step: median %.4f  %s" % (k, sorted(v)[len(v) // 2], [round(x, 4) for x in v]))
