#!/usr/bin/env python3
"""the MlpVAE (reference vae/models.py:271-299) SGD step at batch 512, both precisions, with per-launch kernel times when run under
rocprofv3 (tools/gpu_prof.sh).   python tools/mlp_vae_bench.py [--steps 30]"""
import argparse, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from vae.models import MlpVAE

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=30); ap.add_argument("--batch", type=int, default=512); ap.add_argument("--precision", default="bf16,fp32"); ap.add_argument("--frames", default="u8", choices=["u8", "f32"], help="frame table in HBM: uint8 camera bytes (round 5) or float32 k / 255")
args = ap.parse_args()
B = args.batch
rng = np.random.RandomState(0)
frames_u8 = rng.randint(0, 256, (1024, 80, 160, 3), dtype=np.uint8)
frames = frames_u8 if args.frames == "u8" else frames_u8.astype(np.float32) / 255.0
for precision in args.precision.split(","):
    m = MlpVAE(np.array([80, 160, 3]), z_dim=64, model_dir=tempfile.mkdtemp(), precision=precision, seed=0)
    m.init_session(init_logging=False)
    dev = m._need_dev()
    if hasattr(dev, "ensure_batch"):
        dev.ensure_batch(B)
    src = m._frames(frames, int(np.prod([80, 160, 3])), "source_states", keep_u8_ok=True)
    idx = torch.randperm(1024, device=src.device)[:B].to(torch.int32)
    for _ in range(5):
        m._train_minibatch(src, src, idx, B, 1.0 / B, m._eps(B))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m._train_minibatch(src, src, idx, B, 1.0 / B, m._eps(B))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    flops = 6.0 * B * (38400 * 512 + 512 * 256 + 256 * 128 + 64 * 256 + 256 * 512 + 512 * 38400)      # fwd + dgrad + wgrad of every dense layer
    print("MlpVAE %s (%s table) SGD step at batch %d: %.3f ms = %.0f frames/s, %.1f TFLOP/s of dense-layer work" % (precision, src.dtype, B, dt * 1e3, B / dt, flops / dt / 1e12))
