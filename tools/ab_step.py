"""Debug: wall time of the bf16 batch-512 SGD step (two-stream backward, no per-op events) under several values of a mi_set_tuning key.
usage: tools/ab_step.py KEY V0 V1 [V2 ...]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from vae.models import ConvVAE
from mi355 import lib as milib
key = int(sys.argv[1]); vals = [int(v) for v in sys.argv[2:]]
L = milib.get()
B = 512
m = ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=tempfile.mkdtemp(), precision="bf16", seed=0)
m.init_session(init_logging=False)
m.dev.ensure_batch(B)
pool = torch.rand(1024, 38400, device="cuda")
idx = torch.randperm(1024, device="cuda")[:B].to(torch.int32)
def wall(v, n=40):
    prev = L.mi_set_tuning(key, v)
    for _ in range(5):
        m._train_minibatch(pool, pool, idx, B, 1.0 / B, m._eps(B))
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        m._train_minibatch(pool, pool, idx, B, 1.0 / B, m._eps(B))
    t1.record(); torch.cuda.synchronize()
    L.mi_set_tuning(key, prev)
    return t0.elapsed_time(t1) / n
for rep in range(2):
    print("key %d: " % key + "   ".join("%d -> %.4f ms" % (v, wall(v)) for v in vals))
