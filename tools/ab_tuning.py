"""Debug: per-op timing of one bf16 SGD step at batch 512 under two values of a mi_set_tuning key.
usage: tools/ab_tuning.py KEY V0 V1 [op-name-suffix]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from vae.models import ConvVAE
from mi355 import lib as milib
key, v0, v1 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
flt = sys.argv[4] if len(sys.argv) > 4 else ""
L = milib.get()
B = 512
m = ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=tempfile.mkdtemp(), precision="bf16", seed=0)
m.init_session(init_logging=False)
dev = m.dev; dev.ensure_batch(B)
pool = torch.rand(1024, 38400, device="cuda")
idx = torch.randperm(1024, device="cuda")[:B].to(torch.int32)
n_ops = L.mi_vae_op_count(); names = [L.mi_vae_op_name(i).decode() for i in range(n_ops)]
def timing(v):
    prev = L.mi_set_tuning(key, v)
    for _ in range(3):
        m._train_minibatch(pool, pool, idx, B, 1.0 / B, m._eps(B))
    torch.cuda.synchronize()
    L.mi_vae_timing_begin(dev.handle, 1, -1, 8 * n_ops + 8)
    for _ in range(6):
        m._train_minibatch(pool, pool, idx, B, 1.0 / B, m._eps(B))
    torch.cuda.synchronize()
    ms = np.zeros(n_ops, np.float32); cnt = np.zeros(n_ops, np.int32)
    L.mi_vae_timing_collect(dev.handle, ms.ctypes.data, cnt.ctypes.data, n_ops)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        m._train_minibatch(pool, pool, idx, B, 1.0 / B, m._eps(B))
    t1.record(); torch.cuda.synchronize()
    L.mi_set_tuning(key, prev)
    return {names[i]: 1e3 * ms[i] / cnt[i] for i in range(n_ops) if cnt[i]}, t0.elapsed_time(t1) / 20
(a, ta), (b, tb) = timing(v0), timing(v1)
print("key %d: step %.4f ms (value %d) -> %.4f ms (value %d)" % (key, ta, v0, tb, v1))
for k in a:
    if k.endswith(flt) and abs(a[k] - b.get(k, 0)) > 0.02 * a[k]:
        print("  %-16s %7.1f -> %7.1f us" % (k, a[k], b.get(k, float("nan"))))
