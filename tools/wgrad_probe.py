"""Debug: per-op timing of one bf16 SGD step with / without the wgrad atomics (mi_set_tuning key 2)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from vae.models import ConvVAE
from mi355 import lib as milib
L = milib.get()
B = 512
m = ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=tempfile.mkdtemp(), precision="bf16", seed=0)
m.init_session(init_logging=False)
dev = m.dev; dev.ensure_batch(B)
pool = torch.rand(1024, 38400, device="cuda")
idx = torch.randperm(1024, device="cuda")[:B].to(torch.int32)
n_ops = L.mi_vae_op_count(); names = [L.mi_vae_op_name(i).decode() for i in range(n_ops)]
def timing(skip):
    L.mi_set_tuning(2, skip)
    for _ in range(3):
        m._train_minibatch(pool, pool, idx, B, 1.0 / B, m._eps(B))
    torch.cuda.synchronize()
    L.mi_vae_timing_begin(dev.handle, 1, -1, 4 * n_ops + 8)
    for _ in range(3):
        m._train_minibatch(pool, pool, idx, B, 1.0 / B, m._eps(B))
    torch.cuda.synchronize()
    ms = np.zeros(n_ops, np.float32); cnt = np.zeros(n_ops, np.int32)
    L.mi_vae_timing_collect(dev.handle, ms.ctypes.data, cnt.ctypes.data, n_ops)
    return {names[i]: 1e3 * ms[i] / cnt[i] for i in range(n_ops) if cnt[i]}
a, b = timing(0), timing(1)
L.mi_set_tuning(2, 0)
print("wgrad op: us with atomics -> us without")
for k in a:
    if k.endswith("wgrad"):
        print("  %-16s %7.1f -> %7.1f" % (k, a[k], b[k]))
