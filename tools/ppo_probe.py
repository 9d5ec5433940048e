"""PPO update timing probe (BASELINE configs[2]): horizon 128, 4 epochs x 4 minibatches of 32, fp32; also one big-minibatch step."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from ppo import PPO

class Box:
    low, high, shape = np.array([-1.0, 0.0], np.float32), np.array([1.0, 1.0], np.float32), (2,)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = PPO(np.array([67]), Box(), learning_rate=1e-4, lr_decay=1.0, epsilon=0.2, value_scale=1.0, entropy_scale=0.01, initial_std=1.0, model_dir=tempfile.mkdtemp())
m.init_session(init_logging=False)
if hasattr(m.dev, "ensure_batch"):
    m.dev.ensure_batch(M)
dev = m.dev.device
rng = np.random.RandomState(7)
s = torch.from_numpy((0.5 * rng.standard_normal((M, 67))).astype(np.float32)).to(dev)
a = torch.from_numpy(rng.uniform(-1, 1, (M, 2)).astype(np.float32)).to(dev)
R = torch.from_numpy(rng.randn(M).astype(np.float32)).to(dev)
A = torch.from_numpy(rng.randn(M).astype(np.float32)).to(dev)
m.update_old_policy()
for _ in range(5):
    m._step_resident(s, a, R, A, M, M)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 50
for _ in range(n):
    m._step_resident(s, a, R, A, M, M)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("PPO SGD step at minibatch %d: %.1f us  (host enqueue %.1f us per step: below the total = the GPU is the limit, equal = the launches are)"
      % (M, (time.perf_counter() - t0) / n * 1e6, (t1 - t0) / n * 1e6))
