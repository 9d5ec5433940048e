"""Latency of the rollout-loop inference path at batch 1 (SURVEY 8f.3 callers: vae_common.py:45-61, train.py:142, run_eval.py:54):
VAE.encode([frame]) (host frame -> device, conv stack, mean to host) followed by PPO.predict(state) (host -> device, two MLP trunks, action to host)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from vae.models import ConvVAE
from ppo import PPO

class Box:
    low, high, shape = np.array([-1.0, 0.0], np.float32), np.array([1.0, 1.0], np.float32), (2,)
vae = ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=tempfile.mkdtemp(), precision=os.environ.get("MI355_PRECISION", "bf16"), training=False)
vae.init_session(init_logging=False)
agent = PPO(np.array([67]), Box(), model_dir=tempfile.mkdtemp())
agent.init_session(init_logging=False)
rng = np.random.RandomState(0)
frames = rng.randint(0, 256, (64, 80, 160, 3)).astype(np.float32) / 255.0
meas = rng.rand(64, 3).astype(np.float32)
def one(i):
    z = vae.encode([frames[i % 64]])[0]
    return agent.predict(np.concatenate([z, meas[i % 64]]), greedy=True)
for i in range(20): one(i)
torch.cuda.synchronize()
ts = []
for i in range(200):
    t0 = time.perf_counter(); one(i); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
t_enc = []
for i in range(200):
    t0 = time.perf_counter(); vae.encode([frames[i % 64]]); t_enc.append(time.perf_counter() - t0)
print("encode + predict at batch 1: median %.0f us, p90 %.0f us (encode alone: median %.0f us)" % (np.median(ts), np.percentile(ts, 90), np.median(np.array(t_enc) * 1e6)))
# the one-call form: raw uint8 frame + measurements -> action, value, state
from rollout import RolloutStep
step = RolloutStep(vae, agent)
u8 = rng.randint(0, 256, (64, 80, 160, 3), dtype=np.uint8)
for i in range(20): step(u8[i % 64], meas[i % 64])
ts = []
for i in range(500):
    t0 = time.perf_counter(); step(u8[i % 64], meas[i % 64]); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
print("RolloutStep (one call, uint8 frame in, action / value / state out; io=%s): median %.1f us, p90 %.1f us, min %.1f us" % (step.io, np.median(ts), np.percentile(ts, 90), ts.min()))
