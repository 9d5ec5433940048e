"""Debug: oracle (CPU baseline) throughput against the torch thread count -- how bench.py's cpu_baseline thread default was chosen.
Lives under tests/ because it runs the oracle (test infrastructure).  python tests/dbg_cpu_threads_probe.py"""
import os
import sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vae_oracle as vo
fr = np.random.RandomState(0).rand(256,80,160,3).astype(np.float32); eps=np.zeros((256,64),np.float32)
for th in (16, 32, 64, 128):
    torch.set_num_threads(th)
    o = vo.OracleVAE(seed=0); o.train_step(fr, fr, eps)
    t=time.time(); o.train_step(fr, fr, eps); dt=time.time()-t
    print("threads", th, "frames/s", 256/dt, flush=True)
