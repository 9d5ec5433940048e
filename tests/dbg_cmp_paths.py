"""Debug: run one bf16 (or fp32) forward+backward of the ConvVAE at batch B on every kernel generation and report, per
workspace tensor / gradient, the max abs difference against the first-generation kernels.  (Under tests/: it uses the oracle's parameters.)  python tests/dbg_cmp_paths.py [B] [prec]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from vae.models import ConvVAE
from mi355 import lib as milib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
L = milib.get()
rng = np.random.RandomState(0)
frames = (rng.randint(0, 256, (B, 80, 160, 3)).astype(np.float32) / 255.0)
eps = rng.standard_normal((B, 64)).astype(np.float32)

sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import vae_oracle as vo
PARAMS = vo.init_vae_params(0, 64, (80, 160, 3), (80, 160, 3))
_r = np.random.RandomState(1)
for _k in PARAMS:
    if _k.endswith("bias"):
        PARAMS[_k] = (0.05 * _r.standard_normal(PARAMS[_k].shape)).astype(np.float32)

def run(gemm2, tapmin, precision=None):
    L.mi_set_tuning(0, gemm2); L.mi_set_tuning(1, tapmin)
    m = ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=tempfile.mkdtemp(), precision=precision or prec, seed=0)
    m.set_weights(PARAMS)
    m.init_session(init_logging=False)
    rs = np.random.RandomState(5)
    w = m.get_weights() if hasattr(m, "get_weights") else None
    src = m._frames(frames, 38400, "src")
    e = m._eps(B, eps)
    m.dev.forward(src, src, None, B, 1.0 / B, e, 1, 1)
    m.dev.backward(src, None, e, 1.0 / B, 0)
    torch.cuda.synchronize()
    g = m.dev.export_grads()
    ws = m.dev.workspace.clone() if hasattr(m.dev, "workspace") else None
    return {k: np.asarray(v, np.float64) for k, v in g.items()}, m.dev.losses.cpu().numpy().copy()

truth, ltruth = run(0, -1, "fp32")
ref, lref = run(0, -1)
res = {"gen1": ref}
for name, cfg in (("gemm2", (1, -1)), ("tapconv-all", (1, 1))):
    res[name], l = run(*cfg)
    print("==", name, "losses", l, "ref", lref, "fp32", ltruth)
print("max-norm error of each bf16 path against the fp32-mode gradients:")
for k in sorted(ref):
    print("   %-36s" % k, "  ".join("%s %.4f" % (n, np.abs(res[n][k] - truth[k]).max() / max(np.abs(truth[k]).max(), 1e-30)) for n in res))
