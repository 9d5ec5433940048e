"""Debug: per-phase cycle breakdown of the tapwgrad kernel (s_memtime stamps).  python tools/trace_tapwgrad.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from mi355 import lib as milib
L = milib.get()
if len(sys.argv) > 2:
    L.mi_set_tuning(int(sys.argv[1]), int(sys.argv[2]))       # e.g. "7 0": wave = (tap, output tile) layout
B = 512
bf = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
LAYERS = {"conv2.wgrad": ("conv", 39, 79, 32, 64, 4), "conv3.wgrad": ("conv", 18, 38, 64, 128, 4), "conv4.wgrad": ("conv", 8, 18, 128, 256, 4),
          "deconv3.wgrad": ("deconv", 18, 38, 64, 32, 5), "deconv2.wgrad": ("deconv", 8, 18, 128, 64, 4), "deconv1.wgrad": ("deconv", 3, 8, 256, 128, 4)}
scr = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")     # the engine's split-reduction scratch
db = torch.zeros(256, device="cuda")
for nm, (kind, IH, IW, Ci, Co, k) in LAYERS.items():
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    if kind == "conv":
        OH, OW = (IH - k) // 2 + 1, (IW - k) // 2 + 1
        x = torch.randn(B, IH, IW, Ci, device="cuda", generator=g).to(bf); dy = torch.randn(B, OH, OW, Co, device="cuda", generator=g).to(bf)
        dw = torch.zeros(k, k, Ci, Co, device="cuda")
        f = lambda: L.mi_conv2d_nhwc_wgrad_ws(st, 1, x.data_ptr(), None, 0, B, IH, IW, Ci, dy.data_ptr(), k, k, Co, dw.data_ptr(), scr.data_ptr(), scr.numel(), db.data_ptr())
    else:
        OH, OW = (IH - 1) * 2 + k, (IW - 1) * 2 + k
        x = torch.randn(B, IH, IW, Ci, device="cuda", generator=g).to(bf); dy = torch.randn(B, OH, OW, Co, device="cuda", generator=g).to(bf)
        dw = torch.zeros(k, k, Co, Ci, device="cuda")
        f = lambda: L.mi_deconv2d_nhwc_wgrad_ws(st, 1, dy.data_ptr(), B, OH, OW, Co, x.data_ptr(), k, k, Ci, dw.data_ptr(), scr.data_ptr(), scr.numel(), db.data_ptr())
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    cap = 1024 * 8 * 32
    buf = torch.zeros(cap, dtype=torch.int64, device="cuda")
    L.mi_debug_set_trace(buf.data_ptr(), cap)
    f(); torch.cuda.synchronize()
    L.mi_debug_set_trace(None, 0)
    t = buf.cpu().numpy().reshape(-1, 8, 32)
    t = t[t[:, 0, 0] != 0]
    if t.shape[0] == 0:                                  # (a kernel without trace stamps, e.g. tapwgrad_cw_kernel)
        print("%s: %.1f us/launch (no trace stamps in this kernel)" % (nm, us))
        continue
    nst = int((t[0, 0] != 0).sum())
    d = np.diff(t[:, :, :nst].astype(np.float64), axis=2).mean(axis=(0, 1))
    tot = (t[:, :, nst - 1] - t[:, :, 0]).astype(np.float64)
    names = ["setup+issue0"] + sum([["s%d.wait" % i, "s%d.issue" % i] for i in range(4)], [])[: nst - 3]
    names[2 + 0] = names[2]
    lab = ["setup+issue0"]
    for i in range((nst - 3) // 2):
        lab += ["s%d.barrier" % i if i == 0 else "s%d.compute(prev)+barrier" % i, "s%d.issue" % i]
    lab += ["remaining steps", "dW stores"]
    if os.environ.get("MI355_LIB", "").endswith("lib_epi.so"):      # python -m mi355.build --variant epi -DMI355_TW_EPI_STAMPS: the epilogue's phases
        lab = lab[:-1] + ["wait for the block", "LDS writes", "barrier", "LDS reads + adds", "slab stores"]
    print("%s: %.1f us/launch, %d blocks, lifetime mean %.0f (min %.0f max %.0f) cycles" % (nm, us, t.shape[0], tot.mean(), tot.min(), tot.max()))
    print("   " + "  ".join("%s=%.0f" % (n, v) for n, v in zip(lab, d)))
