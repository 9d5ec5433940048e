"""Debug: per-phase cycle breakdown of the register-weight kernel (rwconv.hip) on deconv3.fwd / conv2.dgrad at batch 512: s_memtime stamps of
lane 0 of every wave (mi_debug_set_trace) -> per class: wait for the staged slot range, MFMA phase and epilogue phase per tile, wave lifetime;
per CU: how many blocks were resident on average (from the stamps' HW_ID word).   usage: python tools/trace_rwconv.py [deconv3.fwd|conv2.dgrad]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from mi355 import lib as milib
L = milib.get()
B = 512
bf = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
name = sys.argv[1] if len(sys.argv) > 1 else "deconv3.fwd"
g = torch.Generator(device="cuda"); g.manual_seed(0)
if name == "deconv3.fwd":
    IH, IW, Ci, Co, k = 18, 38, 64, 32, 5
    OH, OW = (IH - 1) * 2 + k, (IW - 1) * 2 + k
    x = torch.randn(B, IH, IW, Ci, device="cuda", generator=g).to(bf); w = (torch.randn(k, k, Co, Ci, device="cuda", generator=g) * 0.05).to(bf)
    b = torch.zeros(Co, device="cuda"); out = torch.empty(B, OH, OW, Co, device="cuda", dtype=bf)
    f = lambda: L.mi_deconv2d_nhwc_fwd(st, 1, x.data_ptr(), B, IH, IW, Ci, w.data_ptr(), b.data_ptr(), k, k, Co, 1, out.data_ptr())
else:
    IH, IW, Ci, Co, k = 39, 79, 32, 64, 4
    OH, OW = (IH - k) // 2 + 1, (IW - k) // 2 + 1
    dy = torch.randn(B, OH, OW, Co, device="cuda", generator=g).to(bf); w = (torch.randn(k, k, Ci, Co, device="cuda", generator=g) * 0.05).to(bf)
    mask = torch.randn(B, IH, IW, Ci, device="cuda", generator=g).relu().to(bf); dx = torch.empty(B, IH, IW, Ci, device="cuda", dtype=bf)
    f = lambda: L.mi_conv2d_nhwc_dgrad(st, 1, dy.data_ptr(), B, OH, OW, Co, w.data_ptr(), k, k, Ci, IH, IW, mask.data_ptr(), dx.data_ptr())
for _ in range(3):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
cap = 2048 * 8 * 32
buf = torch.zeros(cap, dtype=torch.int64, device="cuda")
L.mi_debug_set_trace(buf.data_ptr(), cap)
f(); torch.cuda.synchronize()
L.mi_debug_set_trace(None, 0)
t = buf.cpu().numpy().reshape(-1, 8, 32)[:, :4, :]
t = t[t[:, 0, 0] != 0].astype(np.float64)
nb = t.shape[0]
t0 = t[:, :, 0].min()
print("%s: %.1f us / launch alone; %d blocks traced; kernel span %.0f ticks (s_memtime: 100 MHz)" % (name, us, nb, t[:, :, :20].max() - t0))
ntile = 8
for c in range(4):
    w_ = t[:, c, :]
    life = w_[:, 3 + 2 * ntile] - w_[:, 0]
    mf = np.array([w_[:, 4 + 2 * i] - (w_[:, 3] if i == 0 else w_[:, 3 + 2 * i]) for i in range(ntile)])
    ep = np.array([w_[:, 5 + 2 * i] - w_[:, 4 + 2 * i] for i in range(ntile)])
    print("class %d: dma-issue %.0f  weights-req %.0f  wait-landed %.0f | mfma/tile %s | epi/tile %s | life %.0f" % (
        c, (w_[:, 1] - w_[:, 0]).mean(), (w_[:, 2] - w_[:, 1]).mean(), (w_[:, 3] - w_[:, 2]).mean(),
        " ".join("%.0f" % v for v in mf.mean(axis=1)), " ".join("%.0f" % v for v in ep.mean(axis=1)), life.mean()))
# residency: HW_ID word (stamp index 4 + 2*ntile): bits [11:8] CU id, [15:13] SE (gfx9 layout); count blocks alive over time per (xcc, se, cu)
hw = t[:, 0, 4 + 2 * ntile].astype(np.int64)
start, end = t[:, :, 0].min(axis=1), t[:, :, 3 + 2 * ntile].max(axis=1)
span = end.max() - start.min()
print("sum of block lifetimes / (kernel span x 256 CUs) = %.2f blocks resident per CU on average" % ((end - start).sum() / (span * 256)))
print("HW_ID samples:", [hex(int(v)) for v in hw[:6]])
