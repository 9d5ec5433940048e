"""Debug: per-phase cycle breakdown of the tapconv kernel on one layer (s_memtime stamps of lane 0 of every wave).
usage: python tools/trace_tapconv.py <layer>   layer in conv2.fwd conv3.fwd deconv3.fwd conv2.dgrad deconv2.dgrad deconv3.dgrad
       python tools/trace_tapconv.py variants  per-layer timing of the tile / epilogue variants"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from mi355 import lib as milib
L = milib.get()
if os.environ.get("TRACE_DBG"):
    L.mi_set_tuning(2, int(os.environ["TRACE_DBG"]))     # 3: direct epilogue without its stores
B = int(os.environ.get("TRACE_B", "512"))
bf = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
LAYERS = {  # kind, IH, IW, Cin, Cout, k
    "conv2.fwd": ("conv", 39, 79, 32, 64, 4), "conv3.fwd": ("conv", 18, 38, 64, 128, 4),
    "deconv3.fwd": ("deconv", 18, 38, 64, 32, 5), "deconv2.fwd": ("deconv", 8, 18, 128, 64, 4),
    "conv2.dgrad": ("cdgrad", 39, 79, 32, 64, 4), "deconv3.dgrad": ("ddgrad", 18, 38, 64, 32, 5), "deconv2.dgrad": ("ddgrad", 8, 18, 128, 64, 4)}

def run(name, trace):
    kind, IH, IW, Ci, Co, k = LAYERS[name]
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    if kind == "conv":
        OH, OW = (IH - k) // 2 + 1, (IW - k) // 2 + 1
        x = torch.randn(B, IH, IW, Ci, device="cuda", generator=g).to(bf); w = torch.randn(Co, k * k * Ci, device="cuda", generator=g).to(bf) * 0.05
        b = torch.zeros(Co, device="cuda"); out = torch.empty(B, OH, OW, Co, device="cuda", dtype=bf)
        f = lambda: L.mi_conv2d_nhwc_fwd(st, 1, x.data_ptr(), None, 0, B, IH, IW, Ci, w.data_ptr(), 1, b.data_ptr(), k, k, Co, 1, out.data_ptr())
    elif kind == "deconv":
        OH, OW = (IH - 1) * 2 + k, (IW - 1) * 2 + k
        x = torch.randn(B, IH, IW, Ci, device="cuda", generator=g).to(bf); w = torch.randn(k, k, Co, Ci, device="cuda", generator=g).to(bf) * 0.05
        b = torch.zeros(Co, device="cuda"); out = torch.empty(B, OH, OW, Co, device="cuda", dtype=bf)
        f = lambda: L.mi_deconv2d_nhwc_fwd(st, 1, x.data_ptr(), B, IH, IW, Ci, w.data_ptr(), b.data_ptr(), k, k, Co, 1, out.data_ptr())
    elif kind == "cdgrad":                                 # conv dgrad: dy [B,OH,OW,Co] -> dx [B,IH,IW,Ci]
        OH, OW = (IH - k) // 2 + 1, (IW - k) // 2 + 1
        dy = torch.randn(B, OH, OW, Co, device="cuda", generator=g).to(bf); w = torch.randn(k, k, Ci, Co, device="cuda", generator=g).to(bf) * 0.05
        mask = torch.randn(B, IH, IW, Ci, device="cuda", generator=g).to(bf); dx = torch.empty(B, IH, IW, Ci, device="cuda", dtype=bf)
        f = lambda: L.mi_conv2d_nhwc_dgrad(st, 1, dy.data_ptr(), B, OH, OW, Co, w.data_ptr(), k, k, Ci, IH, IW, mask.data_ptr(), dx.data_ptr())
    else:                                                  # deconv dgrad: dy [B,OH,OW,Co] -> dx [B,IH,IW,Ci], weights [Ci][k*k*Co]
        OH, OW = (IH - 1) * 2 + k, (IW - 1) * 2 + k
        dy = torch.randn(B, OH, OW, Co, device="cuda", generator=g).to(bf); w = torch.randn(Ci, k * k * Co, device="cuda", generator=g).to(bf) * 0.05
        mask = torch.randn(B, IH, IW, Ci, device="cuda", generator=g).to(bf); dx = torch.empty(B, IH, IW, Ci, device="cuda", dtype=bf)
        f = lambda: L.mi_deconv2d_nhwc_dgrad(st, 1, dy.data_ptr(), B, OH, OW, Co, w.data_ptr(), 1, k, k, Ci, mask.data_ptr(), dx.data_ptr())
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    if not trace:
        return us, None
    cap = 4096 * 8 * 32
    buf = torch.zeros(cap, dtype=torch.int64, device="cuda")
    L.mi_debug_set_trace(buf.data_ptr(), cap)
    f(); torch.cuda.synchronize()
    L.mi_debug_set_trace(None, 0)
    t = buf.cpu().numpy().reshape(-1, 8, 32)
    return us, t

LAYERS.update({"conv3.dgrad": ("cdgrad", 18, 38, 64, 128, 4), "conv4.dgrad": ("cdgrad", 8, 18, 128, 256, 4), "deconv1.fwd": ("deconv", 3, 8, 256, 128, 4),
               "conv4.fwd": ("conv", 8, 18, 128, 256, 4), "deconv1.dgrad": ("ddgrad", 3, 8, 256, 128, 4)})
name = sys.argv[1] if len(sys.argv) > 1 else "conv2.fwd"
if name == "variants":                                     # timing only: big tile vs small tile vs gemm2 for every layer
    for nm in LAYERS:
        res = []
        for label, cfg in (("direct-epi", {5: 1, 1: 1, 6: 1}), ("staged-epi", {5: 1, 1: 1, 6: 0}), ("small+direct", {5: 2, 1: 1, 6: 1}), ("gemm2", {1: -1})):
            prev = {k: L.mi_set_tuning(k, v) for k, v in cfg.items()}
            us, _ = run(nm, False)
            for k, v in prev.items():
                L.mi_set_tuning(k, v)
            res.append("%s %.1f" % (label, us))
        print("%-14s %s" % (nm, "   ".join(res)))
    sys.exit(0)
if len(sys.argv) > 2:
    L.mi_set_tuning(5, int(sys.argv[2])); L.mi_set_tuning(1, 1)
for nm in ([name] if name != "all" else list(LAYERS)):
    us, t = run(nm, True)
    used = t[:, 0, 0] != 0
    t = t[used]
    nb = t.shape[0]
    nst = int((t[0, 0] != 0).sum())
    d = np.diff(t[:, :, :nst].astype(np.float64), axis=2)          # [block, wave, phase]
    # s_memtime ticks are 100 MHz-ish constant clock? report raw ticks and ratio to total
    tot = (t[:, :, nst - 1] - t[:, :, 0]).astype(np.float64)
    print("%s: %.1f us/launch, %d blocks traced, %d stamps; block lifetime (ticks) mean %.0f  min %.0f max %.0f" % (nm, us, nb, nst, tot.mean(), tot.min(), tot.max()))
    names = (["setup->issued", "first tiles land"] + ["step%d" % i for i in range(nst - 6)] + ["last step slice0 + later slices", "epi.stage", "epi.store"])
    m = d.mean(axis=(0, 1))
    print("   " + "  ".join("%s=%.0f" % (n, v) for n, v in zip(names, m)))
