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
elif name == "deconv3.dgrad":                      # conv-form register-weight kernel: 4 stamps per chunk
    IH, IW, Ci, Co, k = 18, 38, 64, 32, 5
    OH, OW = (IH - 1) * 2 + k, (IW - 1) * 2 + k
    dy = torch.randn(B, OH, OW, Co, device="cuda", generator=g).to(bf); wt = (torch.randn(Ci, k * k * Co, device="cuda", generator=g) * 0.05).to(bf)
    mask = torch.randn(B, IH, IW, Ci, device="cuda", generator=g).relu().to(bf); dx = torch.empty(B, IH, IW, Ci, device="cuda", dtype=bf)
    f = lambda: L.mi_deconv2d_nhwc_dgrad(st, 1, dy.data_ptr(), B, OH, OW, Co, wt.data_ptr(), 1, k, k, Ci, mask.data_ptr(), dx.data_ptr())
elif name == "conv3.fwd":                          # conv-form register-weight kernel, 64 -> 128 channels (rwconv_conv_kernel<4, 2>): 4 stamps per chunk
    IH, IW, Ci, Co, k = 18, 38, 64, 128, 4
    OH, OW = (IH - k) // 2 + 1, (IW - k) // 2 + 1
    L.mi_set_tuning(15, 3)
    x = torch.randn(B, IH, IW, Ci, device="cuda", generator=g).relu().to(bf); wt = (torch.randn(Co, k * k * Ci, device="cuda", generator=g) * 0.05).to(bf)
    b = torch.zeros(Co, device="cuda"); out = torch.empty(B, OH, OW, Co, device="cuda", dtype=bf)
    wf = torch.zeros(1 << 20, device="cuda", dtype=torch.uint8)
    use_frag = os.environ.get("FRAG", "1") == "1"      # (timing only: the fragment-ordered copy holds zeros)
    def f():
        if use_frag:
            L.mi_rwconv_next_weights_fragment_ordered(wf.data_ptr())
        L.mi_conv2d_nhwc_fwd(st, 1, x.data_ptr(), None, 0, B, IH, IW, Ci, wt.data_ptr(), 1, b.data_ptr(), k, k, Co, 1, out.data_ptr())
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
raw = buf.cpu().numpy().reshape(-1, 8, 32)
raw = raw[raw[:, 0, 0] != 0]
hw = raw[:, :, 31].copy()
t = raw.astype(np.float64); t[:, :, 31] = 0
nb = t.shape[0]
nw = int((t[0, :, 0] != 0).sum())
print("%s: %.1f us / launch alone; %d blocks traced, %d waves per block" % (name, us, nb, nw))
if name in ("deconv3.dgrad", "conv3.fwd"):
    for c in range(nw):
        w_ = t[:, c, :31]
        n = int((w_ > 0).sum(axis=1).min())
        nch = (n - 3) // 4
        d = lambda a, b_: (w_[:, a] - w_[:, b_]).mean()
        print("wave %d: stage-req %.0f weights-req %.0f | per chunk (decode+mask-req, MFMA loop, epilogue, wait+barrier): %s | lifetime %.0f" % (
            c, d(1, 0), d(2, 1), "  ".join("%.0f/%.0f/%.0f/%.0f" % (d(4 + 4 * i, 3 + 4 * i), d(5 + 4 * i, 4 + 4 * i), d(6 + 4 * i, 5 + 4 * i),
                                                                    d(7 + 4 * i, 6 + 4 * i) if 7 + 4 * i < n else 0) for i in range(nch)), (w_.max(axis=1) - w_[:, 0]).mean()))
    sys.exit(0)
for c in range(nw):
    w_ = t[:, c, :31]
    n = (w_ > 0).sum(axis=1)                       # stamps per wave: 3 + 2 chunks
    nch = ((n - 3) // 2).astype(int)
    k = max(int(nch.min()), 1)
    comp = np.array([w_[:, 4 + 2 * i] - w_[:, 3 + 2 * i] for i in range(k)])
    wait = np.array([w_[:, 5 + 2 * i] - w_[:, 4 + 2 * i] for i in range(k - 1)]) if k > 1 else np.zeros((1, 1))
    simd = (hw[:, c] >> 4) & 3
    print("wave %d: stage-req %.0f weights-req %.0f first-landed %.0f | compute per chunk %s | barrier wait %s | simd histogram %s" % (
        c, (w_[:, 1] - w_[:, 0]).mean(), (w_[:, 2] - w_[:, 1]).mean(), (w_[:, 3] - w_[:, 2]).mean(),
        " ".join("%.0f" % v for v in comp.mean(axis=1)), " ".join("%.0f" % v for v in wait.mean(axis=1)), np.bincount(simd, minlength=4).tolist()))
same = 0
for b in range(nb):
    sd = (hw[b, :nw] >> 4) & 3
    if nw == 8 and all(sd[i] == sd[i + 4] for i in range(4)) and len(set(sd[:4].tolist())) == 4:
        same += 1
print("blocks whose waves w and w+4 share a SIMD (and waves 0-3 cover all four): %d of %d" % (same, nb))
