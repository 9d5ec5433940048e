"""Per-kernel timing of the four small-grid layers at batch 512, alone on the GPU: activation-resident kernels (csrc/ares_tile.hpp) vs the general tile kernels."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "carla-ppo_amd"))
from mi355 import lib as milib

L = milib.get()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
st = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16
g = torch.Generator(device="cuda"); g.manual_seed(0)
x0 = torch.randn(B, 8, 18, 128, device="cuda", generator=g).to(bf)
x1 = torch.randn(B, 3, 8, 256, device="cuda", generator=g).to(bf)
w = (torch.randn(4, 4, 128, 256, device="cuda", generator=g) / 45.0)
wb = w.to(bf).contiguous()
bias0 = torch.zeros(256, device="cuda"); bias1 = torch.zeros(128, device="cuda")
nb = int(L.mi_ares_weight_bytes())
wf0 = torch.empty(nb, device="cuda", dtype=torch.uint8); wf1 = torch.empty(nb, device="cuda", dtype=torch.uint8)
L.mi_ares_pack_weights(st, 0, w.data_ptr(), wf0.data_ptr()); L.mi_ares_pack_weights(st, 1, w.data_ptr(), wf1.data_ptr())
wt = torch.empty(16 * 128 * 256, device="cuda", dtype=bf)
offs, Ks, Ns = np.array([0], np.int64), np.array([2048], np.int32), np.array([256], np.int32)
L.mi_transpose_weights(st, 1, w.data_ptr(), wt.data_ptr(), offs.ctypes.data, Ks.ctypes.data, Ns.ctypes.data, 1)
o0 = torch.empty(B, 3, 8, 256, device="cuda", dtype=bf); o1 = torch.empty(B, 8, 18, 128, device="cuda", dtype=bf)
flag = np.zeros(1, np.int32)
thrash = torch.empty(512 << 20, device="cuda", dtype=torch.uint8)


def timed(fn, n=30, cold=False):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        if cold:
            thrash.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


x2 = torch.randn(B, 8, 18, 128, device="cuda", generator=g).to(bf)
w2 = (torch.randn(4, 4, 64, 128, device="cuda", generator=g) / 22.0)
w2b = w2.to(bf).contiguous()
bias2 = torch.zeros(64, device="cuda")
wf2 = torch.empty(nb, device="cuda", dtype=torch.uint8)
L.mi_ares_pack_weights(st, 2, w2.data_ptr(), wf2.data_ptr())
o2 = torch.empty(B, 18, 38, 64, device="cuda", dtype=bf)
os.environ.setdefault("MI355_ARES_MID", "1")
for cold in (False, True):
    e_ = timed(lambda: L.mi_ares_conv(st, 1, 2, x2.data_ptr(), B, wf2.data_ptr(), bias2.data_ptr(), 1, None, o2.data_ptr(), flag.ctypes.data), cold=cold)
    f_ = timed(lambda: L.mi_deconv2d_nhwc_fwd(st, 1, x2.data_ptr(), B, 8, 18, 128, w2b.data_ptr(), bias2.data_ptr(), 4, 4, 64, 1, o2.data_ptr()), cold=cold)
    print("B=%d %s: mid gather form  ares %.1f us (min %.1f)  general %.1f us (min %.1f)  (launched %d)" % ((B, "cold" if cold else "warm") + e_ + f_ + (int(flag[0]),)))
for cold in (False, True):
    tag = "cold (512 MB written in between)" if cold else "warm"
    a = timed(lambda: L.mi_ares_conv(st, 1, 0, x0.data_ptr(), B, wf0.data_ptr(), bias0.data_ptr(), 1, None, o0.data_ptr(), flag.ctypes.data), cold=cold)
    b = timed(lambda: L.mi_conv2d_nhwc_fwd(st, 1, x0.data_ptr(), None, 0, B, 8, 18, 128, wt.data_ptr(), 1, bias0.data_ptr(), 4, 4, 256, 1, o0.data_ptr()), cold=cold)
    c = timed(lambda: L.mi_ares_conv(st, 1, 1, x1.data_ptr(), B, wf1.data_ptr(), bias1.data_ptr(), 1, None, o1.data_ptr(), flag.ctypes.data), cold=cold)
    d = timed(lambda: L.mi_deconv2d_nhwc_fwd(st, 1, x1.data_ptr(), B, 3, 8, 256, wb.data_ptr(), bias1.data_ptr(), 4, 4, 128, 1, o1.data_ptr()), cold=cold)
    print("B=%d %s: conv form  ares %.1f us (min %.1f)  general %.1f us (min %.1f) | gather form  ares %.1f us (min %.1f)  general %.1f us (min %.1f)" % ((B, tag) + a + b + c + d))
