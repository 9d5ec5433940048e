#!/usr/bin/env python3
"""Time the decoder tail at batch 512 (bf16) alone on the GPU with HIP events: the fused launch (mi_deconv2d_tail_fused) against the three ops it
replaces (deconv4 forward + loss, deconv4 input gradient, deconv4 filter gradient).  usage: python tools/dectail_bench.py [--iters 50]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch  # noqa: E402
from mi355 import lib as milib  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=50); ap.add_argument("--batch", type=int, default=512)
args = ap.parse_args()
L = milib.get(); B, IH, IW, Ci, Co, k = args.batch, 39, 79, 32, 3, 4
OH, OW = 80, 160
st = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16
x = torch.randn(B, IH, IW, Ci, device="cuda").relu().to(bf).contiguous()
w = (torch.randn(k, k, Co, Ci, device="cuda") / (4 * Ci) ** 0.5)
wb = w.to(bf).contiguous(); wt = w.permute(3, 0, 1, 2).reshape(Ci, -1).to(bf).contiguous()
bias = torch.zeros(Co, device="cuda")
frames = torch.randint(0, 256, (2048, OH * OW * Co), device="cuda", dtype=torch.int32).to(torch.uint8).contiguous()
idx = torch.randperm(2048, device="cuda")[:B].to(torch.int32)
cap = 1 << 16
lp, bp = torch.zeros(cap, device="cuda"), torch.zeros(cap, 4, device="cuda")
dl = torch.empty(B, OH, OW, Co, device="cuda", dtype=bf); dx = torch.empty_like(x); dw = torch.zeros(k, k, Co, Ci, device="cuda")
bits = torch.zeros(B * IH * IW * 2, device="cuda", dtype=torch.int32)
n = ctypes.c_int(0)
scratch = torch.empty(L.mi_deconv2d_tail_blocks() * 6144, device="cuda", dtype=torch.uint8)

def fwd(): L.mi_deconv2d_nhwc_fwd_bce_u8(st, 1, x.data_ptr(), B, IH, IW, Ci, wb.data_ptr(), bias.data_ptr(), k, k, Co, None, frames.data_ptr(), 1, idx.data_ptr(), OH * OW * Co, 0, 1.0 / B, dl.data_ptr(), lp.data_ptr(), bp.data_ptr(), cap, ctypes.addressof(n))
def dgrad(): L.mi_deconv2d_nhwc_dgrad(st, 1, dl.data_ptr(), B, OH, OW, Co, wt.data_ptr(), 1, k, k, Ci, x.data_ptr(), dx.data_ptr())
def wgrad(): L.mi_deconv2d_nhwc_wgrad(st, 1, dl.data_ptr(), B, OH, OW, Co, x.data_ptr(), k, k, Ci, dw.data_ptr())
def fused(): L.mi_deconv2d_tail_fused(st, 1, x.data_ptr(), B, IH, IW, Ci, wb.data_ptr(), wt.data_ptr(), bias.data_ptr(), k, k, Co, frames.data_ptr(), 1, idx.data_ptr(), OH * OW * Co, 0, 1.0 / B, dx.data_ptr(), dw.data_ptr(), lp.data_ptr(), bp.data_ptr(), cap, ctypes.addressof(n), scratch.data_ptr(), scratch.numel(), 0)
def red(): L.mi_deconv2d_tail_reduce(st, scratch.data_ptr(), n.value, dw.data_ptr())
fused()
for name, fn in (("deconv4.fwd+loss", fwd), ("deconv4.dgrad", dgrad), ("deconv4.wgrad", wgrad), ("fused tail", fused), ("  its slab reduce", red)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-18s %8.1f us   (blocks %d)" % (name, e0.elapsed_time(e1) / args.iters * 1e3, n.value), flush=True)
