#!/usr/bin/env python3
"""The five narrow-layer ops of the ConvVAE step (conv1 fwd / wgrad, deconv4 fwd+loss / dgrad / wgrad) alone on the GPU at batch 512, bf16,
uint8 frames, with HIP events, next to their algorithmic byte counts (what one launch must move through HBM).
    python tools/narrow_bench.py [op ...] [--iters 50]        ops: conv1.fwd deconv4.dgrad deconv4.fwd conv1.wgrad deconv4.wgrad"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from mi355 import lib as milib

ap = argparse.ArgumentParser()
ap.add_argument("ops", nargs="*", default=["conv1.fwd", "deconv4.dgrad", "deconv4.fwd", "conv1.wgrad", "deconv4.wgrad"])
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--thrash", type=int, default=0, help="MB of unrelated traffic (a fill + a copy) between launches: the op then meets cold caches, as inside the step")
args = ap.parse_args()
L = milib.get()
B = args.batch
st = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16
g = torch.Generator(device="cuda"); g.manual_seed(0)
frames = torch.randint(0, 256, (2048, 80, 160, 3), dtype=torch.uint8, device="cuda", generator=g)
idx = torch.randperm(2048, device="cuda", generator=g)[:B].to(torch.int32)
w1 = (torch.randn(4, 4, 3, 32, device="cuda", generator=g) / 7).contiguous()
w1t = w1.permute(3, 0, 1, 2).reshape(32, 48).to(bf).contiguous()
b1 = torch.zeros(32, device="cuda")
act1 = torch.empty(B, 39, 79, 32, dtype=bf, device="cuda")
bits1 = torch.empty(B * 39 * 79 * 2, dtype=torch.int32, device="cuda")
wrote = ctypes.c_int(0)
dec3 = torch.randn(B, 39, 79, 32, device="cuda", generator=g).relu().to(bf)
w4 = (torch.randn(4, 4, 3, 32, device="cuda", generator=g) / 11).contiguous()             # deconv4 kernel [kh,kw,co=3,ci=32]
w4b = w4.to(bf).contiguous()
w4t = w4.permute(3, 0, 1, 2).reshape(32, 48).to(bf).contiguous()                       # K-contiguous copy for the input gradient
b4 = torch.zeros(3, device="cuda")
dlog = (torch.randn(B, 80, 160, 3, device="cuda", generator=g) / 512).to(bf)
gdec3 = torch.empty(B, 39, 79, 32, dtype=bf, device="cuda")
lpart = torch.zeros(1 << 16, device="cuda"); bpart = torch.zeros(4 << 16, device="cuda")
npart = ctypes.c_int(0)
dw = torch.zeros(48 * 32, device="cuda"); db = torch.zeros(32, device="cuda")
scr = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
MB = 1e6
px = B * 39 * 79
CALLS = {
    "conv1.fwd": (lambda: L.mi_conv2d_nhwc_fwd_bits(st, 1, frames.data_ptr(), idx.data_ptr(), 2, B, 80, 160, 3, w1t.data_ptr(), 1, b1.data_ptr(), 4, 4, 32, 1, act1.data_ptr(), bits1.data_ptr(), ctypes.byref(wrote)),
                  B * 38400 + px * 64 + px * 8),
    "deconv4.dgrad": (lambda: L.mi_deconv2d_nhwc_dgrad_bits(st, 1, dlog.data_ptr(), B, 80, 160, 3, w4t.data_ptr(), 1, 4, 4, 32, None, bits1.data_ptr(), gdec3.data_ptr()),
                      B * 38400 * 2 + px * 8 + px * 64),
    "deconv4.fwd": (lambda: L.mi_deconv2d_nhwc_fwd_bce_u8(st, 1, dec3.data_ptr(), B, 39, 79, 32, w4b.data_ptr(), b4.data_ptr(), 4, 4, 3, None, frames.data_ptr(), 1, idx.data_ptr(), 38400, 0, 1.0 / B,
                                                           dlog.data_ptr(), lpart.data_ptr(), bpart.data_ptr(), 1 << 16, ctypes.byref(npart)),
                    px * 64 + B * 38400 + B * 38400 * 2),
    "conv1.wgrad": (lambda: L.mi_conv2d_nhwc_wgrad_ws(st, 1, frames.data_ptr(), idx.data_ptr(), 2, B, 80, 160, 3, gdec3.data_ptr(), 4, 4, 32, dw.data_ptr(), scr.data_ptr(), scr.numel(), db.data_ptr()),
                    B * 38400 + px * 64),
    "deconv4.wgrad": (lambda: L.mi_deconv2d_nhwc_wgrad_ws(st, 1, dlog.data_ptr(), B, 80, 160, 3, dec3.data_ptr(), 4, 4, 32, dw.data_ptr(), scr.data_ptr(), scr.numel(), None),
                      B * 38400 * 2 + px * 64),
}
for op in args.ops:
    f, nbytes = CALLS[op]
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    if args.thrash:
        junk = torch.empty(args.thrash * 500000, dtype=torch.bfloat16, device="cuda"); junk2 = torch.empty_like(junk)
        tot = 0.0
        for _ in range(args.iters):
            junk.fill_(1.0); junk2.copy_(junk)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        us = tot / args.iters * 1e3
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / args.iters * 1e3
    print("%-14s %6.1f us   %6.1f MB algorithmic   %.2f TB/s   (%.0f %% of 8 TB/s)" % (op, us, nbytes / MB, nbytes / us / 1e6, nbytes / us / 1e6 / 8 * 100))
