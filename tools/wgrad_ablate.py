#!/usr/bin/env python3
"""Where the raw-staged filter-gradient kernels (csrc/tapwgrad_tile.hpp) spend their time: every one of the six mid-layer filter gradients of the ConvVAE step ALONE at batch 512
(bf16 operands, bf16 slabs as inside the engine's backward pass, the slab reduce NOT launched: mi_tapwgrad_defer drops it), the product kernel and its timing instantiations
(mi_set_tuning key 2: parts switched off -- results are wrong, durations are what is asked; round 6):
    2  every load inside ONE megabyte (L2 hits: the kernel without its HBM traffic)      3  no loads at all (stale LDS: fragment reads + MFMAs + stores only)
    4  no fragment reads / MFMAs (loads, barriers, stores only)                           5  no slab stores
and key 24 (the step's DMA rows decoded once per wave, one row per lane).  Interleaved rounds behind conditioning launches, medians.
    python tools/wgrad_ablate.py [--iters 30] [--rounds 5]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from mi355 import lib as milib

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=30); ap.add_argument("--rounds", type=int, default=5); ap.add_argument("--batch", type=int, default=512)
args = ap.parse_args()
L = milib.get()
B = args.batch
st = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16
LAYERS = [("conv2", "conv", 39, 79, 32, 64, 4), ("conv3", "conv", 18, 38, 64, 128, 4), ("conv4", "conv", 8, 18, 128, 256, 4),
          ("deconv1", "deconv", 3, 8, 256, 128, 4), ("deconv2", "deconv", 8, 18, 128, 64, 4), ("deconv3", "deconv", 18, 38, 64, 32, 5)]
ws = torch.empty(512 << 20, device="cuda", dtype=torch.uint8)
ops = {}
for name, form, ih, iw, ci, co, k in LAYERS:
    oh, ow = ((ih - k) // 2 + 1, (iw - k) // 2 + 1) if form == "conv" else ((ih - 1) * 2 + k, (iw - 1) * 2 + k)
    x = torch.randn(B, ih, iw, ci, device="cuda").relu().to(bf).contiguous()
    dy = torch.randn(B, oh, ow, co, device="cuda").to(bf).contiguous()
    dw = torch.zeros(k * k * ci * co, device="cuda"); db = torch.zeros(co, device="cuda")
    flops = 2.0 * k * k * ci * co * B * (oh * ow if form == "conv" else ih * iw)
    nbytes = (x.numel() + dy.numel()) * 2
    if form == "conv":
        call = lambda x=x, dy=dy, dw=dw, db=db, g=(ih, iw, ci, k, co): L.mi_conv2d_nhwc_wgrad_ws(st, 1, x.data_ptr(), None, 0, B, g[0], g[1], g[2], dy.data_ptr(), g[3], g[3], g[4], dw.data_ptr(), ws.data_ptr(), ws.numel(), db.data_ptr())   # noqa: E731
    else:
        call = lambda x=x, dy=dy, dw=dw, db=db, g=(oh, ow, co, k, ci): L.mi_deconv2d_nhwc_wgrad_ws(st, 1, dy.data_ptr(), B, g[0], g[1], g[2], x.data_ptr(), g[3], g[3], g[4], dw.data_ptr(), ws.data_ptr(), ws.numel(), db.data_ptr())   # noqa: E731
    ops[name] = (call, flops, nbytes)

L.mi_set_tuning(18, 1)                                    # bf16 slabs, as the engine's backward pass sets them


def run(call, n):
    for _ in range(n):
        L.cdll.mi_tapwgrad_defer(1)                        # (drops the recorded slab reduce of the previous call: the kernel alone)
        call()


MODES = [((0, 0), "product kernel"), ((0, 1), "rows decoded once per wave (key 24)"), ((2, 0), "loads inside 1 MB (no HBM traffic)"), ((3, 0), "no loads"),
         ((4, 0), "no fragment reads / MFMAs"), ((5, 0), "no slab stores")]
for name, (call, flops, nbytes) in ops.items():
    run(call, 100); torch.cuda.synchronize()
    samples = {m: [] for m, _ in MODES}
    for _ in range(args.rounds):
        for (dbg, ldec), _n in MODES:
            L.mi_set_tuning(2, dbg); L.mi_set_tuning(24, 3 if ldec else 0)
            run(call, 3); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(call, args.iters); e1.record(); torch.cuda.synchronize()
            samples[(dbg, ldec)].append(e0.elapsed_time(e1) / args.iters * 1e3)
    L.mi_set_tuning(2, 0); L.mi_set_tuning(24, 0)
    print("%s.wgrad  (%.1f GFLOP = %.1f us at 2.5 PF; %.1f MB of operands = %.1f us at 8 TB/s)" % (name, flops / 1e9, flops / 2.5e15 * 1e6, nbytes / 1e6, nbytes / 8e12 * 1e6))
    for (m, nm) in MODES:
        v = sorted(samples[m])
        print("    %-44s median %6.1f us  (min %.1f, max %.1f)" % (nm, v[len(v) // 2], v[0], v[-1]))
L.cdll.mi_tapwgrad_defer(0)
