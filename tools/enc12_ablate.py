#!/usr/bin/env python3
"""Where the fused encoder head of the forward pass (csrc/enc12_tile.hpp) spends its time: the op alone at batch 512 on camera bytes, the product kernel and the
ablations of its timing instantiation (mi_set_tuning key 23: parts of the kernel switched off -- results are wrong, durations are what is asked).
    python tools/enc12_ablate.py [--iters 50]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from mi355 import lib as milib

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=50); ap.add_argument("--batch", type=int, default=512); ap.add_argument("--only-product", action="store_true"); ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
L = milib.get()
B = args.batch
g = torch.Generator(device="cuda"); g.manual_seed(3)
frames = torch.randint(0, 256, (B + 64, 80, 160, 3), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8).contiguous()
idx = torch.randperm(B + 64, device="cuda", generator=g)[:B].to(torch.int32).contiguous()
w1 = (torch.randn(32, 48, device="cuda", generator=g) / 7).to(torch.bfloat16).contiguous()
w2 = (torch.randn(64, 512, device="cuda", generator=g) / 22).to(torch.bfloat16).contiguous()
b1 = torch.zeros(32, device="cuda"); b2 = torch.zeros(64, device="cuda")
act1 = torch.empty(B, 39, 79, 32, device="cuda", dtype=torch.bfloat16)
bits = torch.empty(B * 39 * 79 * 2, device="cuda", dtype=torch.int32)
act2 = torch.empty(B, 18, 38, 64, device="cuda", dtype=torch.bfloat16)
launched = ctypes.c_int(0)
st = torch.cuda.current_stream().cuda_stream


def run(n):
    for _ in range(n):
        L.mi_conv2d_enc12_fwd(st, milib.MI_BF16, frames.data_ptr(), 2, idx.data_ptr(), B, 80, 160, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                              act1.data_ptr(), bits.data_ptr(), act2.data_ptr(), ctypes.addressof(launched))


NAMES = [(0, "product kernel"), (1 << 10, "timing instantiation, nothing off"), (1, "no act1 / bit-word stores"), (2, "no bit words"), (8, "no act2 stores"), (1 | 8, "no global stores at all"),
         (4, "no conv2 stage"), (16, "no frame loads"), (32, "no LDS writes of conv1"), (64, "no LDS reads in conv2"), (64 | 8, "no LDS reads in conv2, no act2 stores"), (4 | 1, "conv1 compute + LDS only (no conv2, no act1 stores)"),
         (16 | 1 | 32, "conv1 arithmetic only + conv2"), (1 | 4 | 16 | 32, "conv1 arithmetic only"),
         (2048, "ring + pipelined conv2, nothing off"), (2048 | 128, "ring + pipelined conv2, conflict-free read addresses"), (2048 | 4, "ring, no conv2 stage"), (2048 | 1 | 8, "ring + pipelined conv2, no global stores")]
PRODUCT = [(4096, "product: first form (compiler-scheduled loads and LDS reads)"), (4096 | 8192, "product: ring form of the frame loads"),
           (4096 | 8192 | 16384, "product: ring form + pipelined conv2 reads")]
run(200); torch.cuda.synchronize()                      # (clock conditioning: the kernel alone does not hold the chip at its loaded clocks from a cold start)
todo = PRODUCT if args.only_product else PRODUCT + NAMES[1:]
samples = {m: [] for m, _ in todo}
for rnd in range(args.rounds):                          # interleaved rounds, medians: every variant sees the same clock / thermal history
    for mask, name in todo:
        L.mi_set_tuning(23, mask)
        run(5); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(args.iters); e1.record(); torch.cuda.synchronize()
        assert launched.value == 1
        samples[mask].append(e0.elapsed_time(e1) / args.iters * 1e3)
for mask, name in todo:
    v = sorted(samples[mask])
    print("%-72s mask %5d  median %.1f us  (min %.1f, max %.1f)" % (name, mask, v[len(v) // 2], v[0], v[-1]))
L.mi_set_tuning(23, 0)
