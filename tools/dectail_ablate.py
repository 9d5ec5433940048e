#!/usr/bin/env python3
"""Where the decoder tail (csrc/dectail_tile.hpp: deconv4 forward + reconstruction loss + input gradient + filter gradient, one persistent kernel) spends its time: the op alone at
batch 512 on camera-byte labels, the product kernel and the ablations of its timing instantiation (mi_set_tuning key 25: parts switched off -- results are wrong, durations are
what is asked; round 6).  Interleaved rounds behind conditioning launches, medians.
    python tools/dectail_ablate.py [--iters 40] [--rounds 5]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from mi355 import lib as milib

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=40); ap.add_argument("--rounds", type=int, default=5); ap.add_argument("--batch", type=int, default=512)
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
dx = torch.empty_like(x); dw = torch.zeros(k, k, Co, Ci, device="cuda")
n = ctypes.c_int(0)
scratch = torch.empty(L.mi_deconv2d_tail_blocks() * 6144, device="cuda", dtype=torch.uint8)


def run(m):
    for _ in range(m):
        L.mi_deconv2d_tail_fused(st, 1, x.data_ptr(), B, IH, IW, Ci, wb.data_ptr(), wt.data_ptr(), bias.data_ptr(), k, k, Co, frames.data_ptr(), 1, idx.data_ptr(), OH * OW * Co, 0, 1.0 / B,
                                 dx.data_ptr(), dw.data_ptr(), lp.data_ptr(), bp.data_ptr(), cap, ctypes.addressof(n), scratch.data_ptr(), scratch.numel(), 0)


MODES = [(0, "product kernel"), (128, "timing instantiation, nothing off"), (1, "no transcendentals in the loss"), (64, "no fifth slot group (wave 0 does one group like the others)"),
         (32, "no slot groups at all (no forward MFMAs, no loss)"), (2, "no input-gradient phase"), (8, "input gradient computed, not stored"), (4, "no filter-gradient phase"),
         (2 | 4, "phase 1 only (forward + loss)"), (32 | 2 | 4, "staging + barriers only"), (16, "no loads (first tile's data stay)"), (16 | 8, "no loads, no gradient stores"),
         (1 | 64, "no transcendentals, no fifth group")]
run(200); torch.cuda.synchronize()
samples = {m: [] for m, _ in MODES}
for _ in range(args.rounds):
    for mask, _n in MODES:
        L.mi_set_tuning(25, mask)
        run(3); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(args.iters); e1.record(); torch.cuda.synchronize()
        samples[mask].append(e0.elapsed_time(e1) / args.iters * 1e3)
L.mi_set_tuning(25, 0)
print("decoder tail, batch %d: 221.6 MB algorithmic = 27.7 us at 8 TB/s; 14.5 GFLOP; %d blocks" % (B, n.value))
for mask, name in MODES:
    v = sorted(samples[mask])
    print("    %-66s mask %3d  median %6.1f us  (min %.1f, max %.1f)" % (name, mask, v[len(v) // 2], v[0], v[-1]))
