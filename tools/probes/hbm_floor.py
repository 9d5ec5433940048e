"""Calibration: what this box's HBM does for the byte patterns of the narrow layers (torch fill / copy kernels as simple streaming references)."""
import torch
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (101, 160, 404):
    n = mb * 1000 * 1000 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda"); y = torch.empty_like(x)
    us = t(lambda: x.zero_()); print("write %d MB: %.1f us = %.2f TB/s" % (mb, us, mb / us))
    us = t(lambda: y.copy_(x)); print("copy  %d MB (read + write %d MB): %.1f us = %.2f TB/s" % (mb, 2 * mb, us, 2 * mb / us))
    us = t(lambda: x.sum()); print("read  %d MB (reduction): %.1f us = %.2f TB/s" % (mb, us, mb / us))
# cold: 1.2 GB of unrelated traffic before every timed launch (single launch between two events)
junk = torch.empty(600 * 500000, dtype=torch.bfloat16, device="cuda"); junk2 = torch.empty_like(junk)
def cold(f, n=20):
    tot = 0.0
    for _ in range(n):
        junk.fill_(1.0); junk2.copy_(junk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3
for mb in (20, 101, 160):
    n = mb * 1000 * 1000 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda"); y = torch.empty_like(x)
    print("cold write %d MB: %.1f us | cold copy (read + write %d MB): %.1f us | empty launch pair: %.1f us" % (mb, cold(lambda: x.zero_()), 2 * mb, cold(lambda: y.copy_(x)), cold(lambda: None)))
