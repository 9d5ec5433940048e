#!/usr/bin/env python3
"""Time single layer ops of the ConvVAE step at batch 512 (bf16) through the C ABI, alone on the GPU, with HIP events:
    python tools/op_bench.py [op ...] [--iters 50] [--set key=value ...]
ops: deconv3.fwd conv2.dgrad conv2.fwd deconv3.dgrad deconv2.fwd conv3.dgrad conv3.fwd deconv2.dgrad deconv1.fwd conv4.dgrad
Used for kernel A/B work (tuning keys via --set, e.g. --set 13=0) and as the workload of per-kernel rocprofv3 --pmc passes."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402
from mi355 import lib as milib  # noqa: E402

ENC = {"conv2": (39, 79, 32, 64, 4), "conv3": (18, 38, 64, 128, 4), "conv4": (8, 18, 128, 256, 4)}
DEC = {"deconv1": (3, 8, 256, 128, 4), "deconv2": (8, 18, 128, 64, 4), "deconv3": (18, 38, 64, 32, 5)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("ops", nargs="*", default=["deconv3.fwd", "conv2.dgrad"])
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--set", action="append", default=[])
    ap.add_argument("--nomask", action="store_true", help="input gradients without their ReluGrad mask (upper bound of what bit words can save)")
    args = ap.parse_args()
    L = milib.get()
    for kv in args.set:
        k, v = kv.split("=")
        L.mi_set_tuning(int(k), int(v))
    B = args.batch
    st = torch.cuda.current_stream().cuda_stream
    bf = torch.bfloat16

    def rnd(*shape, relu=False):
        t = torch.randn(*shape, device="cuda")
        return (t.relu() if relu else t).to(bf).contiguous()
    for op in args.ops:
        layer, kind = op.split(".")
        if layer in ENC:
            ih, iw, ci, co, k = ENC[layer]
            oh, ow = (ih - k) // 2 + 1, (iw - k) // 2 + 1
            x, y = rnd(B, ih, iw, ci, relu=True), rnd(B, oh, ow, co)
            w = (torch.randn(k, k, ci, co, device="cuda") / (k * k * ci) ** 0.5)
            wt = w.permute(3, 0, 1, 2).reshape(co, -1).to(bf).contiguous()          # K-contiguous [Cout][kh*kw*ci]
            wb = w.to(bf).contiguous()
            bias = torch.zeros(co, device="cuda")
            out = torch.empty_like(y) if kind == "fwd" else torch.empty_like(x)
            if kind == "fwd":
                call = lambda: L.mi_conv2d_nhwc_fwd(st, 1, x.data_ptr(), None, 0, B, ih, iw, ci, wt.data_ptr(), 1, bias.data_ptr(), k, k, co, 1, out.data_ptr())   # noqa: E731
            else:
                call = lambda: L.mi_conv2d_nhwc_dgrad(st, 1, y.data_ptr(), B, oh, ow, co, wb.data_ptr(), k, k, ci, ih, iw, None if args.nomask else x.data_ptr(), out.data_ptr())   # noqa: E731
            flops = 2.0 * oh * ow * co * k * k * ci * B
        else:
            ih, iw, ci, co, k = DEC[layer]
            oh, ow = (ih - 1) * 2 + k, (iw - 1) * 2 + k
            x, y = rnd(B, ih, iw, ci, relu=True), rnd(B, oh, ow, co)
            w = (torch.randn(k, k, co, ci, device="cuda") / (k * k * ci) ** 0.5)
            wb = w.to(bf).contiguous()
            wt = w.permute(3, 0, 1, 2).reshape(ci, -1).to(bf).contiguous()          # [Cin][kh*kw*co]
            bias = torch.zeros(co, device="cuda")
            out = torch.empty_like(y) if kind == "fwd" else torch.empty_like(x)
            if kind == "fwd":
                call = lambda: L.mi_deconv2d_nhwc_fwd(st, 1, x.data_ptr(), B, ih, iw, ci, wb.data_ptr(), bias.data_ptr(), k, k, co, 1, out.data_ptr())   # noqa: E731
            else:
                call = lambda: L.mi_deconv2d_nhwc_dgrad(st, 1, y.data_ptr(), B, oh, ow, co, wt.data_ptr(), 1, k, k, ci, x.data_ptr() if layer != "deconv1" and not args.nomask else None, out.data_ptr())   # noqa: E731
            flops = 2.0 * ih * iw * ci * k * k * co * B
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / args.iters * 1e3
        nbytes = (x.numel() + y.numel() + (x.numel() if kind == "dgrad" and layer in ENC else 0)) * 2
        print("%-14s %8.1f us   %7.1f TFLOP/s   %6.2f TB/s (in + out%s)" % (op, us, flops / us / 1e6, nbytes / us / 1e6, " + mask" if kind == "dgrad" and layer in ENC else ""), flush=True)


if __name__ == "__main__":
    main()
