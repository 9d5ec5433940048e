#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its configuration, measured on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one ConvVAE SGD step (forward, ELBO, backward, gradient all-reduce when N>1, fused TF-Adam) on one minibatch
of 512 synthetic 160x80x3 frames PER GPU in bf16 (BASELINE configs[1]; global batch 512*N -> configs[3] at N=8), frames
already resident in HBM (a pool of synthetic uint8-grid frames, minibatches gathered inside the conv1 / loss kernels).
`value` = frames/s of the whole job = N*512*K / max-over-ranks(time of exactly K steps).

Extra objects on the same JSON line:
  roofline      the dominant kernel of the step (picked from a per-op HIP-event profile during warm-up), timed with HIP
                events on the launch stream over the K timed steps; achieved = algorithmic FLOPs (or bytes) / avg duration.
  cpu_baseline  the CPU oracle (a port of the reference's TF graph; the reference itself needs TensorFlow 1.13) timed on
                this box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
  ppo           PPO update throughput on z=64 latents (BASELINE configs[2]) — reported, not part of `value`.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

METRIC = "VAE+PPO train frames/sec on 160×80×3 synthetic obs at 1/2/4/8 MI355X"
PEAK = {"mfma_bf16": 2.5e15, "mfma_f32": 157.3e12, "hbm": 8.0e12}       # /opt/skills/guides/MI355X_MICROARCH.md (dense, no sparsity)

ENC = [(80, 160, 3, 32), (39, 79, 32, 64), (18, 38, 64, 128), (8, 18, 128, 256)]          # IH, IW, Cin, Cout  (k4 s2)
DEC = [(3, 8, 256, 128, 4), (8, 18, 128, 64, 4), (18, 38, 64, 32, 5), (39, 79, 32, 3, 4)]  # IH, IW, Cin, Cout, k


def op_work(name, B, esz, n_params):
    """Algorithmic work of one launch of op `name` at batch B: (flops, bytes, bound). SURVEY.md 8(d) per-frame figures x B."""
    layer, _, kind = name.partition(".")
    # conv1 / deconv4 (K = 48 / N = 3): skinny streams, HBM bound per SURVEY 8(d): frame fp32 153 600 B, 39x79x32 map, 80x160x3 map
    if layer in ("conv1", "deconv4") and kind in ("fwd", "dgrad", "wgrad"):
        big = 38400.0 * (4 if layer == "conv1" else esz)            # frames are fp32, logits / dlogits are T
        mid = 39 * 79 * 32.0 * esz
        nbytes = {"fwd": big + mid, "wgrad": big + mid, "dgrad": big + 2 * mid}[kind]   # dgrad also reads the ReLU mask
        return None, nbytes * B, "hbm"
    if layer.startswith("conv") and kind in ("fwd", "dgrad", "wgrad"):
        ih, iw, ci, co = ENC[int(layer[4]) - 1]
        oh, ow = (ih - 4) // 2 + 1, (iw - 4) // 2 + 1
        return 2.0 * oh * ow * co * 16 * ci * B, None, "mfma"
    if layer.startswith("deconv") and kind in ("fwd", "dgrad", "wgrad"):
        ih, iw, ci, co, k = DEC[int(layer[6]) - 1]
        return 2.0 * ih * iw * ci * k * k * co * B, None, "mfma"
    if layer in ("heads", "dense1") and kind in ("fwd", "dgrad", "wgrad"):
        return 2.0 * 6144 * (128 if layer == "heads" else 64) * B, None, "mfma"
    if name == "recon_loss":
        return None, 38400.0 * (2 * esz + 4) * B, "hbm"
    if name == "adam":
        return None, n_params * (28.0 + (2 if esz == 2 else 0)), "hbm"
    if kind == "bias_grad":
        if layer.startswith("conv"):
            ih, iw, ci, co = ENC[int(layer[4]) - 1]
            return None, float(((ih - 4) // 2 + 1) * ((iw - 4) // 2 + 1) * co * esz * B), "hbm"
        if layer.startswith("deconv"):
            ih, iw, ci, co, k = DEC[int(layer[6]) - 1]
            return None, float(((ih - 1) * 2 + k) * ((iw - 1) * 2 + k) * co * esz * B), "hbm"
        return None, float((6144 if layer == "dense1" else 128) * esz * B), "hbm"
    return None, float(64 * 6 * 4 * B), "hbm"           # reparam / finalize: tiny


def collect_timing(dev, n_ops):
    ms = np.zeros(n_ops, np.float32)
    cnt = np.zeros(n_ops, np.int32)
    dev.L.mi_vae_timing_collect(dev.handle, ms.ctypes.data, cnt.ctypes.data, n_ops)
    return ms, cnt


def cpu_baseline(batch, seed=0):
    """Oracle (port of the reference graph) SGD steps on the host cores: 1 warm-up + 3 timed steps of the same minibatch shape."""
    from oracle import vae_oracle as vo
    # 16 threads measured fastest for this graph on the GPU box's 2x EPYC 9575F (16: 515, 32: 474, 64: 273, 128: 151 frames/s)
    cores = min(int(os.environ.get("MI355_CPU_BASELINE_THREADS", "16")), os.cpu_count() or 1)
    torch.set_num_threads(cores)
    o = vo.OracleVAE(seed=seed)
    rng = np.random.RandomState(1234)
    frames = rng.randint(0, 256, (batch, 80, 160, 3), dtype=np.uint8).astype(np.float32) / 255.0
    eps = np.random.RandomState(4321).standard_normal((batch, 64)).astype(np.float32)
    o.train_step(frames, frames, eps)
    n, t0 = 3, time.perf_counter()
    for _ in range(n):
        o.train_step(frames, frames, eps)
    dt = time.perf_counter() - t0
    return {"value": batch * n / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d ConvVAE fp32 SGD steps at batch %d (torch-CPU oracle of the reference TF graph; TF 1.13 itself is not runnable)" % (n, batch)}


def ppo_extra(tmp, steps=3):
    """BASELINE configs[2]: PPO update on z=64 latents, horizon 128, 4 minibatch epochs of 32 -> samples/s (reported only)."""
    from ppo import PPO

    class Box:
        low, high, shape = np.array([-1.0, 0.0], np.float32), np.array([1.0, 1.0], np.float32), (2,)
    m = PPO(np.array([67]), Box(), learning_rate=1e-4, lr_decay=1.0, epsilon=0.2, value_scale=1.0, entropy_scale=0.01, initial_std=1.0,
            model_dir=os.path.join(tmp, "ppo"))
    m.init_session(init_logging=False)
    rng = np.random.RandomState(7)
    T = 128
    dev = m.dev.device
    s = torch.from_numpy((0.5 * rng.standard_normal((T, 67))).astype(np.float32)).to(dev)
    a = torch.from_numpy(rng.uniform(-1, 1, (T, 2)).astype(np.float32)).to(dev)
    R = torch.from_numpy(rng.randn(T).astype(np.float32)).to(dev)
    A = torch.from_numpy(rng.randn(T).astype(np.float32)).to(dev)

    def update():
        m.update_old_policy()
        for _ in range(4):
            perm = torch.from_numpy(np.random.RandomState(0).permutation(T)).to(dev)
            for i in range(4):
                mb = perm[i * 32:(i + 1) * 32]
                m._step_resident(s[mb].contiguous(), a[mb].contiguous(), R[mb].contiguous(), A[mb].contiguous(), 32, 32)
    update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        update()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"config": "PPO update, horizon 128, 4 epochs x 4 minibatches of 32, fp32, 1 GPU", "samples_per_s": T / dt, "ms_per_update": dt * 1e3,
            "ms_per_sgd_step": dt * 1e3 / 16}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="frames per GPU per step (BASELINE configs[1])")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--pool", type=int, default=2048, help="synthetic frames resident in HBM per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ppo", action="store_true")
    args = ap.parse_args()

    from mi355 import dist as midist
    world, rank, local = midist.init_from_env("nccl")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(local)
    from vae.models import ConvVAE

    tmp = tempfile.mkdtemp(prefix="mi355_bench_")
    B = args.batch
    model = ConvVAE(np.array([80, 160, 3]), z_dim=64, beta=1.0, learning_rate=1e-4, model_dir=os.path.join(tmp, "vae"), precision=args.precision, seed=0)
    model.init_session(init_logging=False)
    dev = model.dev
    dev.ensure_batch(B)
    device = dev.device

    # synthetic uint8-grid frames, HBM resident (randint(0,256)/255), generated on the device
    g = torch.Generator(device=device)
    g.manual_seed(1234 + rank)
    pool = (torch.randint(0, 256, (args.pool, 38400), device=device, generator=g, dtype=torch.int32).to(torch.float32) / 255.0).contiguous()
    total = args.warmup + args.steps
    idx = torch.stack([torch.randperm(args.pool, device=device, generator=g)[:B] for _ in range(total + 2)]).to(torch.int32).contiguous()
    inv_b = 1.0 / (B * world)
    n_ops = dev.L.mi_vae_op_count()
    names = [dev.L.mi_vae_op_name(i).decode() for i in range(n_ops)]
    esz = 2 if args.precision == "bf16" else 4

    def step(i):
        model._train_minibatch(pool, pool, idx[i], B, inv_b, model._eps(B))

    # ---- warm-up (untimed); the last warm-up steps run with every op bracketed by HIP events to find the dominant kernel ----
    for i in range(max(args.warmup - 2, 0)):
        step(i)
    torch.cuda.synchronize()
    dev.L.mi_vae_timing_begin(dev.handle, 1, -1, 2 * n_ops + 8)
    for i in range(max(args.warmup - 2, 0), args.warmup):
        step(i)
    if args.warmup == 0:                                  # --warmup 0: the dominant kernel is still found on one extra, untimed step
        step(total + 1)
    torch.cuda.synchronize()
    ms_all, cnt_all = collect_timing(dev, n_ops)
    per_op = {names[i]: float(ms_all[i] / cnt_all[i]) for i in range(n_ops) if cnt_all[i] > 0}
    dominant = max(per_op, key=per_op.get) if per_op else None

    # ---- timed region: exactly K steps, barrier + synchronize on both sides; the dominant op keeps its two HIP events ----
    if dominant is not None:
        dev.L.mi_vae_timing_begin(dev.handle, 2, names.index(dominant), args.steps + 4)
    midist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        step(i)
    torch.cuda.synchronize()
    midist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())
    roofline = None
    if dominant is not None:
        ms_d, cnt_d = collect_timing(dev, n_ops)
        di = names.index(dominant)
        avg_s = float(ms_d[di] / max(cnt_d[di], 1)) * 1e-3
        flops, nbytes, bound = op_work(dominant, B, esz, dev.n_flat)
        if bound == "mfma":
            peak = PEAK["mfma_bf16" if args.precision == "bf16" else "mfma_f32"]
            ach = flops / avg_s
            roofline = {"kernel": dominant, "bound": "mfma", "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": ach / peak,
                        "traffic": None, "avg_launch_ms": avg_s * 1e3, "launches_timed": int(cnt_d[di]), "algorithmic_flops_per_launch": flops}
        else:
            ach = nbytes / avg_s
            roofline = {"kernel": dominant, "bound": "hbm", "achieved": ach / 1e9, "peak": PEAK["hbm"] / 1e9, "unit": "GB/s", "frac": ach / PEAK["hbm"],
                        "traffic": None, "avg_launch_ms": avg_s * 1e3, "launches_timed": int(cnt_d[di]), "algorithmic_bytes_per_launch": nbytes}
    # HBM traffic of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, gfx950 x2 read correction),
    # measured offline on this same workload and committed under profiles/ (PMC counters cannot be read from inside the bench)
    if roofline is not None:
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            for op, rec in tj["ops"].items():
                if dominant in [x.strip().split(" ")[0] for x in op.split("/")]:
                    roofline["traffic"] = rec["hbm_bytes_per_launch"]
                    roofline["traffic_source"] = "profiles/r01_pmc_traffic.json: " + tj["provenance"]
        except Exception:
            pass
    losses = dev.losses.cpu().numpy()

    if rank == 0:
        frames_per_s = B * world * args.steps / elapsed
        step_flops = 776494080.0 * B                      # VAE train step, per frame
        out = {
            "metric": METRIC, "value": frames_per_s, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "ConvVAE SGD step (fwd+ELBO+bwd+TF-Adam), batch=%d per GPU, 160x80x3 frames, z_dim=64, rgb target (BASELINE configs[1]; global batch %d)" % (B, B * world),
                       "global_batch": B * world, "parallelism": "dp%d" % world if world > 1 else "single", "frames_resident_in_hbm": args.pool,
                       "storage": "bf16 activations/weights, fp32 accumulate, fp32 master weights+Adam" if args.precision == "bf16" else "fp32"},
            "roofline": roofline,
            "step_model_flops_utilisation": {"algorithmic_tflops_per_step": step_flops / 1e12,
                                             "achieved_tflops": step_flops * args.steps / elapsed / 1e12,
                                             "frac_of_mfma_peak": step_flops * args.steps / elapsed / PEAK["mfma_bf16" if args.precision == "bf16" else "mfma_f32"]},
            "per_op_ms": {k: round(v, 4) for k, v in sorted(per_op.items(), key=lambda kv: -kv[1])},
            "final_losses": {"reconstruction": float(losses[0]), "kl": float(losses[1])},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B)
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_ppo:
            try:
                out["ppo"] = ppo_extra(tmp)
            except Exception as e:      # the headline metric must still print
                out["ppo"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    midist.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
