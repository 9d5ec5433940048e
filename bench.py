#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its configuration, measured on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one ConvVAE SGD step (forward, ELBO, backward, gradient all-reduce when N>1, fused TF-Adam) on one minibatch
of 512 synthetic 160x80x3 frames PER GPU in bf16 (BASELINE configs[1]; global batch 512*N -> configs[3] at N=8), frames
already resident in HBM (a pool of synthetic uint8 camera frames -- the reference's own frame format, normalised to k/255
inside the kernels that read them; minibatches are gathered inside the conv1 / loss kernels; the reparameterisation noise is
drawn inside the reparameterisation kernel).  Nothing runs on the timed path but the library's own kernels.
`value` = frames/s of the whole job = N*512*K / max-over-ranks(time of exactly K steps).

Extra objects on the same JSON line:
  box           in-run calibration of THIS GPU (mi_device_probe, csrc/probe.hip): sustained bf16 MFMA TFLOP/s and the shader clock under it, HBM streaming read /
                copy TB/s, sysfs clocks / power where readable -- measured after --condition-ms of untimed steps (the DVFS ramp) and before the counted
                warm-up; `roofline.frac_of_box` prices the dominant kernel against these instead of the datasheet.
  roofline      the dominant kernel of the step (picked from a per-op HIP-event profile during warm-up), timed with HIP
                events on the launch stream over the K timed steps; achieved = algorithmic FLOPs (or bytes) / avg duration;
                `bound` = whichever of the two floors (algorithmic bytes / 8 TB/s, algorithmic FLOPs / MFMA peak) is higher.
  cpu_baseline  the CPU oracle (a port of the reference's TF graph; the reference itself needs TensorFlow 1.13) timed on
                this box's host cores: >= 3 warm-up + >= 10 timed steps of the same minibatch shape (rank 0, N=1 only);
                `ppo` inside it: the oracle's PPO update (configs[2]) timed the same way.
  parity        measured in THIS run: the benchmarked engine (and the fp32 engine) against the oracle on the cpu_baseline's
                batch-512 inputs: relative deviation of the two losses and of encode() (max-normalised).
  fp32          throughput of the exact-fp32 mode (the mode that meets the 1e-4 tolerance) on the same workload.
  ppo           PPO update throughput on z=64 latents (BASELINE configs[2]) — reported, not part of `value`.
  replay        BASELINE configs[4] on this GPU: VAE encode + GAE + PPO minibatch SGD over 1024 trajectories x 128 steps.
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
PEAK = {"mfma_bf16": 2.5e15, "mfma_f32": 157.3e12, "hbm": 8.0e12,
        "mfma_bf16x3": 2.5e15 / 4}      # split storage: 2 bf16 MFMAs per 8 k-values instead of 1 per 16 -> a quarter of the bf16 rate per algorithmic FLOP       # /opt/skills/guides/MI355X_MICROARCH.md (dense, no sparsity)

ENC = [(80, 160, 3, 32), (39, 79, 32, 64), (18, 38, 64, 128), (8, 18, 128, 256)]          # IH, IW, Cin, Cout  (k4 s2)
DEC = [(3, 8, 256, 128, 4), (8, 18, 128, 64, 4), (18, 38, 64, 32, 5), (39, 79, 32, 3, 4)]  # IH, IW, Cin, Cout, k


def op_work(name, B, esz, n_params, frame_bytes=4, tail_fused=False, enc_fused=False):
    """Algorithmic work of one launch of op `name` at batch B: (flops, bytes). SURVEY.md 8(d) per-frame figures x B; bytes = every operand
    read once + the result written once (+ the ReLU-grad mask tensor for input gradients), weights included.  frame_bytes: 4 = fp32 frame
    tables, 1 = uint8 tables."""
    layer, _, kind = name.partition(".")
    if layer.startswith("conv") and kind in ("fwd", "dgrad", "wgrad"):
        ih, iw, ci, co = ENC[int(layer[4]) - 1]
        oh, ow = (ih - 4) // 2 + 1, (iw - 4) // 2 + 1
        flops = 2.0 * oh * ow * co * 16 * ci * B
        xin = ih * iw * ci * (frame_bytes if layer == "conv1" else esz) * B
        yout = oh * ow * co * esz * B
        w = 16 * ci * co * (esz if kind != "wgrad" else 4)
        if layer == "conv2" and kind == "dgrad" and enc_fused:
            # the encoder head of backward in one launch (enchead_bwd_kernel): conv2's input gradient + conv1's filter / bias gradient; reads dy2, the ReLU bit words of
            # conv1's output (8 bytes per pixel) and the frames once; the gradient of conv1's output stays on chip
            f1 = 2.0 * ih * iw * ci * 48 * B
            return flops + f1, float(yout + ih * iw * 8 * B + 80 * 160 * 3 * frame_bytes * B + w + 48 * 32 * 4)
        nbytes = {"fwd": xin + yout + w, "dgrad": yout + 2 * xin + w, "wgrad": xin + yout + w}[kind]     # dgrad: dy in, mask in, dx out
        return flops, float(nbytes)
    if layer.startswith("deconv") and kind in ("fwd", "dgrad", "wgrad"):
        ih, iw, ci, co, k = DEC[int(layer[6]) - 1]
        oh, ow = (ih - 1) * 2 + k, (iw - 1) * 2 + k
        flops = 2.0 * ih * iw * ci * k * k * co * B
        xin, yout = ih * iw * ci * esz * B, oh * ow * co * esz * B
        w = k * k * ci * co * (esz if kind != "wgrad" else 4)
        if layer == "deconv4" and kind == "fwd" and tail_fused:
            # the decoder tail in one launch (dectail_kernel): deconv4 forward + loss + its input gradient + its filter gradient = 3 x the layer's
            # FLOPs; reads deconv3's output and the labels once, writes the gradient of deconv3's output; logits and dlogits stay on chip
            return 3.0 * flops, float(2 * xin + oh * ow * co * frame_bytes * B + 2 * w)
        if layer == "deconv4" and kind == "fwd":         # fused with the reconstruction loss: labels in, dlogits out, the logits stay on chip
            return flops, float(xin + oh * ow * co * frame_bytes * B + yout + w)
        nbytes = {"fwd": xin + yout + w, "dgrad": yout + 2 * xin + w, "wgrad": xin + yout + w}[kind]
        return flops, float(nbytes)
    if layer in ("heads", "dense1") and kind in ("fwd", "dgrad", "wgrad"):
        n = 128 if layer == "heads" else 64
        return 2.0 * 6144 * n * B, float((6144 + n) * esz * B + 6144 * n * (esz if kind != "wgrad" else 4))
    if name == "recon_loss":
        return None, 38400.0 * (2 * esz + frame_bytes) * B
    if name == "adam":
        return None, n_params * (28.0 + (2 if esz == 2 else 0))
    if kind == "bias_grad":
        if layer.startswith("conv"):
            ih, iw, ci, co = ENC[int(layer[4]) - 1]
            return None, float(((ih - 4) // 2 + 1) * ((iw - 4) // 2 + 1) * co * esz * B)
        if layer.startswith("deconv"):
            ih, iw, ci, co, k = DEC[int(layer[6]) - 1]
            return None, float(((ih - 1) * 2 + k) * ((iw - 1) * 2 + k) * co * esz * B)
        return None, float((6144 if layer == "dense1" else 128) * esz * B)
    return None, float(64 * 6 * 4 * B)                  # reparam / finalize: tiny


def roofline_of(name, avg_s, B, esz, n_params, precision, frame_bytes, tail_fused=False, enc_fused=False):
    """Roofline object of one op: the bound is decided by comparing the arithmetic intensity with the ridge point."""
    flops, nbytes = op_work(name, B, esz, n_params, frame_bytes, tail_fused, enc_fused)
    peak_f = PEAK[{"bf16": "mfma_bf16", "bf16x3": "mfma_bf16x3"}.get(precision, "mfma_f32")]
    t_hbm = nbytes / PEAK["hbm"]
    t_mfma = (flops or 0.0) / peak_f
    out = {"kernel": name, "avg_launch_ms": avg_s * 1e3, "algorithmic_bytes_per_launch": nbytes, "algorithmic_flops_per_launch": flops,
           "arithmetic_intensity_flop_per_byte": (flops / nbytes) if flops else None, "ridge_flop_per_byte": peak_f / PEAK["hbm"],
           "floor_ms": {"hbm": t_hbm * 1e3, "mfma": t_mfma * 1e3}, "traffic": None}
    if t_mfma > t_hbm:
        ach = flops / avg_s
        out.update({"bound": "mfma", "achieved": ach / 1e12, "peak": peak_f / 1e12, "unit": "TFLOP/s", "frac": ach / peak_f})
    else:
        ach = nbytes / avg_s
        out.update({"bound": "hbm", "achieved": ach / 1e9, "peak": PEAK["hbm"] / 1e9, "unit": "GB/s", "frac": ach / PEAK["hbm"]})
        if flops:
            out["mfma_frac_for_reference"] = flops / avg_s / peak_f
    return out


def collect_timing(dev, n_ops):
    ms = np.zeros(n_ops, np.float32)
    cnt = np.zeros(n_ops, np.int32)
    dev.L.mi_vae_timing_collect(dev.handle, ms.ctypes.data, cnt.ctypes.data, n_ops)
    return ms, cnt


def cpu_inputs(batch):
    frames = np.random.RandomState(1234).randint(0, 256, (batch, 80, 160, 3), dtype=np.uint8)
    eps = np.random.RandomState(4321).standard_normal((batch, 64)).astype(np.float32)
    return frames, eps


def cpu_baseline(batch, seed=0, warm=3, timed=10):
    """Oracle (port of the reference graph) SGD steps on the host cores (>= 3 warm-up + >= 10 timed steps of the same
    minibatch shape) + the oracle's PPO update (configs[2]); also returns what the parity object needs (oracle losses / encodings)."""
    from oracle import ppo_oracle as po
    from oracle import vae_oracle as vo
    # 16 threads measured fastest for this graph on the GPU box's 2x EPYC 9575F (16: 515, 32: 474, 64: 273, 128: 151 frames/s)
    cores = min(int(os.environ.get("MI355_CPU_BASELINE_THREADS", "16")), os.cpu_count() or 1)
    torch.set_num_threads(cores)
    u8, eps = cpu_inputs(batch)
    frames = u8.astype(np.float32) / 255.0
    params = vo.init_vae_params(seed)
    # the checker's side of the parity object (computed here: every oracle call of this file sits in this one function): one batch-512 step on "trained-like"
    # parameters (Glorot weights + small random biases, the inputs of tests/test_a_c2_b512_gpu.py: with the all-zero biases of a fresh initialisation half of the
    # decoder's pre-activations sit within an ulp of the ReLU threshold and no two fp32 evaluations agree on their masks) -- losses, posterior means, the gradients in
    # float32, float64 and under bf16-storage emulation, and TF-Adam's first update from the float32 / emulated gradients
    tl = {k: v.copy() for k, v in params.items()}
    brng = np.random.RandomState(seed + 1)
    for k in tl:
        if k.endswith("bias"):
            tl[k] = (0.05 * brng.standard_normal(tl[k].shape)).astype(np.float32)
    (recon, kl, _), g32, fw = vo.vae_loss_and_grads(tl, frames, frames, eps)
    _, g64, _ = vo.vae_loss_and_grads(tl, frames, frames, eps, beta=1.0, dtype=torch.float64)
    _, gemu, _ = vo.vae_loss_and_grads(tl, frames, frames, eps, beta=1.0, storage="bf16")
    ref = {"params": tl, "recon": recon, "kl": kl, "mean": fw["mean"].numpy(), "g32": g32, "g64": g64, "gemu": gemu, "adam": {}}
    for name, gg in (("g32", g32), ("gemu", gemu)):
        want = {k: v.copy() for k, v in tl.items()}
        vo.AdamTF({k: v.shape for k, v in tl.items()}).step(want, gg, 1e-4)
        ref["adam"][name] = want
    o = vo.OracleVAE(params={k: v.copy() for k, v in params.items()})
    for _ in range(warm):
        o.train_step(frames, frames, eps)
    t0 = time.perf_counter()
    for _ in range(timed):
        o.train_step(frames, frames, eps)
    dt = time.perf_counter() - t0
    out = {"value": batch * timed / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "cores_total": os.cpu_count(), "kind": "port",
           "cores_note": "`cores` = threads actually used (the fastest setting measured for this graph on the 2-socket host; MI355_CPU_BASELINE_THREADS overrides), "
                         "`cores_total` = os.cpu_count() of this box",
           "sample": "%d warm-up + %d timed ConvVAE fp32 SGD steps at batch %d (torch-CPU oracle of the reference TF graph; TF 1.13 itself is not runnable)" % (warm, timed, batch)}
    # PPO update on the same cores: horizon 128, 4 epochs x 4 minibatches of 32 (configs[2])
    hp = dict(learning_rate=1e-4, lr_decay=1.0, epsilon=0.2, value_scale=1.0, entropy_scale=0.01, initial_std=1.0)
    m = po.OraclePPO([67], po.ActionSpace(), seed=1, **hp)
    rng = np.random.RandomState(7)
    T = 128
    s = (0.5 * rng.standard_normal((T, 67))).astype(np.float32)
    a = rng.uniform(-1, 1, (T, 2)).astype(np.float32)
    R, A = rng.randn(T).astype(np.float32), rng.randn(T).astype(np.float32)

    def update():
        m.update_old_policy()
        for mb in po.minibatch_schedule(T, 32, 4):
            m.train(s[mb], a[mb], R[mb], A[mb])
    for _ in range(3):
        update()
    t0 = time.perf_counter()
    for _ in range(10):
        update()
    dtp = (time.perf_counter() - t0) / 10
    out["ppo"] = {"samples_per_s": T / dtp, "ms_per_update": dtp * 1e3, "sample": "3 warm-up + 10 timed PPO updates (horizon 128, 4 epochs x 4 minibatches of 32), torch-CPU oracle, same cores"}
    return out, ref


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def parity_object(tmp, ref, batch):
    """The HIP engines against the oracle on the cpu_baseline's inputs, measured in this run: forward losses + encode of one batch-512 pass, and (round 4,
    VERDICT r03 item 2) the GRADIENTS and the TF-Adam update of that step under the criteria of tests/test_a_c2_b512_gpu.py -- `grad_worst` is the worst
    per-tensor ratio measured / allowed (<= 1 passes), `adam_off_fraction` the largest per-tensor fraction of weights whose first Adam update differs from the
    oracle's by more than `adam_lim_of_lr` x lr."""
    from vae.models import ConvVAE
    u8, eps = cpu_inputs(batch)
    frames = u8.astype(np.float32) / 255.0
    params = ref["params"]
    out = {"inputs": "batch %d, frames / noise seeds 1234 / 4321 (SURVEY 8d), parameters: Glorot weights (seed 0) + N(0, 0.05) biases -- the inputs of tests/test_a_c2_b512_gpu.py; "
                     "oracle = torch-CPU port of the reference graph run in float32, float64 and under bf16-storage emulation (cpu_baseline leg)" % batch}
    g32, g64, gemu = ref["g32"], ref["g64"], ref["gemu"]       # float32 oracle, exact (float64 run of the same graph), bf16-STORAGE emulation (the bf16 engine's rounding points)
    e_o32 = {k: _rel(g32[k], g64[k]) for k in g32}            # the fp32 oracle's own distance from exact
    for prec in ("bf16", "bf16x3", "fp32"):
        m = ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=os.path.join(tmp, "parity_" + prec), precision=prec, seed=0)
        m.set_weights(params)
        m.init_session(init_logging=False)
        src = m._frames(frames, 38400, "src")
        e = m._eps(batch, eps)
        m.dev.forward(src, src, None, batch, 1.0 / batch, e, 1, 1)
        got = m.dev.losses.cpu().numpy().copy()
        mean = m.dev._view(1, batch * 64).cpu().numpy().reshape(batch, 64).copy()
        m.dev.backward(src, None, e, 1.0 / batch, 0)
        g = m.dev.export_grads()
        if prec == "bf16":        # as close to the exact-fp32 gradients as the emulation is: e_dev <= 1.25 e_emul + 2e-3 of the tensor max
            ratios = {k: _rel(g[k], g32[k]) / (1.25 * _rel(gemu[k], g32[k]) + 2e-3) for k in g32}
            crit, want_g, lim, frac_lim = "e_dev / (1.25 e_emul + 2e-3), distances to the exact-fp32 gradient in units of the tensor max", "gemu", 0.5, 0.05
        else:                     # distance to the float64 gradient: <= max(floor, factor x the fp32 oracle's own distance)
            floor, factor = (2e-4, 2.0) if prec == "fp32" else (1e-3, 4.0)
            ratios = {k: _rel(g[k], g64[k]) / max(floor, factor * e_o32[k]) for k in g32}
            crit, want_g = "e_dev / max(%.0e, %.0f x the fp32 oracle's own distance), distances to the float64 gradient in units of the tensor max" % (floor, factor), "g32"
            lim, frac_lim = (0.02, 5e-3) if prec == "fp32" else (0.05, 2e-2)
        want = ref["adam"][want_g]
        m._adam_step()
        got_p = m.dev.export_params()
        off = {k: float(np.mean(np.abs((got_p[k] - params[k]) - (want[k] - params[k])) > lim * 1e-4)) for k in want}
        worst_k = max(ratios, key=ratios.get)
        out[prec] = {"recon_loss_rel": float(abs(got[0] / ref["recon"] - 1)), "kl_loss_rel": float(abs(got[1] / ref["kl"] - 1)),
                     "encode_rel_of_max": float(np.abs(mean - ref["mean"]).max() / np.abs(ref["mean"]).max()),
                     "grad_worst": float(ratios[worst_k]), "grad_worst_tensor": worst_k, "grad_criterion": crit,
                     "grad_rel_of_max_vs_fp32_oracle": float(max(_rel(g[k], g32[k]) for k in g32)),
                     "adam_off_fraction": float(max(off.values())), "adam_off_fraction_limit": frac_lim, "adam_lim_of_lr": lim}
        m.dev.close()
    out["note"] = ("fp32 = exact-fp32 MFMA engine (the drop-in's default; north_star's 1e-4); bf16x3 = split storage (every element two bf16 halves hi + lo, "
                   "products on the bf16 MFMA pipe as hi/lo partial products, fp32 accumulate): the fast mode that meets the 1e-4; bf16 = the benchmarked throughput mode (bf16 storage, fp32 "
                   "accumulate): its deviation is bf16 rounding of activations / weights, stated here instead of claimed away; TF's own fp32 kernel "
                   "rounding is unpinned (TF 1.13 not runnable): the oracle is pinned to the reference's serialized graphs at 1e-9 in float64")
    return out


def ppo_extra(tmp, steps=5):
    """BASELINE configs[2]: PPO update on z=64 latents, horizon 128, 4 minibatch epochs of 32 -> samples/s (reported only)."""
    from ppo import PPO

    class Box:
        low, high, shape = np.array([-1.0, 0.0], np.float32), np.array([1.0, 1.0], np.float32), (2,)
    m = PPO(np.array([67]), Box(), learning_rate=1e-4, lr_decay=1.0, epsilon=0.2, value_scale=1.0, entropy_scale=0.01, initial_std=1.0,
            model_dir=os.path.join(tmp, "ppo"), seed=0)
    m.init_session(init_logging=False)
    rng = np.random.RandomState(7)
    T = 128
    dev = m.dev.device
    s = torch.from_numpy((0.5 * rng.standard_normal((T, 67))).astype(np.float32)).to(dev)
    a = torch.from_numpy(rng.uniform(-1, 1, (T, 2)).astype(np.float32)).to(dev)
    R = torch.from_numpy(rng.randn(T).astype(np.float32)).to(dev)
    A = torch.from_numpy(rng.randn(T).astype(np.float32)).to(dev)

    lp = torch.empty(T, device=dev)

    def update():
        # train.py:192-204 with the horizon batch resident on the device: theta_old <- theta, log pi_old(a|s) of the batch once, then 4 epochs x 4 shuffled
        # minibatches of 32 whose rows are gathered INSIDE the step's kernels (mi_ppo_train_step_idx)
        m.update_old_policy()
        m.dev.logp_old(s, a, T, lp)
        for _ in range(4):
            perm = torch.from_numpy(np.random.RandomState(0).permutation(T).astype(np.int32)).to(dev)
            for i in range(4):
                m._step_rows(s, a, R, A, lp, perm[i * 32:(i + 1) * 32], 32, 32)
    update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        update()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # a PPO minibatch step moves 284 B per sample + 369,505 parameters x 24 B (theta, m, v, g read; theta, m, v written) and does 2,441,600 FLOP per sample:
    # at minibatch 32 both floors are ~1 us -- the step is bound by the latency of its five dependent launches, priced here against HBM as SURVEY says to
    step_s = dt / 16
    step_bytes = 369505 * 24.0 + 284.0 * 32
    step_flops = 2441600.0 * 32
    return {"config": "PPO update, horizon 128, 4 epochs x 4 minibatches of 32, fp32, 1 GPU", "samples_per_s": T / dt, "ms_per_update": dt * 1e3,
            "ms_per_sgd_step": dt * 1e3 / 16,
            "roofline": {"kernel": "PPO minibatch SGD step (five dependent launches: mi_ppo_train_step_idx)", "bound": "launch latency (priced against HBM, SURVEY 8d)",
                         "algorithmic_bytes_per_step": step_bytes, "algorithmic_flops_per_step": step_flops, "achieved": step_bytes / step_s / 1e9, "peak": PEAK["hbm"] / 1e9,
                         "unit": "GB/s", "frac": step_bytes / step_s / PEAK["hbm"], "floor_ms": {"hbm": step_bytes / PEAK["hbm"] * 1e3, "mfma_f32": step_flops / PEAK["mfma_f32"] * 1e3},
                         "launches_per_step": 5, "us_per_launch": step_s * 1e6 / 5,
                         "note": "a dependent kernel boundary in this chain costs 1.5-1.6 us (profiles/r05_ppo.md): five launches = ~8 us of the step are boundaries, the rest "
                                 "is five kernels of 32-row matrices on an empty chip"}}


def fp32_extra(tmp, B, pool_u8, idx, steps=40, warm=5, precision="fp32"):
    """The same workload on the exact-fp32 engine (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak) or on the split-storage engine (precision "bf16x3":
    hi/lo bf16 halves, two bf16 MFMAs per 8 k-values): the two modes that meet north_star's 1e-4."""
    from vae.models import ConvVAE
    m = ConvVAE(np.array([80, 160, 3]), z_dim=64, beta=1.0, learning_rate=1e-4, model_dir=os.path.join(tmp, "vae_" + precision), precision=precision, seed=0)
    m.init_session(init_logging=False)
    m.dev.ensure_batch(B)
    n = min(pool_u8.shape[0], 1024)
    pool = torch.empty(n, 38400, device=pool_u8.device)
    m.dev.L.mi_u8_to_unit_f32(m.dev.stream(), pool_u8.data_ptr(), pool.data_ptr(), pool.numel())
    sel = (idx.to(torch.int64) % n).to(torch.int32).contiguous()
    for i in range(warm):
        m._train_minibatch(pool, pool, sel[i], B, 1.0 / B, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        m._train_minibatch(pool, pool, sel[(warm + i) % sel.shape[0]], B, 1.0 / B, None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fl = 776494080.0 * B * steps / dt
    res = {"frames_per_s": B * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps}
    if precision == "fp32":
        res.update({"frac_of_157TF": fl / PEAK["mfma_f32"], "storage": "fp32 activations / weights, exact-fp32 MFMA; fp32 frame table"})
    else:
        # MFMA work of the split mode: 2 bf16 MFMAs per 8 k-values = 4x the bf16 engine's issue time for the same algorithmic FLOPs
        res.update({"algorithmic_tflops": fl / 1e12, "bf16_mfma_issue_frac": 4.0 * fl / PEAK["mfma_bf16"],
                    "storage": "split: activations / weights as hi | lo bf16 halves (4 bytes), a.b + a.swap16(b) on v_mfma_f32_32x32x16_bf16, fp32 accumulate; fp32 master weights + Adam; fp32 frame table"})
        n_ops = m.dev.L.mi_vae_op_count()
        names = [m.dev.L.mi_vae_op_name(i).decode() for i in range(n_ops)]
        m.dev.L.mi_vae_timing_begin(m.dev.handle, 1, -1, 2 * n_ops + 8)
        for i in range(2):
            m._train_minibatch(pool, pool, sel[i], B, 1.0 / B, None)
        torch.cuda.synchronize()
        ms_all, cnt_all = collect_timing(m.dev, n_ops)
        res["per_op_ms"] = {names[i]: round(float(ms_all[i] / cnt_all[i]), 4) for i in np.argsort(-ms_all) if cnt_all[i] > 0}
    m.dev.close()
    return res


def mlp_extra(tmp, B, pool_u8, idx, steps=30, warm=5):
    """the MlpVAE (vae/models.py:271-299, 38400-512-256 | 64 | 256-512-38400) SGD step on the same frames, bf16 storage."""
    from vae.models import MlpVAE
    m = MlpVAE(np.array([80, 160, 3]), z_dim=64, model_dir=os.path.join(tmp, "mlp"), precision="bf16", seed=0)
    m.init_session(init_logging=False)
    m.dev.ensure_batch(B)
    n = min(pool_u8.shape[0], 1024)
    pool = pool_u8[:n]                                    # round 5: the uint8 camera-byte table itself (normalised where the minibatch rows are staged and in the loss kernel)
    sel = (idx.to(torch.int64) % n).to(torch.int32).contiguous()
    for i in range(warm):
        m._train_minibatch(pool, pool, sel[i], B, 1.0 / B, m._eps(B))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        m._train_minibatch(pool, pool, sel[(warm + i) % sel.shape[0]], B, 1.0 / B, m._eps(B))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flops = 6.0 * B * (38400 * 512 + 512 * 256 + 256 * 128 + 64 * 256 + 256 * 512 + 512 * 38400)
    return {"frames_per_s": B * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "dense_tflops": flops * steps / dt / 1e12,
            "frame_table": "uint8 camera bytes, k/255 while the rows are staged / in the loss kernel",
            "note": "native MlpVAE engine (csrc/mlp_engine.hip): one C call per step; 39.5 M parameters: the Adam pass alone moves 1.1 GB per step"}


def replay_extra(tmp, rows, T=128, batch=2048, epochs=4):
    """BASELINE configs[4] on ONE GPU: encode 1024 x 129 uint8 frames, values, GAE, PPO minibatch SGD (global minibatch 2048)."""
    import replay
    from ppo import PPO
    from vae.models import ConvVAE

    class Box:
        low, high, shape = np.array([-1.0, 0.0], np.float32), np.array([1.0, 1.0], np.float32), (2,)
    vae = ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=os.path.join(tmp, "rvae"), precision="bf16", training=False, seed=0)
    vae.init_session(init_logging=False)
    ppo = PPO(np.array([67]), Box(), learning_rate=1e-4, lr_decay=1.0, epsilon=0.2, value_scale=1.0, entropy_scale=0.01, initial_std=1.0,
              model_dir=os.path.join(tmp, "rppo"), seed=0)
    ppo.init_session(init_logging=False)
    rng = np.random.default_rng(1234)
    frames = rng.integers(0, 256, (rows, T + 1, 80, 160, 3), dtype=np.uint8)
    meas = np.stack([rng.uniform(-1, 1, (rows, T + 1)), rng.uniform(0, 1, (rows, T + 1)), rng.uniform(0, 30, (rows, T + 1))], axis=-1).astype(np.float32)
    actions = np.stack([rng.uniform(-1, 1, (rows, T)), rng.uniform(0, 1, (rows, T))], axis=-1).astype(np.float32)
    rewards, dones = rng.uniform(0, 1, (rows, T)), np.zeros((rows, T))
    stages = {}
    replay.replay_update(vae, ppo, frames[:8], meas[:8], actions[:8], rewards[:8], dones[:8], 0.99, 0.95, 1, batch)      # warm-up: engines sized
    # ... and one untimed update of the FULL shape, so that both timed variants below run warm (the host-fed figure used to come from the cold first run and
    # the resident one from the warm second -- not comparable as presented)
    replay.replay_update(vae, ppo, frames, meas, actions, rewards, dones, 0.99, 0.95, 1, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = replay.replay_update(vae, ppo, frames, meas, actions, rewards, dones, 0.99, 0.95, epochs, batch, stage_times=stages)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = rows * T
    res = {"config": "synthetic replay (BASELINE configs[4] on 1 GPU): %d trajectories x %d steps, bf16 VAE encode of %d uint8 frames, values, GAE + per-row "
                     "normalisation, PPO SGD %d epochs x minibatch %d (fp32)" % (rows, T, rows * (T + 1), epochs, batch),
           "sgd_steps": len(out["losses"]), "last_loss": out["losses"][-1]["loss"] if out["losses"] else None}
    # the same update with the frame table already resident in HBM (uploaded outside the timed region: the bench contract's form) ...
    table = torch.from_numpy(frames.reshape(rows * (T + 1), -1)).to("cuda").view(rows, T + 1, 80, 160, 3)
    torch.cuda.synchronize()
    stages_r = {}
    t0 = time.perf_counter()
    out_r = replay.replay_update(vae, ppo, table, meas, actions, rewards, dones, 0.99, 0.95, epochs, batch, stage_times=stages_r)
    torch.cuda.synchronize()
    dt_r = time.perf_counter() - t0
    res["resident"] = {"seconds": dt_r, "samples_per_s": n / dt_r, "frames": "uint8 table resident in HBM before the timed region (%.2f GB)" % (table.numel() / 1e9),
                       "encode_frames_per_s": rows * (T + 1) / stages_r["encode"] if "encode" in stages_r else None}
    res["resident"].update({k + "_s": v for k, v in stages_r.items()})
    # ... and handed over as a host array (what a caller that holds the recording in host memory pays: a PCIe figure)
    res["from_host"] = {"seconds": dt, "samples_per_s": n / dt, "includes": "host->device upload of %.2f GB of uint8 frames over PCIe inside the encode stage" % (frames.nbytes / 1e9),
                        "encode_frames_per_s": rows * (T + 1) / stages["encode"] if "encode" in stages else None}
    res["from_host"].update({k + "_s": v for k, v in stages.items()})
    res["seconds"], res["samples_per_s"] = dt_r, n / dt_r
    if "sgd" in stages_r:
        res["ppo_sgd_samples_per_s"] = n * epochs / stages_r["sgd"]
        # the large-minibatch PPO step: 2,441,600 FLOP per sample in exact fp32 (157 TFLOP/s) against 8.87 MB of optimiser traffic + 284 B per sample per step
        sgd_steps = max(len(out_r["losses"]), 1)
        step_s = stages_r["sgd"] / sgd_steps
        res["ppo_sgd_roofline"] = {"minibatch": batch, "ms_per_step": step_s * 1e3, "mfma_f32_frac": 2441600.0 * batch / step_s / PEAK["mfma_f32"],
                                   "hbm_frac": (369505 * 24.0 + 284.0 * batch) / step_s / PEAK["hbm"], "bound": "launch latency / exact-fp32 MFMA (SURVEY 8d)"}
    efps = res["resident"].get("encode_frames_per_s")
    if efps:
        # 118,778,880 FLOP per encoded frame (bf16 MFMA), 38,400 camera bytes in + 256 B out
        res["encode_roofline"] = {"frames_per_s": efps, "achieved_tflops": efps * 118778880.0 / 1e12, "mfma_bf16_frac": efps * 118778880.0 / PEAK["mfma_bf16"],
                                  "hbm_frac_algorithmic": efps * (38400.0 + 256.0) / PEAK["hbm"], "bound": "mfma (AI = 3,073 FLOP / B on uint8 frames)"}
    del out_r
    return res


def step_traffic(B):
    """HBM bytes of one WHOLE SGD step (PMC FETCH_SIZE x 2 + WRITE_SIZE summed over every kernel of a step, from the committed rocprofv3 passes of this round) next to SURVEY
    8(d)'s two reference points: the algorithmic bytes (frames + noise + optimiser traffic) and the practical un-fused activation traffic (every activation written in forward and
    read in backward, the same for the gradients, bf16).  Counters cannot be read from inside the bench: the figure is the profile's, for batch 512."""
    alg = B * (38400.0 + 256.0) + 2584387 * 24.0         # this path: uint8 camera bytes + noise per frame, 24 B of optimiser traffic per parameter
    alg_survey = B * (153600.0 + 256.0) + 2584387 * 24.0  # SURVEY 8(d) as written: float32 frames (0.141 GB at batch 512)
    practical = B * 3.0e6 + 2584387 * 24.0
    out = {"algorithmic_gb": alg / 1e9, "algorithmic_gb_survey_fp32_frames": alg_survey / 1e9, "practical_unfused_gb": practical / 1e9, "pmc_gb_per_step": None, "source": None}
    for fn in ("r06_pmc_traffic.json", "r05_pmc_traffic.json"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", fn)))
            st = tj.get("step")
            if st:
                out["pmc_gb_per_step"], out["source"] = st["hbm_bytes"] / 1e9, "profiles/%s (%d kernels of one step, batch %d)" % (fn, st["kernels"], st.get("batch", 512))
            else:                                      # (older files: per-op records only -- every op runs once per step except the two averaged pairs)
                tot = sum(r["hbm_bytes_per_launch"] * (2 if "averaged" in op else 1) for op, r in tj["ops"].items())
                out["pmc_gb_per_step"], out["source"] = tot / 1e9, "profiles/%s (sum of its per-op records)" % fn
            break
        except Exception:
            continue
    if out["pmc_gb_per_step"]:
        out["x_algorithmic"], out["x_practical"] = out["pmc_gb_per_step"] / out["algorithmic_gb_survey_fp32_frames"], out["pmc_gb_per_step"] / out["practical_unfused_gb"]
    return out


def _watchdog(seconds, last_words):
    """A timer that ends THIS process (exit code 0) after `seconds` unless cancelled; rank 0 passes `last_words` (prints the result line it already has)."""
    import threading

    def fire():
        try:
            if last_words is not None:
                last_words()
        finally:
            os._exit(0)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def dp_schedules(step, n_steps, world, rank, device):
    """Data-parallel runs only: the same SGD steps under BOTH gradient-bucket schedules of the library communicator -- ncclAllReduce and one-hop reduce-scatter + all-gather
    (csrc/comm.hip) -- timed the same way in one run, and the number of ranks RCCL itself reports (the first multi-GPU run has something to compare).
    A schedule that has never moved a byte between two devices must not be able to take the headline down with it: a watchdog ends every rank of a leg that does not come
    back (rank 0 has printed nothing yet; main() prints the line without this object in that case -- see _watchdog)."""
    from mi355 import dist as midist
    from mi355 import lib as milib
    c = midist.mi_comm()
    if c is None:
        return {"schedules": None, "schedules_note": "torch.distributed carries the buckets (no library communicator): one schedule only"}
    L = milib.get()
    out = {"ranks_seen": int(L.mi_comm_ranks(c.handle))}
    has = torch.tensor([int(L.mi_comm_has_rsag())], dtype=torch.int32, device=device)
    torch.distributed.all_reduce(has, op=torch.distributed.ReduceOp.MIN)
    start_algo = 1 if "reduce-scatter" in midist.comm_note else 0
    res = {}
    for algo, name in ((0, "ncclAllReduce"), (1, "reduce_scatter_all_gather")):
        if algo == 1 and not int(has.item()):
            res[name] = None
            continue
        torch.cuda.synchronize(); midist.barrier()
        L.mi_comm_set_algo(c.handle, algo)                # (every rank, at the same point of the program: the schedule is a property of the communicator)
        for i in range(3):
            step(i)
        torch.cuda.synchronize(); midist.barrier()
        t0 = time.perf_counter()
        for i in range(n_steps):
            step(i)
        torch.cuda.synchronize()
        midist.barrier()
        t = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        res[name] = float(t[0].item()) / n_steps * 1e3
    torch.cuda.synchronize(); midist.barrier()
    L.mi_comm_set_algo(c.handle, start_algo)
    out["schedules_ms_per_step"] = res
    out["schedule_of_the_timed_region"] = "reduce_scatter_all_gather" if start_algo else "ncclAllReduce"
    return out


def _library_stamp():
    """ABI version + modification time of the loaded libmi355_carla.so: an A/B run states WHICH build it measured (DESIGN finding 28)."""
    from mi355 import lib as milib
    L = milib.get()
    path = milib.LIB_PATH
    return {"abi": int(L.mi_abi_version()), "built": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime(os.path.getmtime(path))) if path and os.path.exists(path) else None}


def _sysfs_gpu(local):
    """Best-effort readings of the amdgpu sysfs files of this rank's GPU (absent in some containers: every field may be missing): current shader / memory clock,
    power cap and average power.  Never fatal."""
    out = {}
    try:
        import glob
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
        if not cards:
            return out
        base = os.path.dirname(cards[min(local, len(cards) - 1)])

        def starred(fn):
            for line in open(os.path.join(base, fn)).read().splitlines():
                if line.strip().endswith("*"):
                    return float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
            return None
        for key, fn in (("sysfs_sclk_mhz", "pp_dpm_sclk"), ("sysfs_mclk_mhz", "pp_dpm_mclk")):
            try:
                v = starred(fn)
                if v is not None:
                    out[key] = v
            except Exception:
                pass
        for hw in glob.glob(os.path.join(base, "hwmon", "hwmon*")):
            for key, fn, scale in (("power_cap_w", "power1_cap", 1e-6), ("power_avg_w", "power1_average", 1e-6), ("power_now_w", "power1_input", 1e-6)):
                try:
                    out[key] = float(open(os.path.join(hw, fn)).read().strip()) * scale
                except Exception:
                    pass
    except Exception:
        pass
    return out


def box_probe(dev, local=0, millis=5):
    """In-run box calibration: the library's own MFMA / HBM probes (csrc/probe.hip, mi_device_probe) on this GPU, right now -- what the
    datasheet peaks of PEAK are worth on the box that produced this line.  ~60 ms of GPU time, outside the timed region."""
    L = dev.L
    want = int(L.mi_device_probe_scratch_bytes())
    free = torch.cuda.mem_get_info()[0]
    nbytes = max(min(want, int(free * 0.5)), 64 << 20)
    scratch = torch.empty(nbytes, device=dev.device, dtype=torch.uint8)
    out8 = np.zeros(8, np.float32)
    torch.cuda.synchronize()
    L.mi_device_probe(dev.stream(), scratch.data_ptr(), nbytes, millis, out8.ctypes.data)
    del scratch
    box = {"mfma_bf16_tflops": round(float(out8[0]), 1), "sclk_mhz": round(float(out8[1]), 0),
           "mfma_bf16_tflops_first_launch": round(float(out8[2]), 1), "sclk_mhz_first_launch": round(float(out8[3]), 0),
           "hbm_read_tbps": round(float(out8[4]), 3), "hbm_copy_tbps": round(float(out8[5]), 3), "compute_units": int(out8[6]),
           "hbm_probe_bytes": int(out8[7]),
           "how": "mi_device_probe: 3 x %d ms of v_mfma_f32_32x32x16_bf16 (4 accumulators, 2 waves / SIMD, 1 block / CU; rate = mean of launches 2-3; sclk = s_memtime / "
                  "s_memrealtime of one wave), 16-byte streaming read and half-to-half copy of the probe buffer (best of 3)" % millis}
    box.update(_sysfs_gpu(local))
    return box


def condition_clocks(step, target_ms, world, device, chunk=20, sync=None):
    """Untimed steps in chunks of `chunk` until `target_ms` of wall time have passed; returns how many ran.  step(i) runs step number i.
    Every step of a data-parallel run is a collective, so the ranks must leave this WALL-CLOCK loop after the same number of steps: each rank reading only its own clock,
    one rank could run a chunk more than another -- whose all-reduces then wait forever (the two-rank bench test hung on exactly that, once in seven runs).  The ranks
    therefore agree after every chunk -- a float sum through the same transport as the gradients; any rank past the target ends the loop for all."""
    sync = sync or torch.cuda.synchronize
    done_steps = 0
    t_c = time.perf_counter()
    while True:
        for _ in range(chunk):
            step(done_steps); done_steps += 1
        sync()
        done = (time.perf_counter() - t_c) * 1e3 >= target_ms
        if world > 1:
            flag = torch.tensor([1.0 if done else 0.0], dtype=torch.float32, device=device)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.SUM)
            done = bool(flag.item() > 0.0)
        if done:
            return done_steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=512, help="frames per GPU per step (BASELINE configs[1])")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "bf16x3"])
    ap.add_argument("--pool", type=int, default=2048, help="synthetic frames resident in HBM per GPU")
    ap.add_argument("--frames", default="u8", choices=["u8", "f32"], help="format of the HBM-resident frame pool (bf16 engine: uint8 camera bytes by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ppo", action="store_true")
    ap.add_argument("--no-fp32", action="store_true")
    ap.add_argument("--no-x3", action="store_true")
    ap.add_argument("--no-mlp", action="store_true")
    ap.add_argument("--no-replay", action="store_true")
    ap.add_argument("--no-dp-form", action="store_true", help="skip the N = 1 timing of the data-parallel form of the step (recording communicator)")
    ap.add_argument("--replay-rows", type=int, default=1024)
    ap.add_argument("--condition-ms", type=float, default=300.0, help="untimed steps in FRONT of the counted warm-up until this much wall time has passed: the box reaches its "
                                                                    "sustained clocks / power state before anything is measured (0 = off)")
    ap.add_argument("--no-box", action="store_true", help="skip the in-run box calibration (mi_device_probe)")
    ap.add_argument("--long-steps", type=int, default=200, help="a second, longer timed leg of the same step behind the counted one (`value_200`): shows whether the driver's "
                                                                "20-step sample (16 ms) is representative; 0 = off")
    args = ap.parse_args()

    from mi355 import dist as midist
    # (test hook: MI355_BENCH_BACKEND=gloo MI355_BENCH_ONE_DEVICE=1 runs the N > 1 code path with all ranks on cuda:0 -- RCCL wants one device per rank)
    world, rank, local = midist.init_from_env(os.environ.get("MI355_BENCH_BACKEND", "nccl"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    if os.environ.get("MI355_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    from vae.models import ConvVAE

    tmp = tempfile.mkdtemp(prefix="mi355_bench_")
    B = args.batch
    model = ConvVAE(np.array([80, 160, 3]), z_dim=64, beta=1.0, learning_rate=1e-4, model_dir=os.path.join(tmp, "vae"), precision=args.precision, seed=0)
    model.init_session(init_logging=False)
    dev = model.dev
    dev.ensure_batch(B)
    device = dev.device
    u8_pool = args.frames == "u8" and args.precision == "bf16"

    # synthetic uint8 camera frames, HBM resident (randint(0,256) [/255]), generated on the device
    g = torch.Generator(device=device)
    g.manual_seed(1234 + rank)
    pool_u8 = torch.randint(0, 256, (args.pool, 38400), device=device, generator=g, dtype=torch.int32).to(torch.uint8).contiguous()
    if u8_pool:
        pool = pool_u8
    else:
        pool = torch.empty(args.pool, 38400, device=device)
        dev.L.mi_u8_to_unit_f32(dev.stream(), pool_u8.data_ptr(), pool.data_ptr(), pool.numel())
    total = args.warmup + args.steps
    n_idx = min(total + 2, 64)                            # 64 different minibatches of the pool, cycled
    idx = torch.stack([torch.randperm(args.pool, device=device, generator=g)[:B] for _ in range(n_idx)]).to(torch.int32).contiguous()
    inv_b = 1.0 / (B * world)
    n_ops = dev.L.mi_vae_op_count()
    names = [dev.L.mi_vae_op_name(i).decode() for i in range(n_ops)]
    esz = 2 if args.precision == "bf16" else 4
    frame_bytes = 1 if u8_pool else 4

    def step(i):
        model._train_minibatch(pool, pool, idx[i % n_idx], B, inv_b, None)     # eps=None: drawn inside the reparameterisation kernel

    # ---- clock conditioning (untimed, in front of the counted warm-up): the driver's command is 5 + 20 steps = 25 ms of GPU work from an idle
    # box, i.e. measured on the DVFS ramp; >= --condition-ms of the very same steps first, then the box calibration, then the counted warm-up ----
    conditioned = condition_clocks(step, args.condition_ms, world, device) if args.condition_ms > 0 else 0
    box = None
    if not args.no_box:
        try:
            box = box_probe(dev, local)
        except Exception as e:
            box = {"error": repr(e)}
        for _ in range(20 if args.condition_ms > 0 else 0):      # (the probes ran at their own power point: back to the step's before the warm-up)
            step(conditioned); conditioned += 1
    # ---- warm-up (untimed); two of the warm-up steps run with every op bracketed by HIP events to find the dominant kernel ----
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
    # The profile above runs one op at a time (isolated launch times; the two front-runners -- the decoder tail and deconv3's filter gradient -- are within a microsecond
    # of each other there, so the crown used to change from run to run and `roofline.frac` with it: 0.35 one run, 0.20 the next).  The dominant kernel of the STEP is the
    # one that takes longest IN the step, next to its neighbour on the other backward queue: the four front-runners are timed again inside whole two-queue steps (events
    # around that op only, 6 untimed steps each) and the longest in-step average wins.
    per_op_in_step = {}
    if per_op and args.precision == "bf16" and args.warmup > 0:
        for cand in sorted(per_op, key=per_op.get, reverse=True)[:4]:
            dev.L.mi_vae_timing_begin(dev.handle, 2, names.index(cand), 16)
            for i in range(6):
                step(total + 2 + i)
            torch.cuda.synchronize()
            ms_c, cnt_c = collect_timing(dev, n_ops)
            ci = names.index(cand)
            if cnt_c[ci] > 0:
                per_op_in_step[cand] = float(ms_c[ci] / cnt_c[ci])
        if per_op_in_step:
            dominant = max(per_op_in_step, key=per_op_in_step.get)

    # ---- timed region: exactly K steps, barrier + synchronize on both sides; (eager mode) the dominant op keeps its two HIP events ----
    events_live = dominant is not None
    if events_live:
        dev.L.mi_vae_timing_begin(dev.handle, 2, names.index(dominant), args.steps + 4)
    midist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        step(i)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    midist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, t_local], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t[0].item())
    losses = dev.losses.cpu().numpy()
    roofline = None
    if dominant is not None:
        how = "HIP events on the launch stream around every launch of this op inside the timed region"
        ms_d, cnt_d = collect_timing(dev, n_ops)
        di = names.index(dominant)
        avg_s = float(ms_d[di] / max(cnt_d[di], 1)) * 1e-3
        tail_fused = "deconv4.dgrad" not in per_op and args.precision == "bf16"      # the decoder tail ran as one launch (timed as deconv4.fwd)
        enc_fused = "conv1.wgrad" not in per_op and args.precision == "bf16"         # conv2's input gradient + conv1's filter gradient ran as one launch (timed as conv2.dgrad)
        roofline = roofline_of(dominant, avg_s, B, esz, dev.n_flat, args.precision, frame_bytes, tail_fused, enc_fused)
        if dominant == "conv2.dgrad" and enc_fused:
            roofline["kernel"] = "conv2.dgrad = encoder head of backward (enchead_bwd_kernel: conv2 input gradient + conv1 filter / bias gradient)"
        if dominant == "deconv4.fwd" and tail_fused:
            roofline["kernel"] = "deconv4.fwd = decoder tail (dectail_kernel: deconv4 forward + reconstruction loss + input gradient + filter gradient)"
        roofline["launches_timed"] = int(cnt_d[di])
        if box and "error" not in box:
            # the same achieved rate against what THIS box sustains (mi_device_probe, minutes^-1 ago): separates "the kernel" from "the box" when two runs disagree
            box_peak = box["hbm_read_tbps"] * 1e3 if roofline["bound"] == "hbm" else box["mfma_bf16_tflops"] * (PEAK[{"bf16": "mfma_bf16", "bf16x3": "mfma_bf16x3"}.get(args.precision, "mfma_f32")] / PEAK["mfma_bf16"])
            if box_peak > 0:
                roofline["frac_of_box"] = roofline["achieved"] / box_peak
                roofline["box_peak"] = box_peak
        roofline["timing"] = how
        # HBM traffic of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, gfx950 x2 read correction),
        # measured offline on this same workload and committed under profiles/ (PMC counters cannot be read from inside the bench)
        for fn in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", fn)))
                for op, rec in tj["ops"].items():
                    if dominant in [x.strip().split(" ")[0] for x in op.split("/")]:
                        roofline["traffic"] = rec["hbm_bytes_per_launch"]
                        roofline["traffic_source"] = "profiles/%s: %s" % (fn, tj["provenance"])
                        if rec.get("us"):
                            # the same kernel ALONE on the GPU (the profile's counter pass serialises the launches): in the step it shares the CUs and HBM with
                            # the kernel on the other backward stream, so the event time above is longer than its own (DESIGN finding 25)
                            roofline["alone_in_profile"] = {"launch_ms": rec["us"] * 1e-3, "frac": (roofline["algorithmic_bytes_per_launch"] / (rec["us"] * 1e-6)) / PEAK["hbm"]
                                                            if roofline["bound"] == "hbm" else None}
                if roofline["traffic"] is not None:
                    break
            except Exception:
                pass

    # ---- a longer leg of the very same step, same run (the driver's --steps 20 is a 16 ms sample): barrier + synchronize on both sides, MAX over ranks ----
    long_leg = None
    if args.long_steps > args.steps:
        midist.barrier()
        torch.cuda.synchronize()
        tl0 = time.perf_counter()
        for i in range(args.long_steps):
            step(total + i)
        torch.cuda.synchronize()
        midist.barrier()
        tl = torch.tensor([time.perf_counter() - tl0], device=device, dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(tl, op=torch.distributed.ReduceOp.MAX)
        long_leg = {"steps": args.long_steps, "ms_per_step": float(tl[0].item()) / args.long_steps * 1e3, "frames_per_s": B * world * args.long_steps / float(tl[0].item()),
                    "note": "the same step, %d more steps timed the same way right behind the counted region (not the headline: `value` is the counted K steps)" % args.long_steps}

    # what the DATA-PARALLEL form of this very step costs one rank before any byte travels (round 6): at N = 1 the same C call the N > 1 run issues (mi_vae_train_step_dp: the three
    # backward parts in bucket order, every bucket handed to the communicator, the join, Adam) on a RECORDING communicator -- the driver computes scaling efficiency from the N = 1
    # line, and the part that is schedule, not transport, can be read here
    dp_form = None
    if world == 1 and args.precision == "bf16" and not args.no_dp_form and hasattr(dev, "train_step_dp"):
        try:
            import ctypes
            from vae.models import adam_alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON
            hcomm = ctypes.c_void_p(); rec_log = np.zeros((64, 4), np.int64)
            dev.L.mi_comm_init_recording(ctypes.addressof(hcomm), 0, 1, rec_log.ctypes.data, 64)
            a_ = adam_alpha(1e-4, np.float32(0.9), np.float32(0.999))
            f_dp = lambda i: dev.train_step_dp(hcomm, pool, pool, idx[i % n_idx], B, inv_b, None, a_, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON)
            for i in range(20):
                f_dp(i)
            torch.cuda.synchronize(); td0 = time.perf_counter()
            for i in range(100):
                f_dp(i)
            torch.cuda.synchronize(); t_dp = (time.perf_counter() - td0) / 100 * 1e3
            for i in range(20):
                step(i)
            torch.cuda.synchronize(); td0 = time.perf_counter()
            for i in range(100):
                step(i)
            torch.cuda.synchronize(); t_sg = (time.perf_counter() - td0) / 100 * 1e3
            dev.L.mi_comm_destroy(hcomm)
            dp_form = {"ms_per_step_dp_call_recording_comm": t_dp, "ms_per_step_single_rank_call": t_sg, "ratio": t_dp / t_sg, "steps": 100,
                       "note": "mi_vae_train_step_dp on a recording communicator (no collective is issued) against mi_vae_train_step, 100 steps each, back to back in this run: "
                               "what weak-scaling efficiency loses to the bucket-ordered schedule before transport"}
        except Exception as e:
            dp_form = {"error": repr(e)}

    # data parallel: what one rank spends per step, and how much of it is gradient all-reduce that nothing overlaps
    dp = None
    if world > 1:
        per_rank = [torch.zeros(2, device=device, dtype=torch.float64) for _ in range(world)]
        torch.distributed.all_gather(per_rank, torch.tensor([t_local / args.steps * 1e3, 0.0], device=device, dtype=torch.float64))
        # exposed all-reduce: the same steps with the collectives skipped (gradients then differ per rank: parameters are re-broadcast afterwards)
        os.environ["MI355_DP_SKIP_ALLREDUCE"] = "1"
        torch.cuda.synchronize(); midist.barrier()
        t1 = time.perf_counter()
        for i in range(min(args.steps, 50)):
            step(i)
        torch.cuda.synchronize()
        t_nocomm = (time.perf_counter() - t1) / min(args.steps, 50) * 1e3
        os.environ.pop("MI355_DP_SKIP_ALLREDUCE")
        midist.barrier()
        midist.broadcast(dev.params, 0); dev.sync_shadow()
        dp = {"ms_per_step_by_rank": [float(x[0].item()) for x in per_rank], "ms_per_step_without_allreduce_rank0": t_nocomm,
              "exposed_allreduce_ms_rank0": max(t_local / args.steps * 1e3 - t_nocomm, 0.0), "gradient_bytes_per_step": int(dev.n_flat) * 4, "buckets": 3,
              "transport": midist.comm_note}

    if rank == 0:
        frames_per_s = B * world * args.steps / elapsed
        step_flops = 776494080.0 * B                      # VAE train step, per frame
        out = {
            "metric": METRIC, "value": frames_per_s, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp32": "f32", "bf16x3": "bf16x3 (hi + lo bf16 halves per element)"}[args.precision], "data": "synthetic",
            "config": {"workload": "ConvVAE SGD step (fwd+ELBO+bwd+TF-Adam), batch=%d per GPU, 160x80x3 frames, z_dim=64, rgb target (BASELINE configs[1]; global batch %d)" % (B, B * world),
                       "global_batch": B * world, "parallelism": "dp%d" % world if world > 1 else "single", "frames_resident_in_hbm": args.pool,
                       "frame_table": "uint8 camera bytes, k/255 in registers" if u8_pool else "float32 in [0,1]",
                       "noise": "N(0,1) drawn inside the reparameterisation kernel (Philox4x32-10)",
                       "launch": "eager launches: one C call per step (mi_vae_train_step)",
                       "storage": "bf16 activations/weights, fp32 accumulate, fp32 master weights+Adam" if args.precision == "bf16" else "fp32",
                       "library": _library_stamp()},
            "roofline": roofline,
            "box": box,
            "conditioning": {"untimed_steps_before_warmup": conditioned, "target_ms": args.condition_ms},
            "step_model_flops_utilisation": {"algorithmic_tflops_per_step": step_flops / 1e12,
                                             "achieved_tflops": step_flops * args.steps / elapsed / 1e12,
                                             "frac_of_mfma_peak": step_flops * args.steps / elapsed / PEAK[{"bf16": "mfma_bf16", "bf16x3": "mfma_bf16x3"}.get(args.precision, "mfma_f32")]},
            "per_op_ms_note": "every op bracketed by HIP events, one at a time (two warm-up steps): isolated launch times, not in-step times",
            "per_op_in_step_ms": {k: round(v, 4) for k, v in sorted(per_op_in_step.items(), key=lambda kv: -kv[1])},
            "per_op_in_step_note": "the four longest ops timed again INSIDE whole two-queue steps (events around that op only): the dominant kernel of `roofline` is the longest of these",
            "per_op_ms": {k: round(v, 4) for k, v in sorted(per_op.items(), key=lambda kv: -kv[1])},
            "final_losses": {"reconstruction": float(losses[0]), "kl": float(losses[1])},
            "data_parallel": dp,
            "data_parallel_form_at_one_rank": dp_form,
            "value_200": long_leg,
        }
        if roofline is not None and args.precision == "bf16":
            roofline["step_traffic"] = step_traffic(B)
        out["cpu_baseline"] = None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], ref = cpu_baseline(B)
            try:
                out["parity"] = parity_object(tmp, ref, B)
            except Exception as e:
                out["parity"] = {"error": repr(e)}
        if world == 1 and not args.no_mlp and args.precision == "bf16":
            try:
                out["mlp_vae"] = mlp_extra(tmp, B, pool_u8, idx)
            except Exception as e:
                out["mlp_vae"] = {"error": repr(e)}
        if world == 1 and not args.no_x3 and args.precision == "bf16":
            try:
                out["bf16x3"] = fp32_extra(tmp, B, pool_u8, idx, precision="bf16x3")
            except Exception as e:
                out["bf16x3"] = {"error": repr(e)}
        if world == 1 and not args.no_fp32 and args.precision == "bf16":
            try:
                out["fp32"] = fp32_extra(tmp, B, pool_u8, idx)
            except Exception as e:
                out["fp32"] = {"error": repr(e)}
        if world == 1 and not args.no_ppo:
            try:
                out["ppo"] = ppo_extra(tmp)
            except Exception as e:      # the headline metric must still print
                out["ppo"] = {"error": repr(e)}
        if world == 1 and not args.no_replay:
            try:
                out["replay"] = replay_extra(tmp, args.replay_rows)
            except Exception as e:
                out["replay"] = {"error": repr(e)}
        if box and "error" not in box and box.get("mfma_bf16_tflops", 0) > 0 and args.precision == "bf16":
            out["step_model_flops_utilisation"]["frac_of_box_mfma_rate"] = out["step_model_flops_utilisation"]["achieved_tflops"] / box["mfma_bf16_tflops"]
    if world > 1:
        # both bucket schedules, timed in this run -- behind a watchdog: the headline line exists by now and must get out even if a collective schedule that has never run
        # on this many devices does not come back (rank 0 prints it without the schedule object; every rank leaves)
        wd = _watchdog(180.0, (lambda: print(json.dumps(dict(out, data_parallel=dict(out["data_parallel"] or {}, schedules_note="watchdog: the schedule leg did not return"))), flush=True))
                       if rank == 0 else None)
        extra = dp_schedules(step, min(args.steps, 50), world, rank, device)
        wd.cancel()
        if rank == 0 and out.get("data_parallel") is not None:
            out["data_parallel"].update(extra)
    if rank == 0:
        print(json.dumps(out), flush=True)
    midist.barrier()
    if world > 1:
        midist.shutdown()               # the library's own RCCL communicator first, then the process group it was bootstrapped from
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
