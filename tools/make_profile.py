#!/usr/bin/env python3
"""Assemble profiles/<name>.md + profiles/rNN_pmc_traffic.json from the files one GPU call leaves in gpurun_out/:
    prof_<tag>.md (rocprofv3 --kernel-trace --stats via tools/rocpd_summary.py), pmc_<tag>_sq.md / _fetch.md / _write.md (tools/pmc_pass.sh),
    bench_<tag>_full.json (bench.py), timeline_<tag>.md (tools/timeline.sh; optional).  tools/profile_round.sh produces all of them.   usage: python tools/make_profile.py <tag> <out-name> "<title>" """
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
tag, outname, title = sys.argv[1], sys.argv[2], sys.argv[3]
stats = open(R + "gpurun_out/prof_%s.md" % tag).read()
sq = open(R + "gpurun_out/pmc_%s_sq.md" % tag).read()
fe = open(R + "gpurun_out/pmc_%s_fetch.md" % tag).read()
wr = open(R + "gpurun_out/pmc_%s_write.md" % tag).read()
bench = json.load(open(R + "gpurun_out/bench_%s_full.json" % tag))
tl_path = R + "gpurun_out/timeline_%s.md" % tag
timeline = open(tl_path).read() if os.path.exists(tl_path) else "(not collected)"

def rows(md):
    out = []
    for l in md.splitlines()[2:]:
        c = [x.strip() for x in l.strip("|").split("|")]
        if len(c) > 4:
            out.append(c)
    return out
hdr = [x.strip() for x in sq.splitlines()[0].strip("|").split("|")]
sqr = {(r[0], r[1]): dict(zip(hdr, r)) for r in rows(sq)}
fer = {(r[0], r[1]): float(r[4]) for r in rows(fe)}
wrr = {(r[0], r[1]): float(r[4]) for r in rows(wr)}
# (kernel, grid) -> op at the bench geometry (B = 512, bf16); grids identify the layer
# round 1 dispatch (kept so that profiles/r01_* can be regenerated); round 2: the register-weight kernels, the lean narrow kernels
NAMES_R01 = [("tapconv_kernel<bf16, 1, 128, 3, 256, 96>", "1848x1x1", "deconv3.fwd"), ("tapconv_kernel<bf16, 1, 128, 2, 256, 96>", "1722x1x1", "conv2.dgrad"),
         ("tapconv_kernel<bf16, 1, 128, 2, 128, 48>", "800x2x1", "deconv2.fwd / conv3.dgrad"), ("tapconv_kernel<bf16, 1, 128, 2, 256, 96>", "100x4x1", "deconv1.fwd / conv4.dgrad"),
         ("tapconv_kernel<bf16, 0, 128, 2, 128, 48>", "684x1x1", "conv3.fwd / deconv2.dgrad"), ("tapconv_kernel<bf16, 0, 64, 2, 256, 96>", "1482x1x1", "conv2.fwd"),
         ("gemm2_kernel<bf16, 0, 1, 128, 64, false>", "2736x1x1", "deconv3.dgrad"), ("gemm2_kernel<bf16, 0, 1, 128, 64, false>", "96x4x1", "conv4.fwd / deconv1.dgrad"),
         ("tapwgrad_kernel<1, 3, 2, 4, 4, false>", "247x1x1", "deconv3.wgrad"), ("tapwgrad_kernel<0, 2, 4, 2, 2, true>", "247x1x1", "conv2.wgrad"),
         ("tapwgrad_kernel<0, 2, 4, 2, 2, true>", "63x4x1", "conv3.wgrad"), ("tapwgrad_kernel<0, 2, 4, 2, 2, true>", "16x16x1", "conv4.wgrad"),
         ("narrow_wgrad_kernel<float, 3, 1, 12>", None, "conv1.wgrad (+bias)"), ("narrow_wgrad_kernel<bf16, 3, 0, 12>", None, "deconv4.wgrad"),
         ("narrow_conv_kernel<bf16, float>", None, "conv1.fwd"), ("narrow_conv_kernel<bf16, bf16>", None, "deconv4.dgrad"),
         ("gather_narrow_kernel<bf16, 2, 4, 3>", None, "deconv4.fwd + loss")]
NAMES_R02 = [("rwconv_gather_kernel<3, 5, true, 0, true, 0>", None, "deconv3.fwd"), ("rwconv_gather_kernel<2, 4, false, 2, false, 0>", None, "conv2.dgrad"),
         ("rwconv_gather_kernel<3, 5, true, 0, true, 0, 1>", None, "deconv3.fwd"), ("rwconv_gather_kernel<2, 4, false, 2, false, 0, 1>", None, "conv2.dgrad"),
         ("rwconv_gather_kernel<2, 4, true, 0, false, 0, 2>", None, "deconv2.fwd"), ("rwconv_gather_kernel<2, 4, false, 1, false, 0, 2>", None, "conv3.dgrad"),
         ("rwconv_conv_kernel<5, false, true>", None, "deconv3.dgrad"), ("rwconv_conv_kernel<4, true, false>", None, "conv2.fwd"),
         ("rwconv_conv_kernel<5, 1, false, true>", None, "deconv3.dgrad"), ("rwconv_conv_kernel<4, 1, true, false>", None, "conv2.fwd"),
         ("rwconv_conv_kernel<4, 2, true, false>", None, "conv3.fwd"), ("rwconv_conv_kernel<4, 2, false, true>", None, "deconv2.dgrad"),
         ("tapconv_kernel<bf16, 1, 128, 2, 128, 48>", "800x2x1", "deconv2.fwd / conv3.dgrad"), ("tapconv_kernel<bf16, 1, 128, 2, 256, 96>", "100x4x1", "deconv1.fwd / conv4.dgrad"),
         ("tapconv_kernel<bf16, 0, 128, 2, 128, 48>", "684x1x1", "conv3.fwd / deconv2.dgrad"), ("tapconv_kernel<bf16, 0, 64, 2, 256, 96>", "1482x1x1", "conv2.fwd"),
         ("gemm2_kernel<bf16, 0, 1, 128, 64, false>", "2736x1x1", "deconv3.dgrad"), ("gemm2_kernel<bf16, 0, 1, 128, 64, false>", "96x4x1", "conv4.fwd / deconv1.dgrad"),
         ("tapwgrad_kernel<1, 3, 2, 4, 4, false>", "247x1x1", "deconv3.wgrad"), ("tapwgrad_kernel<0, 2, 4, 2, 2, true>", "247x1x1", "conv2.wgrad"),
         ("tapwgrad_kernel<0, 2, 4, 2, 2, true>", "63x4x1", "conv3.wgrad"), ("tapwgrad_kernel<0, 2, 4, 2, 2, true>", "16x16x1", "conv4.wgrad"),
         ("tapwgrad_kernel<1, 2, 4, 2, 2, true>", "62x4x1", "deconv2.wgrad"), ("tapwgrad_kernel<1, 2, 4, 2, 2, true>", "16x16x1", "deconv1.wgrad"),
         # after the 1-D XCD-grouped launch (and the class / tap-row kernel for k = 5): conv3 / conv4 and deconv2 / deconv1 share kernel AND grid
         ("tapwgrad_cw_kernel", None, "deconv3.wgrad"), ("tapwgrad_kernel<0, 2, 4, 2, 2, true>", "248x1x1", "conv2.wgrad"),
         ("tapwgrad_kernel<0, 2, 4, 2, 2, true>", "256x1x1", "conv3.wgrad + conv4.wgrad (same kernel and grid: averaged)"),
         ("tapwgrad_kernel<1, 2, 4, 2, 2, true>", "256x1x1", "deconv2.wgrad + deconv1.wgrad (same kernel and grid: averaged)"),
         ("narrow_wgrad_kernel<unsigned char, 3, 1, 12>", None, "conv1.wgrad (+bias)"), ("narrow_wgrad_kernel<bf16, 3, 0, 12>", None, "deconv4.wgrad"),
         ("narrow_conv48_kernel<unsigned char, 0>", None, "conv1.fwd"), ("narrow_conv48_kernel<bf16, 1>", None, "deconv4.dgrad"),
         ("gather_narrow_kernel<bf16, 2, 4, 3, true>", None, "deconv4.fwd + loss"), ("reduce_fused_kernel", None, "filter-gradient slab reduce (6 layers)"),
         ("adam_tf_kernel", None, "adam")]
# round 3: the decoder tail is one kernel (deconv4 forward + loss + both of its gradients); deconv3's forward no longer writes ReLU bit words
NAMES_R03 = [("dectail_kernel<true>", None, "deconv4.fwd (decoder tail: deconv4 fwd + loss + dgrad + wgrad)"), ("dectail_kernel<false>", None, "deconv4.fwd (decoder tail, library-math loss)"),
             ("rwconv_gather_kernel<3, 5, true, 0, false, 0, 1>", None, "deconv3.fwd"), ("dectail_reduce_kernel", None, "decoder-tail slab reduce"),
             ("enchead_bwd_kernel<unsigned char>", None, "conv2.dgrad (encoder head of backward: conv2 dgrad + conv1 wgrad + bias)"), ("enchead_reduce_kernel", None, "encoder-head slab reduce"),
             ("gemm2_kernel<bf16, 0, 1, 128, 64, false, 2>", "96x4x1", "conv4.fwd / deconv1.dgrad")] + NAMES_R02
# round 4: the small-grid layers and two mid layers on the activation-resident kernels (the general kernels they replace still show up once in a profile whose first
# step ran before the fragment-ordered weight copies existed: not listed)
R04_GONE = ("conv4.fwd / deconv1.dgrad", "deconv1.fwd / conv4.dgrad", "deconv2.fwd", "conv3.dgrad", "deconv2.fwd / conv3.dgrad")
NAMES_R04 = [("ares_conv_kernel<4, 1>", None, "conv4.fwd / deconv1.dgrad"), ("ares_gather_kernel<4, 2>", None, "deconv1.fwd / conv4.dgrad"),
             ("ares_gather2_kernel", None, "deconv2.fwd / conv3.dgrad"), ("reduce_small_fused_kernel", "1365x1x1", "end-of-pass slab sums (one launch)"),
             ("adam_tf_layouts_kernel<bf16>", None, "adam (writes both weight layouts)")] \
    + [n for n in NAMES_R03 if n[2] not in R04_GONE]
# round 5: the latent layers' filter gradients on the LDS-free kernel (both launches have 768 waves: one row, averaged)
# round 5: the slab sums of the filter gradients are two launches (the decoder's three layers behind deconv1's filter gradient, the encoder's three at the end of the pass);
# the latent layers' filter gradients stay on the first-generation kernel (the LDS-free dwgs_kernel serves the MlpVAE's small layers only: DESIGN 3.15)
NAMES_R05 = [("reduce_fused_kernel", "1864x1x1", "slab reduce, decoder filter gradients (mid-pass, filter-gradient queue's idle gap)"),
             ("reduce_fused_kernel", "1568x1x1", "slab reduce, encoder filter gradients (end of pass)"),
             ("wgrad_kernel<bf16, bf16, 8, 16, 128>", "1x96x2", "dense1.wgrad (+ bias row)"), ("wgrad_kernel<bf16, bf16, 8, 16, 128>", "49x2x2", "heads.wgrad (+ bias row)"),
             ("wgrad_pair_kernel<bf16, bf16, 8, 16, 128>", "388x1x1", "dense1.wgrad + heads.wgrad (+ bias rows): ONE launch in the step (the two rows above: per-op timing keeps them apart)"),
             # the encoder head of the forward pass is one kernel (conv1's activation never leaves LDS on its way into conv2); the two kernels it replaces
             # still run twice in a profile (bench.py's isolated per-op table): not listed
             ("enc12_fwd_kernel<unsigned char, 0, 1, 1>", None, "conv1.fwd / conv2.fwd (encoder head of forward: one kernel)")] \
    + [n for n in NAMES_R04 if n[0] != "reduce_fused_kernel" and n[2] not in ("conv1.fwd", "conv2.fwd")]
# round 6: the raw-staged filter gradients decode a step's DMA rows once per wave (template flag LDEC: new kernel names)
NAMES_R06 = NAMES_R05                                     # (same ops; the kernels' template signatures grew: matched by prefix below)
NAMES = NAMES_R06 if tag.startswith("r06") else NAMES_R05 if tag.startswith("r05") else NAMES_R01 if tag.startswith("r01") else (NAMES_R02 if tag.startswith("r02") else (NAMES_R03 if tag.startswith("r03") else NAMES_R04))
lines, traffic, seen_ops = [], {}, set()
step_bytes, step_kernels = 0.0, 0
for kern, grid, op in NAMES:
    def same(name):                                       # later rounds append template flags (LDEC, DBG, ...) behind the round-5 signature: `k<a, b>` also matches `k<a, b, false>`; `k` matches `k<false>`
        name = name.strip("`")
        return name == kern or (kern.endswith(">") and name.startswith(kern[:-1] + ", ")) or (not kern.endswith(">") and name.startswith(kern + "<"))
    keys = [k for k in sqr if same(k[0]) and (grid is None or k[1] == grid)]
    if not keys or op in seen_ops:
        continue
    seen_ops.add(op)
    key = keys[0]
    d = sqr[key]; us = float(d["us"]); gui = float(d["GRBM_GUI_ACTIVE"]) / 8.0
    util = float(d["VALU_MFMA_BUSY_CYCLES"]) / (gui * 1024.0) if gui > 0 else 0.0
    f, w = fer.get(key), wrr.get(key)
    fmb = 2 * f / 1e3 if f else 0.0; wmb = w / 1e3 if w else 0.0
    nmf, nva = float(d.get("INSTS_MFMA", 0) or 0), float(d.get("INSTS_VALU", 0) or 0)
    vpm = nva / nmf if nmf > 0 else float("nan")
    kern = key[0].strip("`")
    lines.append("| %s | `%s` %s | %.1f | %.1f%% | %.1f | %.1f | %.1f | %.2f |" % (op, kern, key[1], us, 100 * util, vpm, fmb, wmb, (fmb + wmb) / us))
    traffic[op] = {"kernel": kern, "grid": key[1], "us": us, "fetch_mb_corrected": fmb, "write_mb": wmb, "hbm_bytes_per_launch": (fmb + wmb) * 1e6, "mfma_util": util,
                   "valu_per_mfma": None if nmf <= 0 else vpm}
    mult = 2 if "averaged" in op else 1                   # (two layers of a step share kernel and grid: one averaged row, two launches per step)
    if "ONE launch in the step" in op: mult = 0           # (its two problems are the two single-launch rows of the per-op timing pass: counted there)
    step_bytes += mult * (fmb + wmb) * 1e6; step_kernels += mult
doc = """# %s

ConvVAE bf16 SGD step, batch 512 (BASELINE configs[1]), 1x MI355X.  ONE `gpurun` call = one box, one library (stamp and the box's own calibration: `config.library` / `box` in the bench line below).  Sources: `rocprofv3 --kernel-trace --stats` of
`python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ppo --no-fp32 --no-mlp --no-replay --no-x3 --no-dp-form` (rocpd database summarised by `tools/rocpd_summary.py`) and three separate
`rocprofv3 --pmc` passes (`tools/pmc_pass.sh`: SQ/GRBM counters, FETCH_SIZE alone, WRITE_SIZE alone -- one TCC-derived counter per pass, no trace
domains combined with --pmc).  Assembled by `tools/make_profile.py`.

## bench.py line of the same commit (un-profiled run, with CPU baseline and PPO extra)
```json
%s
```

## Per-kernel derived metrics (B = 512)

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).  HBM bytes = 2 x FETCH_SIZE (the gfx950 correction of
MI355X_MICROARCH.md: wide coalesced reads are tallied at half their size; calibrated for 16-byte-per-lane reads, the 8-byte fp32 frame reads of the
conv1 kernels of round 1 may be over-counted) + WRITE_SIZE, both reported in KB by rocprofv3.  Last column: (read + write) / duration; 8 TB/s spec, ~6.3 achievable.

| op | kernel, grid | us (alone: the counter pass serialises) | MFMA util | VALU / MFMA instructions | HBM read MB | HBM write MB | HBM TB/s |
|---|---|---:|---:|---:|---:|---:|---:|
%s

Whole step (every row above once, the two averaged rows twice; the small latent-layer / reparameterisation / loss kernels are not in the table): **%.2f GB of HBM traffic in %d kernels**
against SURVEY 8(d)'s 1.54 GB "practical un-fused" and 0.141 GB algorithmic.

## Kernel time table (every step of the profiled run: clock conditioning + warm-up + timed steps; the averages are IN-STEP times)

%s

## One training step, dispatch by dispatch (separate `rocprofv3 --kernel-trace` run; queue 1 = caller's stream, the other = filter-gradient stream)

%s

## PMC pass 1 (SQ / GRBM), averaged per dispatch

%s

## PMC pass 2 (FETCH_SIZE, KB, uncorrected)

%s

## PMC pass 3 (WRITE_SIZE, KB)

%s
""" % (title, json.dumps(bench), "\n".join(lines), step_bytes / 1e9, step_kernels, stats, timeline, "\n".join(sq.splitlines()[:32]), "\n".join(fe.splitlines()[:26]), "\n".join(wr.splitlines()[:26]))
open(R + "profiles/%s.md" % outname, "w").write(doc)
json.dump({"provenance": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 3, batch 512 bf16; FETCH_SIZE doubled per MI355X_MICROARCH.md",
           "ops": traffic, "step": {"hbm_bytes": step_bytes, "kernels": step_kernels, "batch": 512}}, open(R + "profiles/%s_pmc_traffic.json" % tag[:3], "w"), indent=1)
print("\n".join(lines))
