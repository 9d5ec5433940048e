"""Model-level parity: the drop-in ConvVAE (HIP path, through the C ABI) vs the CPU oracle on identical seeded inputs.

Tolerances (stated per north_star): fp32 mode — losses / outputs / gradients within 1e-4 relative of the oracle;
bf16 mode — compared with the oracle's bf16-storage emulation (same rounding points, fp32 accumulate): losses 2e-3,
gradients 3e-2 of each tensor's max (bf16 has 8 mantissa bits; deviations are reported, not hidden).
Index work (minibatch permutations) is the reference's own legacy-numpy shuffle, hence bit-exact by construction;
tested in test_config1_epoch_* through identical epoch metrics."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vae_oracle as vo  # noqa: E402
from vae.models import ConvVAE, MlpVAE, bce_loss, bce_loss_v2, mse_loss  # noqa: E402


def synth_frames(n, seed=1234):
    return np.random.RandomState(seed).randint(0, 256, (n, 80, 160, 3), dtype=np.uint8).astype(np.float32) / 255.0


def make(tmp_path, precision, target_c=3, params=None, **kw):
    m = ConvVAE(np.array([80, 160, 3]), np.array([80, 160, target_c]), z_dim=64, model_dir=str(tmp_path), precision=precision, **kw)
    if params is not None:
        m.set_weights(params)
    m.init_session(init_logging=False)
    return m


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def trained_like_params(seed=0, target_c=3):
    """Glorot weights + small random biases so that ReLU masks / biases carry signal on every path."""
    p = vo.init_vae_params(seed, 64, (80, 160, 3), (80, 160, target_c))
    rng = np.random.RandomState(seed + 1)
    for k in p:
        if k.endswith("bias"):
            p[k] = (0.05 * rng.standard_normal(p[k].shape)).astype(np.float32)
    return p


# bf16x3 (split storage): losses and outputs at the fp32 limits (1e-4); every single op is within 2e-5 of float64 on its own inputs (test_ops_gpu).
# Gradients of the WHOLE graph: the tensors behind the 64-d bottleneck (dense1 / deconv1 / deconv2) are ill-conditioned -- many pre-activations sit near
# the ReLU threshold, and the deviation there is proportional to the operand precision: the fp32 engine measures ~6e-5 of the tensor max on dense1 at
# B = 6, the split engine (2^-17 instead of 2^-24 per operand, ~1e-5 per op) 6.2e-3, deconv1 4.3e-3, deconv2 1.7e-3, every other tensor <= 2e-4; at
# B = 512 (test_b512) 2-4 x the fp32 ORACLE's own distance from float64.  Limit 1e-2, against 3e-2 for bf16 storage.
@pytest.mark.parametrize("precision,storage,tol_loss,tol_grad", [("fp32", "fp32", 1e-4, 1e-4), ("bf16x3", "fp32", 1e-4, 1e-2), ("bf16", "bf16", 2e-3, 3e-2)])
def test_train_step_losses_grads_and_adam(tmp_path, precision, storage, tol_loss, tol_grad):
    params = trained_like_params()
    B = 6
    frames = synth_frames(B)
    eps = np.random.RandomState(4321).standard_normal((B, 64)).astype(np.float32)
    (recon, kl, _), grads, fw = vo.vae_loss_and_grads(params, frames, frames, eps, beta=1.0, storage=storage)
    m = make(tmp_path, precision, params=params)
    src = m._frames(frames, 38400, "src")
    e = m._eps(B, eps)
    m.dev.forward(src, src, None, B, 1.0 / B, e, 1, 1)
    got = m.dev.losses.cpu().numpy()
    assert abs(got[0] / recon - 1) < tol_loss and abs(got[1] / kl - 1) < max(tol_loss, 2e-3 if precision == "bf16" else 0), (got, recon, kl)
    mean = m.dev._view(1, B * 64).cpu().numpy().reshape(B, 64)
    assert rel_err(mean, fw["mean"].numpy()) < (1e-4 if precision != "bf16" else 2e-2)
    m.dev.backward(src, None, e, 1.0 / B, 0)
    g = m.dev.export_grads()
    if precision != "bf16":
        worst = {k: rel_err(g[k], grads[k]) for k in grads}
        bad = {k: v for k, v in worst.items() if v > tol_grad}
    else:
        # bf16 storage: ReLU masks of pre-activations within one bf16 ulp of zero flip with the fp32 summation order, so two
        # correct kernels differ by whole gradient entries.  Accuracy statement that does not depend on the order: the device
        # gradients are as close to the exact fp32 gradients as the oracle's own bf16-storage emulation is (within 1.25 x + 0.2 % of the tensor max).
        _, exact, _ = vo.vae_loss_and_grads(params, frames, frames, eps, beta=1.0, storage="fp32")
        bad = {}
        for k in grads:
            e_dev, e_emul = rel_err(g[k], exact[k]), rel_err(grads[k], exact[k])
            if e_dev > 1.25 * e_emul + 2e-3:
                bad[k] = (e_dev, e_emul)
    assert not bad, bad
    # three full SGD steps: parameters track the oracle's TF-Adam trajectory
    o = vo.OracleVAE(params=params, storage=storage)
    m2 = make(tmp_path, precision, params=params)
    for s in range(3):
        ee = np.random.RandomState(100 + s).standard_normal((B, 64)).astype(np.float32)
        ro, ko = o.train_step(frames, frames, ee)
        rg, kg = m2.train_step(frames, frames, eps=ee)
        assert abs(rg / ro - 1) < tol_loss * (1 if precision != "bf16" else 3), (s, rg, ro)
    got_p = m2.dev.export_params()
    # Adam's first steps move every weight by ~lr regardless of gradient scale (sign-like updates): a last-bit gradient difference near zero moves a
    # weight by up to 2 lr, in the oracle as much as on the device.  Both are therefore measured against the SAME three steps run in float64
    # (forward, gradients and the Adam recurrence): per tensor, the RMS error of the device's update must not exceed twice the fp32 oracle's
    # own (floor: 0.2 % of lr per step -- bf16x3: 15 %, bf16 storage: 25 %).
    from collections import OrderedDict
    ex = OrderedDict((k, v.astype(np.float64)) for k, v in params.items())
    adam64 = vo.AdamTF(OrderedDict((k, v.shape) for k, v in ex.items()), dtype=np.float64)
    for s in range(3):
        ee = np.random.RandomState(100 + s).standard_normal((B, 64)).astype(np.float32)
        _, g64, _ = vo.vae_loss_and_grads(ex, frames, frames, ee, beta=1.0, dtype=torch.float64)
        adam64.step(ex, g64, 1e-4)
    # (measured: fp32 ~1e-5 like the oracle itself; bf16x3 2.4e-2 on conv1's kernel and 0.11 on conv1's 32-element bias: two sign flips of ~zero gradients)
    floor = {"fp32": 0.002, "bf16x3": 0.15, "bf16": 0.25}[precision] * 3e-4
    rows = []
    for k, v in o.params.items():
        upd_x = ex[k] - params[k].astype(np.float64)
        e_o = np.sqrt(np.mean(((v.astype(np.float64) - params[k]) - upd_x) ** 2))
        e_d = np.sqrt(np.mean(((got_p[k].astype(np.float64) - params[k]) - upd_x) ** 2))
        rows.append((k, e_d / 3e-4, e_o / 3e-4))
        assert e_d <= 2.0 * e_o + floor, (k, e_d / 3e-4, e_o / 3e-4)
    print("\n3 Adam steps (%s), RMS update error / (3 lr) vs the float64 trajectory (device, oracle):" % precision)
    for r in rows:
        print("  %-38s %.3e  %.3e" % r)
    assert m2.beta1_power == pytest.approx(0.9 ** 4, rel=1e-6)


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("bf16x3", 1e-4), ("bf16", 3e-2)])
def test_encode_reconstruct_generate(tmp_path, precision, tol):
    params = trained_like_params(3)
    frames = synth_frames(5, seed=7)
    o = vo.OracleVAE(params=params, training=False, storage="fp32" if precision != "bf16" else "bf16")
    m = make(tmp_path, precision, params=params, training=False)
    assert rel_err(m.encode(frames), o.encode(frames)) < tol
    assert m.encode([frames[0]])[0].shape == (64,)                       # vae_common.py:48 call pattern
    rec, rec_o = m.reconstruct(frames), o.reconstruct(frames)
    assert len(rec) == 5 and rec[0].shape == (80, 160, 3)
    assert np.abs(np.stack(rec) - np.stack(rec_o)).max() < (2e-5 if precision != "bf16" else 2e-2)
    z = np.random.RandomState(0).standard_normal((3, 64)).astype(np.float32)
    g, g_o = m.generate_from_latent(z), o.generate_from_latent(z)
    assert g.shape == (3, 38400) and np.abs(g - g_o).max() < (2e-5 if precision != "bf16" else 2e-2)
    assert np.array_equal(m.decode(z), g)
    with pytest.raises(ValueError):
        m.encode(frames * 1.5)                                           # verify_range


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("variant", ["seg", "kl_tol", "bce_v2", "mse"])
def test_variants_fp32(tmp_path, variant, precision):
    tc = 1 if variant == "seg" else 3
    params = trained_like_params(5, tc)
    B = 4
    frames = synth_frames(B, seed=11)
    tgt = frames if tc == 3 else (np.random.RandomState(2).randint(0, 13, (B, 80, 160, 1)).astype(np.float32) / 12.0)
    eps = np.random.RandomState(9).standard_normal((B, 64)).astype(np.float32)
    kw = dict(beta=1.0, kl_tolerance=0.5 if variant == "kl_tol" else 0.0, loss_fn={"bce_v2": "bce_v2", "mse": "mse"}.get(variant, "bce"))
    (recon, kl, _), grads, _ = vo.vae_loss_and_grads(params, frames, tgt, eps, **kw)
    m = make(tmp_path, precision, target_c=tc, params=params, kl_tolerance=kw["kl_tolerance"],
             loss_fn={"bce": bce_loss, "bce_v2": bce_loss_v2, "mse": mse_loss}[kw["loss_fn"]])
    src = m._frames(frames, 38400, "src")
    tg = src if tc == 3 else m._frames(tgt, 12800, "tgt")
    e = m._eps(B, eps)
    m.dev.forward(src, tg, None, B, 1.0 / B, e, 1, 1)
    got = m.dev.losses.cpu().numpy()
    assert abs(got[0] / recon - 1) < 1e-4 and abs(got[1] / kl - 1) < 1e-4, (got, recon, kl)
    m.dev.backward(src, None, e, 1.0 / B, 0)
    g = m.dev.export_grads()
    bad = {k: rel_err(g[k], grads[k]) for k in grads if rel_err(g[k], grads[k]) > 2e-4}
    assert not bad, bad


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_config1_epoch_evaluate_then_train_fp32(tmp_path, precision):
    """BASELINE config 1: 1k synthetic frames, val = first 10 %, batch 32: evaluate() then train_one_epoch() (28 steps),
    same legacy-numpy permutations and injected noise in oracle and HIP path -> identical epoch metrics (1e-4)."""
    N, bs = 1000, 32
    frames = synth_frames(N)
    val, train = frames[:100], frames[100:]
    params = vo.init_vae_params(0)
    steps_v, steps_t = len(val) // bs, len(train) // bs
    eps_rng = np.random.RandomState(4321)
    eps_v = eps_rng.standard_normal((steps_v, bs, 64)).astype(np.float32)
    eps_t = eps_rng.standard_normal((steps_t, bs, 64)).astype(np.float32)
    o = vo.OracleVAE(params=params)
    np.random.seed(0)
    it = iter(eps_v)
    ov = o.evaluate(val, val, bs, lambda n: next(it))
    it = iter(eps_t)
    ot = o.train_one_epoch(train, train, bs, lambda n: next(it))
    m = make(tmp_path, precision, params=params)
    np.random.seed(0)
    gv = m.evaluate(val, val, bs, eps=eps_v)
    m.train_one_epoch(train, train, bs, eps=eps_t)
    gt = m.last_train_metrics
    assert abs(gv[0] / ov[0] - 1) < 1e-4 and abs(gv[1] / ov[1] - 1) < 1e-4, (gv, ov)
    # KL at initialisation is ~7e-3 and is the sum of 64 cancelling fp32 terms of magnitude ~1 (1 + lv - mu^2 - e^lv, the
    # reference's own formula, vae/models.py:7-9): its fp32 rounding floor is 64 * 2^-24 ~ 4e-6 absolute, and 28 early-Adam
    # steps (sign-like updates) amplify last-bit gradient differences.  1e-4 relative OR that absolute floor.
    # (bf16x3: the same floor scaled to its operand precision -- measured 8e-6 absolute on a KL of 6.7e-3 after the 28 steps; reconstruction loss 4e-8)
    kl_floor = 4e-6 if precision == "fp32" else 2e-5
    assert abs(gt[0] / ot[0] - 1) < 1e-4 and (abs(gt[1] / ot[1] - 1) < 1e-4 or abs(gt[1] - ot[1]) < kl_floor), (gt, ot)
    assert m.get_step_idx() == 1 and o.step_idx == 1
    # the epoch really trained: reconstruction loss dropped well below the untrained 38400*ln2
    assert gt[0] < gv[0]


@pytest.mark.parametrize("fmt", ["npz", "tf"])
def test_checkpoint_roundtrip_and_tf_names(tmp_path, golden_dir, fmt, monkeypatch):
    """fmt = tf: the files are the reference's own format (tf.train.Saver bundle, mi355/tf_bundle.py): same variable names, shapes AND
    dtypes as the reference's shipped model.ckpt-232.index, restored through the same load_latest_checkpoint()."""
    monkeypatch.setenv("MI355_CKPT_FORMAT", fmt)
    ref = json.load(open(os.path.join(golden_dir, "ref_variables.json")))["vae_rgb"]
    m = make(tmp_path / "a", "fp32", params=trained_like_params(1))
    frames = synth_frames(4)
    eps = np.zeros((4, 64), np.float32)
    m.train_step(frames, frames, eps=eps)
    m.step_idx = 7
    sd = m.state_dict()
    assert {k: list(np.shape(v)) for k, v in sd.items()} == {k: v["shape"] for k, v in ref.items()}   # every TF global variable
    m.save()
    assert os.path.exists(os.path.join(m.checkpoint_dir, "checkpoint"))
    if fmt == "tf":
        from mi355 import tf_bundle as tb
        assert sorted(os.listdir(m.checkpoint_dir)) == ["checkpoint", "model.ckpt-7.data-00000-of-00001", "model.ckpt-7.index"]
        ours, _ = tb.read_index(os.path.join(m.checkpoint_dir, "model.ckpt-7.index"))
        theirs, _ = tb.read_index(os.path.join(golden_dir, "ref_index", "vae_rgb.index"))
        assert {k: (e["dtype"], e["shape"], e["size"]) for k, e in ours.items()} == {k: (e["dtype"], e["shape"], e["size"]) for k, e in theirs.items()}
    m2 = ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=str(tmp_path / "a"), precision="fp32")
    m2.init_session(init_logging=False)
    assert m2.load_latest_checkpoint() is True and m2.get_step_idx() == 7
    r1, r2 = m.train_step(frames, frames, eps=eps), m2.train_step(frames, frames, eps=eps)
    assert r1 == pytest.approx(r2, rel=1e-6)
    m3 = ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=str(tmp_path / "empty"), precision="fp32")
    m3.init_session(init_logging=False)
    assert m3.load_latest_checkpoint() is None


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16"])
def test_full_batch_properties_b512(tmp_path, precision):
    """Size-independent properties at BASELINE's full per-GPU batch (512): zero weights => logits 0 => recon = P*ln2,
    KL = 0, dlogits = (0.5 - y)/B exactly representable sums; one Adam step moves every deconv4 bias by exactly lr."""
    zero = {k: np.zeros(s, np.float32) for k, s in vo.vae_variable_specs(64).items()}
    m = make(tmp_path, precision, params=zero)
    B = 512
    frames = synth_frames(B, seed=3)
    recon, kl = m.train_step(frames, frames, eps=np.zeros((B, 64), np.float32))
    assert recon == pytest.approx(38400 * np.log(2.0), rel=2e-6) and abs(kl) < 1e-6
    p = m.dev.export_params()
    db = p["vae/decoder/deconv4/bias"]
    assert np.allclose(np.abs(db), 1e-4, rtol=1e-3) and np.isfinite(np.concatenate([v.ravel() for v in p.values()])).all()
    # gradient wrt deconv4 bias = mean_b sum_pix (0.5 - y): sign of the update is its negative
    gsign = np.sign((0.5 - frames.reshape(-1, 3)).sum(0))
    assert np.array_equal(np.sign(db), -gsign)


def _dev_table(title, rows):
    """Printed (pytest -s / captured on failure, and written next to the test run): the measured deviations are part of the parity statement."""
    print("\n" + title)
    for k, v in rows:
        print("  %-38s %s" % (k, v))


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16"])
def test_b512_train_step_against_the_oracle(tmp_path, precision):
    """BASELINE configs[1] AT ITS BENCHMARKED SIZE (batch 512): one full SGD step -- forward losses, posterior mean, all 22 gradient tensors,
    TF-Adam update -- of the HIP path against the CPU oracle on the same seeded inputs (not against another HIP engine).

    fp32 mode (the 1e-4 parity mode): losses 1e-4 relative, encode() output 1e-4 of its max; gradients: every tensor within 2e-4 of its max of
    the EXACT (float64 oracle) gradient, or -- for the tensors whose fp32 evaluation is itself ill-conditioned at this batch size -- no
    further from it than twice the fp32 oracle is.  (Measured: dense1 / deconv1 / deconv2 kernel gradients of the reference-style fp32 CPU
    computation differ from the float64 result by 1.9e-3 / 2.7e-3 / 4e-4 of the tensor maximum, and by as much from THEMSELVES when torch
    merely sums in a different thread order: ReLU-mask flips of pre-activations within an ulp of zero right behind the 64-d bottleneck.
    No fp32 implementation, TensorFlow's included, can agree with another to 1e-4 on those three tensors; all others agree to < 2e-4.)
    Parameter update after Adam within 2 % of lr on >= 99.5 % of the weights (the first Adam step is lr * g / (|g| + 1e-8): sign-like, so
    last-bit gradient differences flip a handful of |g| ~ 1e-8 entries).
    bf16 mode (the throughput mode, bf16 storage + fp32 accumulate): compared with the oracle's bf16-STORAGE emulation (same rounding points);
    the measured deviations are PRINTED and bounded: reconstruction loss 2e-3, KL 2e-2 (a 64-term cancelling sum of ~1e-2 magnitude at
    initialisation), posterior mean 3e-2 of its max, each gradient no further from the exact-fp32 gradient than 2x the emulation's own
    distance x 1.25 + 2e-3 of the tensor max."""
    B = 512
    params = trained_like_params()
    frames = synth_frames(B)
    eps = np.random.RandomState(4321).standard_normal((B, 64)).astype(np.float32)
    storage = "fp32" if precision != "bf16" else "bf16"
    (recon, kl, _), grads, fw = vo.vae_loss_and_grads(params, frames, frames, eps, beta=1.0, storage=storage)
    m = make(tmp_path, precision, params=params)
    src = m._frames(frames, 38400, "src")
    e = m._eps(B, eps)
    m.dev.forward(src, src, None, B, 1.0 / B, e, 1, 1)
    got = m.dev.losses.cpu().numpy().copy()
    mean = m.dev._view(1, B * 64).cpu().numpy().reshape(B, 64).copy()
    m.dev.backward(src, None, e, 1.0 / B, 0)
    g = m.dev.export_grads()
    d_recon, d_kl, d_mean = abs(got[0] / recon - 1), abs(got[1] / kl - 1), rel_err(mean, fw["mean"].numpy())
    worst = {k: rel_err(g[k], grads[k]) for k in grads}
    rows = [("reconstruction loss rel", "%.3e" % d_recon), ("kl loss rel", "%.3e" % d_kl), ("posterior mean / max", "%.3e" % d_mean)]
    if precision != "bf16":
        # bf16x3 (split storage, ~2^-17 per operand): the same statement as fp32 for losses and outputs (1e-4, measured 6e-8 / 6e-7 / < 1e-4); gradients:
        # floor 1e-3 of the tensor max, or 4 x (instead of 2 x) the fp32 oracle's own distance from float64 on the ReLU-flip-sensitive tensors behind
        # the bottleneck (measured: dense1 3.6e-3 vs the oracle's 1.6e-3, deconv1 9.3e-3 vs 2.7e-3, deconv2 7.1e-4 vs 4.0e-4; all others <= 2.1e-4)
        floor, factor = (2e-4, 2.0) if precision == "fp32" else (1e-3, 4.0)
        _, exact, _ = vo.vae_loss_and_grads(params, frames, frames, eps, beta=1.0, dtype=torch.float64)
        bad = {}
        for k in grads:
            e_dev, e_o32 = rel_err(g[k], exact[k]), rel_err(grads[k], exact[k])
            rows.append(("grad " + k, "dev-vs-exact %.3e  fp32-oracle-vs-exact %.3e  dev-vs-fp32-oracle %.3e" % (e_dev, e_o32, worst[k])))
            if e_dev > max(floor, factor * e_o32):
                bad[k] = (e_dev, e_o32)
        _dev_table("B=512 %s HIP path vs the oracle (limits 1e-4 / 1e-4 / 1e-4 / max(%.0e, %.0f x the fp32 oracle's own distance from float64)):" % (precision, floor, factor), rows)
        assert d_recon < 1e-4 and d_kl < 1e-4 and d_mean < 1e-4, rows[:3]
        assert not bad, bad
    else:
        _, exact, _ = vo.vae_loss_and_grads(params, frames, frames, eps, beta=1.0, storage="fp32")
        bad = {}
        for k in grads:
            e_dev, e_emul = rel_err(g[k], exact[k]), rel_err(grads[k], exact[k])
            rows.append(("grad " + k, "dev-vs-exact %.3e  emulation-vs-exact %.3e  dev-vs-emulation %.3e" % (e_dev, e_emul, worst[k])))
            if e_dev > 1.25 * e_emul + 2e-3:
                bad[k] = (e_dev, e_emul)
        _dev_table("B=512 bf16 HIP path vs the oracle's bf16-storage emulation (limits 2e-3 / 2e-2 / 3e-2 / 1.25 x emulation + 2e-3):", rows)
        assert d_recon < 2e-3 and d_kl < 2e-2 and d_mean < 3e-2, rows[:3]
        assert not bad, bad
    # the optimiser half of the same step: TF-Adam on the device gradients vs the oracle's AdamTF on the ORACLE gradients
    adam = vo.AdamTF({k: v.shape for k, v in params.items()})
    want = {k: v.copy() for k, v in params.items()}
    adam.step(want, grads, 1e-4)
    m._adam_step()
    got_p = m.dev.export_params()
    lim = (0.02 if precision == "fp32" else (0.05 if precision == "bf16x3" else 0.5)) * 1e-4
    frac = {k: float(np.mean(np.abs((got_p[k] - params[k]) - (want[k] - params[k])) > lim)) for k in want}
    _dev_table("fraction of weights whose Adam update differs by more than %.0e:" % lim, [(k, "%.2e" % v) for k, v in frac.items()])
    assert max(frac.values()) < (5e-3 if precision == "fp32" else (2e-2 if precision == "bf16x3" else 0.05)), frac


def test_kernel_generations_agree_at_batch_512(tmp_path):
    """BASELINE configs[1] size (batch 512, bf16): one forward + backward on the production dispatch (tapconv / tapwgrad / narrow
    kernels) and on the first-generation kernels, both against the fp32 engine as truth.  Losses agree to 1e-5.  Gradients: the two
    bf16 paths differ from each other by bf16 storage noise (ReLU-mask flips of near-zero pre-activations, largest right after the
    64-d bottleneck: dense1 / deconv1 see 3-4 % between any two bf16 accumulation orders and 6-7 % against fp32), so the criterion
    is that the production path is no further from the fp32 gradients than the simple kernels are -- a size-independent check that
    the raw-staged kernels handle full-size grids, split ranges and the slab reduction exactly like the simple ones."""
    from mi355 import lib as milib
    L = milib.get()
    B = 512
    rng = np.random.RandomState(5)
    frames = (rng.randint(0, 256, (B, 80, 160, 3)).astype(np.float32) / 255.0)
    eps = rng.standard_normal((B, 64)).astype(np.float32)
    params = trained_like_params()

    def run(cfg, precision="bf16"):
        prev = {k: L.mi_set_tuning(k, v) for k, v in cfg.items()}
        try:
            m = make(tmp_path, precision, params=params)
            src = m._frames(frames, 38400, "src")
            e = m._eps(B, eps)
            m.dev.forward(src, src, None, B, 1.0 / B, e, 1, 1)
            m.dev.backward(src, None, e, 1.0 / B, 0)
            return m.dev.losses.cpu().numpy().copy(), m.dev.export_grads()
        finally:
            for k, v in prev.items():
                L.mi_set_tuning(k, v)

    l_ref, g_ref = run({}, "fp32")
    l_new, g_new = run({0: 1, 1: 300, 3: 1, 4: 1})
    l_old, g_old = run({0: 0, 1: -1, 3: 0, 4: 0})
    assert abs(l_new[0] / l_old[0] - 1) < 1e-5 and abs(l_new[1] / l_old[1] - 1) < 2e-3, (l_new, l_old)
    assert abs(l_new[0] / l_ref[0] - 1) < 1e-4, (l_new, l_ref)
    bad = {}
    for k in g_ref:
        e_new, e_old = rel_err(g_new[k], g_ref[k]), rel_err(g_old[k], g_ref[k])
        if e_new > 1.25 * e_old + 2e-3 or rel_err(g_new[k], g_old[k]) > 8e-2:
            bad[k] = (e_new, e_old, rel_err(g_new[k], g_old[k]))
    assert not bad, bad


# ----------------------------------------------------------------------------------------------------------------------
# MlpVAE (reference vae/models.py:271-299) against the oracle's restatement
def _mlp_params(seed, src_shape, tgt_shape, enc, dec, scale=1.0):
    p = vo.init_mlp_vae_params(seed, z_dim=64, source_shape=src_shape, target_shape=tgt_shape, encoder_sizes=enc, decoder_sizes=dec)
    rng = np.random.RandomState(seed + 1)
    for k in p:
        if k.endswith("bias"):
            p[k] = (0.05 * rng.standard_normal(p[k].shape)).astype(np.float32)       # non-zero biases: the bias paths are exercised
        else:
            p[k] = (scale * p[k]).astype(np.float32)
    return p


@pytest.mark.parametrize("variant", ["rgb", "seg_small", "kl_tol_mse"])
def test_mlp_vae_fp32_matches_oracle(tmp_path, variant):
    """Losses, every gradient tensor, one TF-Adam step, encode / reconstruct / generate_from_latent of the MlpVAE in exact-fp32 mode."""
    src_shape = (80, 160, 3)
    tgt_shape = (80, 160, 1) if variant == "seg_small" else src_shape
    enc, dec = ((64, 32), (32, 64)) if variant == "seg_small" else ((512, 256), (256, 512))
    kw = dict(beta=1.0, kl_tolerance=0.0, loss_fn="bce")
    if variant == "kl_tol_mse":
        kw = dict(beta=4.0, kl_tolerance=0.5, loss_fn="mse")
    B = 6
    rng = np.random.RandomState(11)
    src = (rng.randint(0, 256, (B,) + src_shape).astype(np.float32) / 255.0)
    tgt = src if tgt_shape == src_shape else (rng.randint(0, 2, (B,) + tgt_shape).astype(np.float32))
    eps = rng.standard_normal((B, 64)).astype(np.float32)
    params = _mlp_params(3, src_shape, tgt_shape, enc, dec)
    m = MlpVAE(np.array(src_shape), np.array(tgt_shape), encoder_sizes=enc, decoder_sizes=dec, z_dim=64, model_dir=str(tmp_path / "m"), precision="fp32",
               beta=kw["beta"], kl_tolerance=kw["kl_tolerance"], loss_fn={"bce": bce_loss, "mse": mse_loss}[kw["loss_fn"]], learning_rate=1e-5)        # (one Adam step moves EVERY weight by ~lr: 1e-3 on 38400-wide layers blows the logits up to thousands)
    m.set_weights(params)
    m.init_session(init_logging=False)
    (recon, kl, _), grads, fw = vo.mlp_vae_loss_and_grads(params, src, tgt, eps, **kw)
    s_dev = m._frames(src, int(np.prod(src_shape)), "src")
    t_dev = s_dev if tgt is src else m._frames(tgt, int(np.prod(tgt_shape)), "tgt")
    e_dev = m._eps(B, eps)
    m.dev.forward(s_dev, t_dev, None, B, 1.0 / B, e_dev, 1, 1)
    m.dev.backward(s_dev, None, e_dev, 1.0 / B, 0)
    l = m.dev.losses.cpu().numpy()
    assert l[0] == pytest.approx(recon, rel=1e-4) and l[1] == pytest.approx(kl, rel=1e-4, abs=4e-6)
    g = m.dev.export_grads()
    assert set(g) == set(grads)
    for k in grads:
        assert rel_err(g[k], grads[k]) < 1e-4, (k, rel_err(g[k], grads[k]))
    # one Adam step (tf.train.AdamOptimizer form) against the oracle's AdamTF
    adam = vo.AdamTF({k: v.shape for k, v in params.items()})
    want = {k: v.copy() for k, v in params.items()}
    adam.step(want, g, 1e-5)            # the DEVICE gradients: the first Adam step is lr * g / (|g| + 1e-8), i.e. +-lr for any |g| >> 1e-8 and
                                        # arbitrarily sensitive where |g| ~ 1e-8 (dead-ReLU columns of the 38400-wide layers)
    m._adam_step()
    got = m.dev.export_params()
    for k in want:
        assert np.abs(got[k] - want[k]).max() <= 2e-6 + 1e-5 * np.abs(want[k]).max(), k
    # inference surface on the updated weights (the device's own copy: the comparison above allows 1e-5 of Adam rounding per weight, which
    # the 38400-wide first layer would amplify past the 1e-4 used here)
    import torch
    p_t = {k: torch.from_numpy(v) for k, v in got.items()}
    fw2 = vo.mlp_vae_forward(p_t, src, sample=False)
    assert rel_err(m.encode(src), fw2["mean"].numpy()) < 1e-4
    z = rng.standard_normal((3, 64)).astype(np.float32)
    dec_ref = torch.sigmoid(vo.mlp_vae_forward(p_t, None, z_override=z)["logits"]).numpy()
    assert rel_err(m.generate_from_latent(z), dec_ref) < 1e-4
    if tgt_shape != src_shape:          # the reference reshapes reconstructions with the SOURCE shape (vae/models.py:193-197): same failure here
        with pytest.raises(ValueError):
            m.reconstruct(src, eps=eps)
        return
    rec = m.reconstruct(src, eps=eps)
    rec_ref = torch.sigmoid(vo.mlp_vae_forward(p_t, src, eps, sample=True)["logits"]).numpy()
    assert len(rec) == B and rec[0].shape == src_shape
    assert rel_err(np.stack([r.reshape(-1) for r in rec]), rec_ref) < 1e-4


def test_mlp_vae_bf16_trains_and_checkpoints(tmp_path):
    """bf16 storage mode of the MlpVAE: losses within bf16 tolerance of the oracle's bf16-storage emulation, gradients as close to the fp32
    truth as the emulation is, a few SGD steps lower the loss, and the state dict round-trips through a checkpoint (TF variable names)."""
    src_shape, enc, dec, B = (80, 160, 3), (512, 256), (256, 512), 16
    rng = np.random.RandomState(5)
    src = (rng.randint(0, 256, (B,) + src_shape).astype(np.float32) / 255.0)
    eps = rng.standard_normal((B, 64)).astype(np.float32)
    params = _mlp_params(7, src_shape, src_shape, enc, dec)
    m = MlpVAE(np.array(src_shape), z_dim=64, model_dir=str(tmp_path / "b"), precision="bf16", learning_rate=2e-5)
    m.set_weights(params)
    m.init_session(init_logging=False)
    (r32, k32, _), g32, _ = vo.mlp_vae_loss_and_grads(params, src, src, eps)
    (rb, kb, _), gb, _ = vo.mlp_vae_loss_and_grads(params, src, src, eps, storage="bf16")
    s_dev, e_dev = m._frames(src, 38400, "src"), m._eps(B, eps)
    m.dev.forward(s_dev, s_dev, None, B, 1.0 / B, e_dev, 1, 1)
    m.dev.backward(s_dev, None, e_dev, 1.0 / B, 0)
    l = m.dev.losses.cpu().numpy()
    assert l[0] == pytest.approx(rb, rel=2e-3) and l[1] == pytest.approx(kb, rel=2e-2, abs=1e-4)
    g = m.dev.export_grads()
    for k in g32:
        e_dev_, e_emul = rel_err(g[k], g32[k]), rel_err(gb[k], g32[k])
        assert e_dev_ <= 2 * e_emul + 1e-2, (k, e_dev_, e_emul)
    m.dev.grads.zero_()
    first = m.train_step(src, src, eps=eps)
    for _ in range(5):
        last = m.train_step(src, src, eps=eps)
    assert last[0] < first[0]
    m.step_idx = 3
    sd = m.state_dict()
    assert "vae/encoder/dense_1/kernel" in sd and "vae/vae/decoder/dense_2/bias/Adam_1" in sd and sd["vae/decoder/dense_2/kernel"].shape == (512, 38400)
    m.save()
    m2 = MlpVAE(np.array(src_shape), z_dim=64, model_dir=str(tmp_path / "b"), precision="bf16", learning_rate=2e-5)
    m2.init_session(init_logging=False)
    assert m2.load_latest_checkpoint() is True and m2.get_step_idx() == 3
    assert m.train_step(src, src, eps=eps) == pytest.approx(m2.train_step(src, src, eps=eps), rel=1e-6)


def test_comm_c_abi_one_rank_world():
    """The collective half of the C ABI against the real RCCL on this box, world size 1 (what one GPU allows): rendezvous id, communicator,
    in-stream and side-stream all-reduce (identity at one rank), the join, broadcast, destroy.  librccl.so.1 is bound at run time."""
    import ctypes
    from mi355 import lib as milib
    L = milib.get()
    assert L.mi_comm_id_bytes() == 128
    idb = np.zeros(128, np.uint8)
    L.mi_comm_unique_id(idb.ctypes.data)
    assert idb.any()
    h = ctypes.c_void_p()
    L.mi_comm_init(ctypes.addressof(h), 0, 1, idb.ctypes.data)
    try:
        st = torch.cuda.current_stream().cuda_stream
        x = torch.randn(100003, device="cuda")
        ref = x.clone()
        L.mi_allreduce_sum_f32(h, st, x.data_ptr(), x.numel())
        y = x * 2                                           # work queued behind the bucket on the caller's stream
        L.mi_allreduce_sum_f32_async(h, st, x.data_ptr(), 4096)
        L.mi_allreduce_sum_f32_async(h, st, x.data_ptr() + 4 * 4096, x.numel() - 4096)
        L.mi_comm_wait(h, st)
        L.mi_comm_wait(h, st)                               # nothing pending: no-op
        L.mi_broadcast(h, st, x.data_ptr(), x.numel() * 4, 0)
        torch.cuda.synchronize()
        assert torch.equal(x, ref) and torch.equal(y, ref * 2)
        with pytest.raises(milib.MiError):
            L.mi_broadcast(h, st, x.data_ptr(), 16, 3)      # root outside the communicator
    finally:
        L.mi_comm_destroy(h)


def _run_two_ranks(tmp_path, backend, port):
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "dp")
    os.makedirs(out)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", port, os.path.join(ROOT, "tests", "dp_gpu_worker.py"), out, backend], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(os.path.join(out, "rank0.npz")), np.load(os.path.join(out, "rank1.npz"))


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_data_parallel_two_ranks_on_the_gpu(tmp_path, backend):
    """The product's data-parallel path end to end with world_size 2 (tests/dp_gpu_worker.py): the all-reduced gradients of a global minibatch
    equal the single-process gradients of the same minibatch, the epoch metrics agree, both ranks end with identical parameters, and those
    equal the single-process result up to Adam's sensitivity where |g| ~ 1e-8.
    gloo: two ranks share this GPU (RCCL refuses two ranks on one device; gloo reduces device tensors through the host).
    nccl: one device per rank, gradients summed by the library's own communicator (mi_comm, RCCL through the C ABI); needs two devices."""
    import dp_gpu_worker as W
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("the RCCL run needs two devices; this box has %d" % torch.cuda.device_count())
    r0, r1 = _run_two_ranks(tmp_path, backend, "29541" if backend == "gloo" else "29543")
    assert int(r0["world"]) == 2
    if backend == "nccl":
        assert str(r0["comm"]).startswith("mi_comm"), str(r0["comm"])
    frames, eps = W.dataset()
    m = W.build(str(tmp_path / "single"), trained_like_params(2))
    grads, losses, params = W.run(m, frames, eps)
    for k, g in grads.items():
        key = "g|" + k.replace("/", "|")
        assert np.array_equal(r0[key], r1[key]), k                                   # the same reduced buffer on both ranks
        assert rel_err(r0[key], g) < 2e-5, (k, rel_err(r0[key], g))
    assert np.allclose(r0["losses"], losses, rtol=1e-5) and np.array_equal(r0["losses"], r1["losses"])
    for k, v in params.items():
        key = "p|" + k.replace("/", "|")
        assert np.array_equal(r0[key], r1[key]), k                                   # replicas stay bit-identical
        assert rel_err(r0[key], v) < 2e-2, (k, rel_err(r0[key], v))
    # PPO.train under data parallelism (each rank passes its 16 of 32 rows): the global loss scalars and the replicas against one process on all 32
    pl, pp = W.run_ppo(str(tmp_path / "ppo_single"))
    assert np.array_equal(r0["ppo_params"], r1["ppo_params"]) and np.array_equal(r0["ppo_losses"], r1["ppo_losses"])
    assert np.allclose(r0["ppo_losses"], pl, rtol=2e-4, atol=1e-6), (r0["ppo_losses"], pl)
    assert rel_err(r0["ppo_params"], pp) < 1e-4, rel_err(r0["ppo_params"], pp)


def test_uint8_frame_tables_upload_bit_exact(tmp_path):
    """Raw uint8 camera frames handed to the training surface are normalised on the device to exactly the float32 values the reference's
    host preprocessing (frame / 255.0) produces: same resident table, same epoch metrics."""
    m = make(tmp_path, "fp32", params=trained_like_params(3))
    u8 = np.random.RandomState(9).randint(0, 256, (12, 80, 160, 3), dtype=np.uint8)
    f32 = u8.astype(np.float32) / 255.0
    a, b = m._frames(u8, 38400, "src"), m._frames(f32, 38400, "src")
    assert a.dtype == torch.float32 and torch.equal(a, b)
    eps = [np.random.RandomState(1).standard_normal((4, 64)).astype(np.float32) for _ in range(3)]
    np.random.seed(3)                                   # the epoch loops draw the reference's legacy-numpy permutation
    r1 = m.evaluate(u8, u8, 4, eps=eps)
    np.random.seed(3)
    r2 = m.evaluate(f32, f32, 4, eps=eps)
    assert list(r1) == list(r2)


@pytest.mark.parametrize("kind", ["seg", "rgb"])
def test_reference_load_vae_and_encode_state_chain(tmp_path, kind, monkeypatch):
    """the drop-in driven exactly as the reference's vae_common.py drives `vae.models` (tests/ref_call_chain.py: the
    restatement is pinned to the real vae_common.py by tests/golden/vae_common_calls.json, test_host_logic.py): ConvVAE(source_shape=array,
    target_shape=array, z_dim, models_dir="vae", model_dir, training=False) -> init_session(init_logging=False) -> load_latest_checkpoint()
    must be True -> np.append(vae.encode([frame / 255])[0], [steer, throttle, speed]) -- against the oracle, 1e-4 (default precision = fp32).
    The checkpoint on disk is a TensorFlow bundle (the reference's own format), written by a trained-mode model first."""
    import ref_call_chain as rc
    import vae.models as drop_in
    monkeypatch.setenv("MI355_CKPT_FORMAT", "tf")
    monkeypatch.delenv("MI355_PRECISION", raising=False)
    tc = 1 if kind == "seg" else 3
    name = ("seg_" if kind == "seg" else "") + "bce_cnn_zdim64_beta1_kl_tolerance0.0_data"
    model_dir = str(tmp_path / "vae" / "models" / name)
    params = trained_like_params(4, tc)
    trainer = ConvVAE(np.array([80, 160, 3]), np.array([80, 160, tc]), z_dim=64, model_dir=model_dir)       # what vae/train_vae.py leaves behind
    trainer.set_weights(params)
    trainer.init_session(init_logging=False)
    trainer.step_idx = 232
    trainer.save()
    vae = rc.restated_load_vae(drop_in, model_dir)
    assert isinstance(vae, drop_in.ConvVAE) and vae.training is False and vae.precision == "fp32" and vae.get_step_idx() == 232
    assert tuple(vae.target_shape) == (80, 160, tc) and vae.z_dim == 64
    env = rc.StubEnv(rc._frame())
    state = rc.restated_encode_state(vae, env)
    assert state.shape == (67,) and state.dtype == np.float64                        # np.append(float32[64], python floats) -> float64
    o = vo.OracleVAE(params=params, target_shape=(80, 160, tc), training=False)
    want = np.append(o.encode([env.observation.astype(np.float32) / 255.0])[0], [-0.25, 0.5, 12.5])
    assert np.array_equal(state[64:], want[64:])
    assert rel_err(state[:64], want[:64]) < 1e-4
    with pytest.raises(Exception, match="Failed to load VAE"):                       # vae_common.py:25-26 on a directory without checkpoints
        rc.restated_load_vae(drop_in, str(tmp_path / "vae" / "models" / ("empty_" + name)))


def test_engine_noise_stream_matches_its_restatement_and_is_normal(tmp_path):
    """The engine-side N(0,1) source (Philox4x32-10 + Box-Muller inside the reparameterisation kernel; standalone mi_normal_philox): the device
    stream equals the numpy restatement pinned by Random123's known-answer vectors (tests/philox_ref.py) to float32 rounding of the
    transcendental functions, successive sampling passes continue the stream (no repeats), and the draws pass a Kolmogorov-Smirnov test."""
    import philox_ref as pr
    from scipy import stats
    m = make(tmp_path, "fp32", params=trained_like_params())
    L, dev = m.dev.L, m.dev
    out = torch.empty(5000, device=dev.device)
    L.mi_normal_philox(dev.stream(), 0xABCDEF0123, 77, out.data_ptr(), out.numel())
    assert np.allclose(out.cpu().numpy(), pr.normal(0xABCDEF0123, 77, 5000), rtol=0, atol=2e-5)
    B = 32
    frames = synth_frames(B)
    src = m._frames(frames, 38400, "src")
    draws = []
    for _ in range(3):                                                  # eps=None: the engine draws
        dev.forward(src, src, None, B, 1.0 / B, None, 1, 0)
        draws.append(dev._view(5, B * 64).cpu().numpy().copy())          # z = mean + sigma * eps
    assert not np.allclose(draws[0], draws[1]) and not np.allclose(draws[1], draws[2])
    want = pr.normal(m._noise_seed, 0, 3 * B * 64).reshape(3, B * 64)
    mean, lv = dev._view(1, B * 64).cpu().numpy(), dev._view(2, B * 64).cpu().numpy()
    for k in range(3):
        assert np.allclose((draws[k] - mean) / np.exp(0.5 * lv), want[k], atol=5e-4), k
    big = torch.empty(1 << 18, device=dev.device)
    L.mi_normal_philox(dev.stream(), 99, 0, big.data_ptr(), big.numel())
    x = big.cpu().numpy().astype(np.float64)
    assert stats.kstest(x, "norm").pvalue > 1e-3 and abs(x.mean()) < 0.01 and abs(x.std() - 1) < 0.01


def test_uint8_frame_tables_in_the_bf16_engine(tmp_path):
    """Raw uint8 camera frames kept as bytes in HBM (bf16 engine): conv1 forward / filter gradient and the fused loss normalise k / 255 in
    registers.  Same losses and gradients as the float32 table of the same frames: the kernels' arithmetic is identical value for value (the
    in-register forms are exact, tests/test_oracle_golden.py); what may differ is the order of the filter-gradient atomics."""
    params = trained_like_params()
    B = 24
    u8 = np.random.RandomState(9).randint(0, 256, (B, 80, 160, 3), dtype=np.uint8)
    f32 = u8.astype(np.float32) / 255.0
    eps = np.random.RandomState(1).standard_normal((B, 64)).astype(np.float32)
    res = []
    for table in (u8, f32):
        m = make(tmp_path, "bf16", params=params)
        t = m._frames(table, 38400, "src", keep_u8_ok=True)
        assert t.dtype == (torch.uint8 if table is u8 else torch.float32)
        e = m._eps(B, eps)
        m.dev.forward(t, t, None, B, 1.0 / B, e, 1, 1)
        losses = m.dev.losses.cpu().numpy().copy()
        m.dev.backward(t, None, e, 1.0 / B, 0)
        res.append((losses, m.dev.export_grads(), m.encode(f32)))
    (l8, g8, z8), (lf, gf, zf) = res
    assert np.array_equal(l8, lf), (l8, lf)
    for k in gf:
        assert rel_err(g8[k], gf[k]) < 1e-5, (k, rel_err(g8[k], gf[k]))
    # the public surface: train_step(uint8, same uint8) trains on the byte table; a float32 engine converts it on upload
    m = make(tmp_path, "bf16", params=params)
    r8 = m.train_step(u8, u8, eps=eps)
    m2 = make(tmp_path, "bf16", params=params)
    rf = m2.train_step(f32, f32, eps=eps)
    assert r8 == pytest.approx(rf, rel=1e-6)
    m3 = make(tmp_path, "fp32", params=params)
    assert m3._frames(u8, 38400, "src", keep_u8_ok=True).dtype == torch.float32


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_captured_graph_step_equals_eager_step(tmp_path, precision, monkeypatch):
    """mi_vae_train_step: the hipGraph replay of the step (minibatch rows and Adam's step size staged through device memory, noise stream
    continued from its device-side offset) gives the same trajectory as the eager launches: losses of 4 steps on different rows and the
    parameters afterwards (identical kernels and arguments; only the order of fp32 atomics in the filter gradients may differ)."""
    params = trained_like_params()
    N, B = 64, 16
    frames = synth_frames(N, seed=5)
    idx = np.random.RandomState(3).permutation(N)[:4 * B].reshape(4, B).astype(np.int32)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MI355_GRAPH", mode)
        m = make(tmp_path, precision, params=params, seed=11)
        table = m._frames(frames, 38400, "src")
        idx_dev = torch.from_numpy(idx).to(m.dev.device)
        losses = []
        for i in range(4):
            m._train_minibatch(table, table, idx_dev[i], B, 1.0 / B, None)        # engine-drawn noise: same seed, same stream in both modes
            losses.append(m.dev.losses.cpu().numpy().copy())
        out[mode] = (np.array(losses), m.dev.export_params(), float(m.beta1_power))
    (l0, p0, b0), (l1, p1, b1) = out["0"], out["1"]
    assert b0 == b1 == pytest.approx(0.9 ** 5, rel=1e-6)
    assert np.allclose(l0, l1, rtol=2e-6 if precision == "fp32" else 2e-4), (l0, l1)
    assert not np.allclose(l1[0], l1[1])                                            # different rows each step: the staged indices are live
    for k in p0:
        frac = np.mean(np.abs(p0[k] - p1[k]) > 0.5 * 1e-4)                          # an Adam step moves a weight by ~lr = 1e-4
        assert frac < (1e-3 if precision == "fp32" else 2e-2), (k, frac)


def test_bench_data_parallel_path_with_two_ranks_on_one_gpu(tmp_path):
    """bench.py --gpus 2 exactly as the driver launches it (torch.distributed.run, two ranks), with the two ranks sharing this GPU over gloo
    (RCCL refuses two ranks on one device): the weak-scaling bookkeeping, the per-rank gather, the exposed all-reduce measurement and the
    parameter re-broadcast run end to end and the one JSON line has the contract's fields."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", MI355_BENCH_BACKEND="gloo", MI355_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29547",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "3", "--batch", "64", "--pool", "256"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2"
    assert d["value"] == pytest.approx(128 * 4 / (d["ms_per_step"] * 4e-3), rel=1e-6)
    dp = d["data_parallel"]
    assert len(dp["ms_per_step_by_rank"]) == 2 and dp["gradient_bytes_per_step"] > 0 and "transport" in dp
    assert d["roofline"] is not None and d["cpu_baseline"] is None


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["bf16", "bf16x3", "fp32"])
def test_workspace_guards_stay_intact_through_training_and_inference(tmp_path, monkeypatch, precision):
    """SURVEY 5 (sanitizer row): the engine's debug mode puts 256 guard bytes behind every workspace region; no kernel of a train step, an
    evaluation, encode / reconstruct / generate may touch them -- and a write past a region IS reported (one guard corrupted on purpose)."""
    import torch
    monkeypatch.setenv("MI355_DEBUG_GUARDS", "1")
    m = make(tmp_path, precision, params=trained_like_params(2))
    frames = synth_frames(96, seed=7)
    n, bad, _ = m.dev.check_guards()
    assert n >= 30 and bad == 0
    np.random.seed(0)
    m.train_one_epoch(frames, frames, 32)
    m.evaluate(frames[:40], frames[:40], 32)                 # partial last minibatch
    z = m.encode(frames[:5]); m.reconstruct(frames[:3]); m.generate_from_latent(z)
    n2, bad, _ = m.dev.check_guards()
    assert n2 == n and bad == 0, m.dev.L.cdll.mi_last_error().decode()
    _, _, off = m.dev.check_guards(guard_index=3)
    m.dev.workspace[off + 17] = 0                            # what an out-of-bounds store would do
    torch.cuda.synchronize()
    _, bad, _ = m.dev.check_guards()
    assert bad == 1 and "guard 3" in m.dev.L.cdll.mi_last_error().decode()
