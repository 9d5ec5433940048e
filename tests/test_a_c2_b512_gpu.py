"""BASELINE configs[1] -- the BENCHMARKED configuration (ConvVAE, batch 512, one MI355X) -- against the CPU oracle, all three storage modes.

Collected FIRST (file name) so that the driver's `pytest -m gpu -x` reaches the configuration the bench line is quoted on before anything else.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vae_oracle as vo  # noqa: E402
from vae.models import ConvVAE, MlpVAE, bce_loss, bce_loss_v2, mse_loss  # noqa: E402,F401
from vae_gpu_common import synth_frames, make, rel_err, trained_like_params, _dev_table, _mlp_params  # noqa: E402,F401


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


@pytest.mark.parametrize("precision", ["bf16", "bf16x3", "fp32"])
def test_two_runs_of_a_step_are_bitwise_equal(tmp_path, precision):
    """Round 4: every filter / bias gradient of every engine reduces its position splits through per-split partial sums added up in a
    FIXED order (no fp32 atomics anywhere on the training path), so the same step run twice gives the same bits: all 22 gradient tensors of two backward
    passes at the benchmarked batch size, and the parameters of two models after three whole SGD steps (mi_vae_train_step: two streams, deferred reductions)."""
    B = 512 if precision != "fp32" else 128               # (the exact-fp32 engine takes 5 ms per step at 512: 128 rows exercise the same split counts)
    params = trained_like_params()
    frames = synth_frames(B, seed=21)
    eps = np.random.RandomState(8).standard_normal((B, 64)).astype(np.float32)
    m = make(tmp_path, precision, params=params)
    src = m._frames(frames, 38400, "src")
    e = m._eps(B, eps)
    runs = []
    for _ in range(3):
        m.dev.grads.zero_()
        m.dev.forward(src, src, None, B, 1.0 / B, e, 1, 1)
        m.dev.backward(src, None, e, 1.0 / B, 0)
        runs.append((m.dev.losses.cpu().numpy().copy(), m.dev.grads.cpu().numpy().copy()))
    for l, g in runs[1:]:
        assert np.array_equal(l, runs[0][0])
        assert np.array_equal(g, runs[0][1]), {k: int((a != b).sum()) for (k, a), b in zip(m.dev._from_flat(g).items(), m.dev._from_flat(runs[0][1]).values()) if (a != b).any()}
    assert np.abs(runs[0][1]).max() > 0
    finals = []
    for _ in range(2):
        m2 = make(tmp_path, precision, params=params)
        for s_ in range(3):
            ee = np.random.RandomState(100 + s_).standard_normal((B, 64)).astype(np.float32)
            m2.train_step(frames, frames, eps=ee)
        finals.append(m2.dev.params.cpu().numpy().copy())
    assert np.array_equal(finals[0], finals[1])
