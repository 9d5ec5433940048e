"""Multi-step TF-Adam trajectories of the HIP path against the SAME steps run in float64 -- the most tolerance-sensitive statements of the suite, collected last.
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
    # Adam's first steps move every weight by ~lr regardless of gradient scale (sign-like updates): a last-bit gradient difference on a ~zero gradient
    # entry moves that weight by up to 2 lr the other way, in the oracle as much as on the device (one such flip in deconv3's 51,200-element kernel is an
    # RMS error of 2.6e-3 of the update -- the round-3 form of this test, an RMS bound with a 2e-3 floor, failed on exactly that).  Both are therefore
    # measured against the SAME three steps run in float64 (forward, gradients and the Adam recurrence), and the statement is flip-tolerant
    # (the form of tests/test_ref_graph_gpu.py): per tensor,
    #   (1) the number of entries whose 3-step update is off by more than `lim` is at most 2 x the fp32 oracle's own count + 2, or `frac` of the tensor;
    #   (2) over the remaining entries the RMS update error is at most 2 x the oracle's + `floor`.
    from collections import OrderedDict
    ex = OrderedDict((k, v.astype(np.float64)) for k, v in params.items())
    adam64 = vo.AdamTF(OrderedDict((k, v.shape) for k, v in ex.items()), dtype=np.float64)
    for s in range(3):
        ee = np.random.RandomState(100 + s).standard_normal((B, 64)).astype(np.float32)
        _, g64, _ = vo.vae_loss_and_grads(ex, frames, frames, ee, beta=1.0, dtype=torch.float64)
        adam64.step(ex, g64, 1e-4)
    step3 = 3e-4                                             # three steps of lr = 1e-4
    lim = {"fp32": 0.02, "bf16x3": 0.05, "bf16": 0.5}[precision] * step3
    frac = {"fp32": 1e-3, "bf16x3": 2e-2, "bf16": 5e-2}[precision]
    floor = {"fp32": 0.002, "bf16x3": 0.025, "bf16": 0.25}[precision] * step3
    rows, bad = [], {}
    for k, v in o.params.items():
        upd_x = ex[k] - params[k].astype(np.float64)
        d_o = np.abs((v.astype(np.float64) - params[k]) - upd_x).ravel()
        d_d = np.abs((got_p[k].astype(np.float64) - params[k]) - upd_x).ravel()
        n_o, n_d = int((d_o > lim).sum()), int((d_d > lim).sum())
        rms_o = float(np.sqrt(np.mean(d_o[d_o <= lim] ** 2))) if (d_o <= lim).any() else 0.0
        rms_d = float(np.sqrt(np.mean(d_d[d_d <= lim] ** 2))) if (d_d <= lim).any() else 0.0
        rows.append((k, d_d.size, n_d, n_o, rms_d / step3, rms_o / step3))
        if n_d > max(2 * n_o + 2, frac * d_d.size) or rms_d > 2.0 * rms_o + floor:
            bad[k] = rows[-1][1:]
    print("\n3 Adam steps (%s) vs the float64 trajectory: tensor, size, entries off by > %.3g of the update (device, oracle), RMS of the rest / (3 lr) (device, oracle):" % (precision, lim / step3))
    for r in rows:
        print("  %-38s %8d %6d %6d   %.3e  %.3e" % r)
    assert not bad, bad
    assert m2.beta1_power == pytest.approx(0.9 ** 4, rel=1e-6)
