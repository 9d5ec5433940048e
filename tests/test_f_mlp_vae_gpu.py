"""MlpVAE (reference vae/models.py:271-299) against the oracle's restatement.
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


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_mlp_vae_backward_in_two_parts_and_whole_step_call(tmp_path, precision):
    """The engine's data-parallel surface and its one-call step (round 4, csrc/mlp_engine.hip): backward(part 1) + backward(part 2) -- between which the host all-reduces
    grads[decoder_offset:] -- leaves bitwise the gradients of backward(part 0), garbage in the gradient buffer beforehand included (backward STORES); mi_mlpvae_train_step
    equals forward + backward + apply_adam bitwise (parameters, both Adam slots), at a batch that is not a multiple of the 32-row stages and through an index vector."""
    import torch
    src_shape, enc, dec, B = (80, 160, 3), (512, 256), (256, 512), 37
    rng = np.random.RandomState(2)
    table = (rng.randint(0, 256, (50,) + src_shape).astype(np.float32) / 255.0)
    eps = rng.standard_normal((B, 64)).astype(np.float32)
    idx = rng.permutation(50)[:B].astype(np.int32)
    params = _mlp_params(9, src_shape, src_shape, enc, dec)

    def model(tag):
        m = MlpVAE(np.array(src_shape), z_dim=64, model_dir=str(tmp_path / tag), precision=precision, learning_rate=1e-4)
        m.set_weights(params)
        m.init_session(init_logging=False)
        return m
    m = model("a")
    s_dev, e_dev, i_dev = m._frames(table, 38400, "src"), m._eps(B, eps), torch.from_numpy(idx).cuda()
    d = m.dev
    d.forward(s_dev, s_dev, i_dev, B, 1.0 / B, e_dev, 1, 1)
    d.grads.fill_(123.0)
    d.backward(s_dev, i_dev, e_dev, 1.0 / B, 0)
    whole = d.grads.clone()
    d.grads.fill_(-7.0)
    d.backward(s_dev, i_dev, e_dev, 1.0 / B, 1)
    assert torch.equal(d.grads[d.decoder_offset:], whole[d.decoder_offset:]) and bool((d.grads[:d.decoder_offset] == -7.0).all())
    d.backward(s_dev, i_dev, e_dev, 1.0 / B, 2)
    assert torch.equal(d.grads, whole)
    assert [(p_, lo, hi) for p_, lo, hi in d.grad_buckets] == [(1, d.decoder_offset, d.n_flat), (2, 0, d.decoder_offset)]
    # three steps: op by op against the one-call form
    m2 = model("b")
    for _ in range(3):
        d.forward(s_dev, s_dev, i_dev, B, 1.0 / B, e_dev, 1, 1)
        d.backward(s_dev, i_dev, e_dev, 1.0 / B, 0)
        m._adam_step()
        m2._train_minibatch(s_dev, s_dev, i_dev, B, 1.0 / B, e_dev)
    for a_, b_ in ((d.params, m2.dev.params), (d.adam_m, m2.dev.adam_m), (d.adam_v, m2.dev.adam_v), (d.weights_t, m2.dev.weights_t)):
        assert torch.equal(a_, b_)
    assert not torch.equal(d.params, torch.from_numpy(d._to_flat(params)).cuda())
