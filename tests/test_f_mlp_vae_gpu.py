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


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_mlp_vae_uint8_frame_tables_equal_the_float_tables(tmp_path, precision):
    """Round 5: the MlpVAE engine reads uint8 camera-byte tables (frames_u8 of mi_mlpvae_*; mi_gather_rows_cast_u8 where the minibatch rows are staged,
    mi_bce_logits_fwd_bwd_u8 for the labels): float32(k) / float32(255) exactly, so two SGD steps through gathered minibatches give BITWISE the losses, posterior means
    and parameters of the same steps on the float32 table of k / 255 (the reference's host preprocessing, vae/train_vae.py:15-18), in both storage modes; the op-level
    entry points against numpy as well (row length with a scalar tail, unaligned rows)."""
    from mi355 import lib as milib
    L = milib.get()
    rng = np.random.RandomState(4)
    N, B = 40, 16
    u8 = rng.randint(0, 256, (N, 80, 160, 3), dtype=np.uint8)
    f32 = u8.astype(np.float32) / np.float32(255.0)
    eps = rng.standard_normal((2, B, 64)).astype(np.float32)
    idx = torch.from_numpy(np.stack([rng.permutation(N)[:B] for _ in range(2)]).astype(np.int32)).cuda()
    params = _mlp_params(5, (80, 160, 3), (80, 160, 3), (64, 32), (32, 64))
    res = []
    for table in (u8, f32):
        m = MlpVAE(np.array([80, 160, 3]), encoder_sizes=(64, 32), decoder_sizes=(32, 64), z_dim=64, model_dir=str(tmp_path / ("m_%s_%s" % (precision, table.dtype))),
                   precision=precision, learning_rate=1e-4, seed=0)
        m.set_weights(params)
        m.init_session(init_logging=False)
        t = m._frames(table, 38400, "src", keep_u8_ok=True)
        assert t.dtype == (torch.uint8 if table.dtype == np.uint8 else torch.float32)
        losses = []
        for s in range(2):
            m._train_minibatch(t, t, idx[s], B, 1.0 / B, m._eps(B, eps[s]))
            losses.append(m.dev.losses.cpu().numpy().copy())
        mean = torch.empty(B, 64, device="cuda")
        m.dev.encode(t, idx[0], B, mean)
        res.append((np.array(losses), mean.cpu().numpy(), m.dev.export_params()))
        m.dev.close()
    (l8, m8, p8), (lf, mf, pf) = res
    assert np.array_equal(l8, lf) and np.array_equal(m8, mf)
    for k in pf:
        assert np.array_equal(p8[k], pf[k]), k
    # op level: odd row length (scalar tail) and a gather; labels through the loss kernel
    for row_len in (38400, 1003):
        tab = rng.randint(0, 256, (7, row_len), dtype=np.uint8)
        sel = np.array([5, 0, 6, 2], np.int32)
        td, sd = torch.from_numpy(tab).cuda(), torch.from_numpy(sel).cuda()
        for code, tt in ((milib.MI_F32, torch.float32), (milib.MI_BF16, torch.bfloat16)):
            out = torch.empty(4, row_len, device="cuda", dtype=tt)
            L.mi_gather_rows_cast_u8(torch.cuda.current_stream().cuda_stream, code, td.data_ptr(), sd.data_ptr(), 4, row_len, out.data_ptr())
            want = torch.from_numpy(tab[sel].astype(np.float32) / np.float32(255.0)).to(tt)
            assert torch.equal(out.cpu(), want), (row_len, code)
    P, Bq = 38400, 3
    logits = torch.from_numpy(rng.standard_normal((Bq, P)).astype(np.float32)).cuda()
    lab8_np = rng.randint(0, 256, (5, P), dtype=np.uint8)
    lab8 = torch.from_numpy(lab8_np).cuda()
    labf = torch.from_numpy(lab8_np.astype(np.float32) / np.float32(255.0)).cuda()      # the HOST quotient (correctly rounded; a device-side `t / 255.0` may multiply by a reciprocal)
    gi = torch.from_numpy(np.array([4, 1, 3], np.int32)).cuda()
    nch = L.mi_recon_loss_chunks(P)
    outs = []
    for u in (True, False):
        dl, part = torch.empty(Bq, P, device="cuda"), torch.empty(Bq * nch, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        if u:
            L.mi_bce_logits_fwd_bwd_u8(st, milib.MI_F32, logits.data_ptr(), lab8.data_ptr(), gi.data_ptr(), P, Bq, P, 0, 1.0 / Bq, dl.data_ptr(), part.data_ptr())
        else:
            L.mi_bce_logits_fwd_bwd(st, milib.MI_F32, logits.data_ptr(), labf.data_ptr(), gi.data_ptr(), P, Bq, P, 0, 1.0 / Bq, dl.data_ptr(), part.data_ptr())
        outs.append((dl.cpu().numpy(), part.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
