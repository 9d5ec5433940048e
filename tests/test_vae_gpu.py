"""Engine-level behaviour of the drop-in ConvVAE on the HIP path: uint8 frame tables, the in-kernel noise stream, workspace guards.
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


@pytest.mark.gpu
def test_adam_launch_writes_the_fragment_ordered_weight_copies(tmp_path):
    """Round 5: the optimiser launch (mi_adam_tf_layouts_frag) also emits the six fragment-ordered bf16 kernel copies of the activation-resident convolutions -- 16-byte granules
    of the tile it holds in LDS -- instead of a separate ares_pack launch at the head of the next step.  After two SGD steps every copy in the engine's workspace equals, bit for
    bit, what mi_ares_pack_weights writes from the current master kernel (conv4: conv form + gather form, deconv1: gather form + conv form, conv3 / deconv2: the mid gather form)."""
    import ctypes
    from mi355 import lib as milib
    L = milib.get()
    m = make(tmp_path, "bf16", params=trained_like_params(6), seed=0)
    frames = synth_frames(16, seed=21)
    eps = np.random.RandomState(2).standard_normal((16, 64)).astype(np.float32)
    for _ in range(2):
        m.train_step(frames, frames, eps=eps)
    torch.cuda.synchronize()
    dev = m.dev
    st = torch.cuda.current_stream().cuda_stream
    nbytes = int(L.mi_ares_weight_bytes())
    lay = dev.layout
    jobs = [(6, "vae/encoder/conv4/kernel", 0, nbytes), (7, "vae/encoder/conv4/kernel", 1, nbytes), (8, "vae/decoder/deconv1/kernel", 1, nbytes), (9, "vae/decoder/deconv1/kernel", 0, nbytes),
            (10, "vae/encoder/conv3/kernel", 2, nbytes // 4), (11, "vae/decoder/deconv2/kernel", 2, nbytes // 4),
            # round 6: the conv-form copies of the same two kernels for the register-weight kernel (conv3 forward, deconv2's input gradient): form 3
            (12, "vae/encoder/conv3/kernel", 3, nbytes // 4), (13, "vae/decoder/deconv2/kernel", 3, nbytes // 4),
            # ... deconv3's kernel for the gather-form register-weight kernel (form 4, 144 fragments) and conv2's for the fused encoder head (form 5, 64 fragments)
            (14, "vae/decoder/deconv3/kernel", 4, 144 * 1024), (15, "vae/encoder/conv2/kernel", 5, 64 * 1024),
            (16, "vae/decoder/deconv3/kernel", 6, 100 * 1024)]                                # ... and deconv3's again for its input gradient (conv form, k = 5: 2 x 50 fragments)
    base = dev.workspace.data_ptr()
    scratch4 = [torch.zeros(nbytes, device="cuda", dtype=torch.uint8) for _ in range(4)]
    for which, name, form, nb in jobs:
        addr = L.mi_vae_buffer(dev.handle, which)
        assert addr, which
        got = dev.workspace[addr - base:addr - base + nb].clone()
        want = torch.zeros(nbytes, device="cuda", dtype=torch.uint8)
        off = lay[name][0]
        if form in (4, 5, 6):
            L.mi_ares_pack_weights(st, form, dev.params.data_ptr() + 4 * off, want.data_ptr())
            if form == 4:                                    # (fragments of taps past the 5 x 5 kernel are never written by the optimiser launch and never read by the kernel: compare the live ones)
                live = torch.zeros(144, dtype=torch.bool)
                for cls in range(4):
                    for tap in range(9):
                        kh, kw = (cls >> 1) + 2 * (2 - tap // 3), (cls & 1) + 2 * (2 - tap % 3)
                        live[(cls * 9 + tap) * 4:(cls * 9 + tap) * 4 + 4] = kh < 5 and kw < 5
                m8 = live.repeat_interleave(1024).cuda()
                got, want = got[:nb][m8], want[:nb][m8]
                nb = int(m8.sum())
        elif form == 3:
            src = dev.params.data_ptr() + 4 * off
            L.mi_ares_pack_weights8(st, dev.params.data_ptr() + 4 * lay["vae/encoder/conv4/kernel"][0], dev.params.data_ptr() + 4 * lay["vae/decoder/deconv1/kernel"][0],
                                    src if which == 12 else None, src if which == 13 else None, scratch4[0].data_ptr(), scratch4[1].data_ptr(), scratch4[2].data_ptr(), scratch4[3].data_ptr(),
                                    None, None, want.data_ptr() if which == 12 else None, want.data_ptr() if which == 13 else None)
        else:
            L.mi_ares_pack_weights(st, form, dev.params.data_ptr() + 4 * off, want.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(got, want[:nb]), (which, name, form)


@pytest.mark.gpu
def test_evaluate_and_train_one_epoch_write_the_three_merge_summary_scalars(tmp_path):
    """vae/models.py:147-151,218,230: the reference's merge_summary holds kl_loss, reconstruction_loss AND learning_rate, and both train_one_epoch and evaluate write all of
    it -- the val log carries the (decayed, logged-only) learning rate as well (VERDICT r05 missing 5)."""
    import glob
    from mi355 import summary as sm
    m = ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=str(tmp_path), precision="bf16", learning_rate=2e-4, lr_decay=0.5, seed=0)
    m.init_session(init_logging=True)
    frames = synth_frames(16, seed=3)
    np.random.seed(0)
    m.evaluate(frames, frames, 8)
    m.train_one_epoch(frames, frames, 8)
    m.evaluate(frames, frames, 8)
    for w in (m.train_writer, m.val_writer):
        w.flush()
    for sub, steps in (("val", [0, 1]), ("train", [0])):
        (path,) = glob.glob(os.path.join(m.log_dir, sub, "events.out.tfevents.*"))
        ver, series = sm.read_events(path, verify=True)
        assert set(series) == {"vae/kl_loss", "vae/reconstruction_loss", "vae/learning_rate"}, (sub, set(series))
        assert [p[0] for p in series["vae/learning_rate"]] == steps
        assert [p[2] for p in series["vae/learning_rate"]] == pytest.approx([2e-4 * 0.5 ** s for s in steps], rel=1e-6)


@pytest.mark.gpu
def test_one_channel_frames_take_the_layer_kernels_not_the_fused_encoder_head(tmp_path):
    """the fused conv1 + conv2 kernel of the forward pass (enc12_tile.hpp) is written for 3-channel frames (frame stride FH FW 3, conv1's kernel as [32][48]);
    a bf16 engine on 80 x 160 x 1 frames must fall back to the layer kernels (run_encoder's gate now checks c[0] == 3) instead of computing on a wrong stride and reading past
    the frame table.  The bf16 engine's posterior means and losses on one-channel frames agree with the exact-fp32 engine's to bf16 accuracy, and a training step runs."""
    p = vo.init_vae_params(0, 64, (80, 160, 1), (80, 160, 1))
    rng = np.random.RandomState(1)
    for k in p:
        if k.endswith("bias"):
            p[k] = (0.05 * rng.standard_normal(p[k].shape)).astype(np.float32)
    frames = np.random.RandomState(5).randint(0, 256, (24, 80, 160, 1), dtype=np.uint8).astype(np.float32) / 255.0
    eps = np.random.RandomState(6).standard_normal((24, 64)).astype(np.float32)
    out = {}
    for precision in ("bf16", "fp32"):
        m = ConvVAE(np.array([80, 160, 1]), z_dim=64, model_dir=str(tmp_path / precision), precision=precision, seed=0)
        m.set_weights(p)
        m.init_session(init_logging=False)
        z = m.encode(frames)
        losses = m.train_step(frames, frames, eps=eps)
        out[precision] = (z, losses)
    assert np.isfinite(out["bf16"][0]).all()
    assert rel_err(out["bf16"][0], out["fp32"][0]) < 2e-2, rel_err(out["bf16"][0], out["fp32"][0])
    assert out["bf16"][1] == pytest.approx(out["fp32"][1], rel=2e-3)
