"""BASELINE configs[0] (train_vae.py's plumbing case: 1k frames, batch 32, evaluate + train_one_epoch) and the rest of the ConvVAE call surface
(encode / reconstruct / generate, loss variants, checkpoints, SURVEY 8 row a22: vae_common.py's load_vae / encode-state chain) against the CPU oracle.
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
