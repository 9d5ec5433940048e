"""Model-level parity of the drop-in PPO (HIP path through the C ABI) vs the CPU oracle; fp32 throughout,
tolerance 1e-4 relative on losses, 2e-4 of each tensor's max on gradients; GAE bit-exact (fp64)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ppo_oracle as po  # noqa: E402
from ppo import PPO  # noqa: E402
import utils  # noqa: E402


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def make_pair(tmp_path, seed=2, **kw):
    space = po.ActionSpace()
    hp = dict(learning_rate=1e-4, lr_decay=1.0, epsilon=0.2, value_scale=1.0, entropy_scale=0.01, initial_std=1.0)
    hp.update(kw)
    o = po.OraclePPO([67], space, seed=seed, **hp)
    m = PPO(np.array([67]), space, model_dir=str(tmp_path), seed=seed, **hp)
    m.set_weights(o.params)
    m.init_session(init_logging=False)
    return o, m


def synth_rollout(T=128, seed=7):
    rng = np.random.RandomState(seed)
    states = (0.5 * rng.standard_normal((T, 67))).astype(np.float32)
    states[:, 64] = rng.uniform(-1, 1, T); states[:, 65] = rng.uniform(0, 1, T); states[:, 66] = rng.uniform(0, 30, T)
    rewards = list(rng.uniform(0, 1, T))
    dones = [False] * T
    return states, rewards, dones, rng


def test_minibatch_gradients_and_losses(tmp_path):
    o, m = make_pair(tmp_path)
    rng = np.random.RandomState(5)
    for k in o.params:                                   # theta != theta_old: ratio != 1, some samples clipped
        o.params[k] = o.params[k] + (0.02 * rng.standard_normal(o.params[k].shape)).astype(np.float32)
    m.set_weights(o.params)
    s = (0.5 * rng.standard_normal((32, 67))).astype(np.float32)
    a = rng.uniform(-1, 1, (32, 2)).astype(np.float32)
    R, A = rng.randn(32).astype(np.float32), rng.randn(32).astype(np.float32)
    scal, grads = o.loss_and_grads(s, a, R, A)
    d = m.dev
    d.forward_backward(m._to_dev(s, (32, 67)), m._to_dev(a, (32, 2)), m._to_dev(R, (32,)), m._to_dev(A, (32,)), 32, 1 / 32.0, 1.0)
    L = d.losses.cpu().numpy()
    for got, key in zip(L, ("policy_loss", "value_loss", "entropy_loss", "loss", "ratio_mean")):
        assert got == pytest.approx(scal[key], rel=1e-4, abs=1e-6), key
    g = d.export_grads()
    bad = {k: rel_err(g[k], grads[k]) for k in grads if rel_err(g[k], grads[k]) > 2e-4}
    assert not bad, bad


@pytest.mark.parametrize("epochs", [4, 3])
def test_config3_full_update_matches_oracle(tmp_path, epochs):
    """BASELINE config 3: horizon 128, minibatch 32, `epochs` epochs; the trainer's hot section (train.py:171-207)
    driven through the drop-in API vs the oracle, same legacy-numpy permutations."""
    o, m = make_pair(tmp_path)
    states, rewards, dones, rng = synth_rollout()
    noise = rng.standard_normal((128, 2)).astype(np.float32)
    acts, vals = o.predict(states, noise=noise)
    acts_g, vals_g = m.predict(states, noise=noise)
    assert rel_err(acts_g, acts) < 1e-4 and rel_err(vals_g, vals) < 1e-4
    _, last = o.predict(states[-1], greedy=True)
    values = [np.float32(v) for v in vals]
    np.random.seed(0)
    logs, ret_o, adv_o = po.ppo_update(o, list(states), list(acts), values, rewards, dones, last, 0.99, 0.95, epochs, 32)
    # --- product path: the reference trainer's lines restated with the drop-in modules ---
    np.random.seed(0)
    advantages = utils.compute_gae(rewards, values, last, dones, 0.99, 0.95)
    assert np.array_equal(advantages, po.compute_gae(rewards, values, last, dones, 0.99, 0.95))        # bit-exact fp64
    returns = advantages + values
    advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
    assert np.array_equal(returns, ret_o)
    s_arr, a_arr = np.array(states), np.array(acts)
    m.update_old_policy()
    for _ in range(epochs):
        indices = np.arange(128)
        np.random.shuffle(indices)
        for i in range(int(np.ceil(128 / 32))):
            mb = indices[i * 32:(i + 1) * 32]
            m.train(s_arr[mb], a_arr[mb], returns[mb], advantages[mb])
    assert m.get_train_step_idx() == 4 * epochs
    got = m.dev.export_params()
    p0 = make_pair(tmp_path)[0].params
    # Where the two fp32 implementations differ after 16 Adam steps comes from fp32 rounding in BOTH of them (early Adam is sign-like: a last-bit
    # gradient difference near zero moves a weight by up to 2 lr).  So both are measured against the SAME update run in float64 (losses, gradients and
    # the Adam recurrence, identical minibatches): the device path must be no further from that exact trajectory than the fp32 oracle is -- factor 2
    # on the per-tensor RMS of the update error, floor 0.02 % of the update scale (the fp32 oracle itself: ~4e-6) -- and likewise for the last minibatch loss.
    import torch
    from collections import OrderedDict
    ex = OrderedDict((k, v.astype(np.float64)) for k, v in p0.items())
    adam64 = po.AdamTF(OrderedDict((k, v.shape) for k, v in ex.items()), dtype=np.float64)
    ex_old = {k.replace("policy/", "policy_old/", 1): torch.from_numpy(v.copy()) for k, v in ex.items()}      # update_old_policy() before the first step
    t64 = lambda a: torch.from_numpy(np.asarray(a, np.float32).astype(np.float64))                            # noqa: E731  (inputs are rounded to f32 at the feed)
    np.random.seed(0)
    loss64 = None
    for mb in po.minibatch_schedule(128, 32, epochs):
        pt = OrderedDict((k, torch.from_numpy(v.copy()).requires_grad_(True)) for k, v in ex.items())
        L64 = po.ppo_losses(pt, ex_old, t64(s_arr[mb]), t64(a_arr[mb]), t64(returns[mb]), t64(advantages[mb]), o.low, o.high, 0.2, 1.0, 0.01)
        L64["loss"].backward()
        adam64.step(ex, {k: v.grad.numpy() for k, v in pt.items()}, 1e-4)
        loss64 = float(L64["loss"].detach())
    rows = []
    for k, v in o.params.items():
        upd_x = ex[k] - p0[k].astype(np.float64)
        e_o = np.sqrt(np.mean(((v.astype(np.float64) - p0[k]) - upd_x) ** 2))
        e_d = np.sqrt(np.mean(((got[k].astype(np.float64) - p0[k]) - upd_x) ** 2))
        scale = max(np.abs(upd_x).max(), 1e-12)
        rows.append((k, e_d / scale, e_o / scale))
        assert e_d <= 2.0 * e_o + 2e-4 * scale, (k, e_d / scale, e_o / scale)
    print("\nPPO 16-step update, RMS update error / max |update| vs the float64 trajectory (device, fp32 oracle):")
    for r in rows:
        print("  %-34s %.3e  %.3e" % r)
    last_losses = m.dev.losses.cpu().numpy()
    assert abs(float(last_losses[3]) - loss64) <= 2.0 * abs(logs[-1]["loss"] - loss64) + 1e-5 * max(1.0, abs(loss64)), (float(last_losses[3]), logs[-1]["loss"], loss64)


def test_partial_minibatch_and_predict_shapes(tmp_path):
    o, m = make_pair(tmp_path, seed=3)
    rng = np.random.RandomState(0)
    s = (0.5 * rng.standard_normal((5, 67))).astype(np.float32)
    a = rng.uniform(-1, 1, (5, 2)).astype(np.float32)
    R, A = rng.randn(5), rng.randn(5)                       # float64 inputs are rounded to f32 at the feed
    m.update_old_policy(); o.update_old_policy()
    so = o.train(s, a, R, A)
    sg = m.train_step(s, a, R, A)
    assert sg["loss"] == pytest.approx(so["loss"], rel=1e-4, abs=1e-6) and sg["prob_ratio"] == pytest.approx(1.0, abs=1e-5)
    act, val = m.predict(s[0], greedy=True)
    ao, vo_ = o.predict(s[0], greedy=True)
    assert act.shape == (2,) and np.ndim(val) == 0 and rel_err(act, ao) < 1e-3 and abs(val - vo_) < 1e-4 * max(1, abs(vo_))
    act2, _ = m.predict(s[0])
    assert (act2 >= np.array([-1, 0]) - 1e-6).all() and (act2 <= np.array([1, 1]) + 1e-6).all()
    m.write_episodic_summaries()
    assert m.get_episode_idx() == 1


@pytest.mark.parametrize("fmt", ["npz", "tf"])
def test_checkpoint_roundtrip_and_tf_names(tmp_path, golden_dir, fmt, monkeypatch):
    monkeypatch.setenv("MI355_CKPT_FORMAT", fmt)
    ref = json.load(open(os.path.join(golden_dir, "ref_variables.json")))["ppo_agent"]
    ref = {k: v for k, v in ref.items() if not k.startswith("vae/")}       # the agent checkpoint also carries the VAE graph
    o, m = make_pair(tmp_path / "a")
    rng = np.random.RandomState(1)
    s, a = (0.5 * rng.standard_normal((8, 67))).astype(np.float32), rng.uniform(-1, 1, (8, 2)).astype(np.float32)
    R, A = rng.randn(8).astype(np.float32), rng.randn(8).astype(np.float32)
    m.update_old_policy()
    m.train(s, a, R, A)
    m.write_episodic_summaries()
    sd = m.state_dict()
    assert {k: list(np.shape(v)) for k, v in sd.items()} == {k: v["shape"] for k, v in ref.items()}
    m.save()
    if fmt == "tf":
        from mi355 import tf_bundle as tb
        ours, _ = tb.read_index(os.path.join(m.checkpoint_dir, "model.ckpt-1.index"))
        theirs, _ = tb.read_index(os.path.join(golden_dir, "ref_index", "ppo_agent.index"))
        theirs = {k: e for k, e in theirs.items() if not k.startswith("vae/")}
        assert {k: (e["dtype"], e["shape"], e["size"]) for k, e in ours.items()} == {k: (e["dtype"], e["shape"], e["size"]) for k, e in theirs.items()}
    m2 = PPO(np.array([67]), po.ActionSpace(), model_dir=str(tmp_path / "a"), learning_rate=1e-4, lr_decay=1.0, value_scale=1.0, initial_std=1.0)
    m2.init_session(init_logging=False)
    assert m2.load_latest_checkpoint() is True and m2.get_episode_idx() == 1 and m2.get_train_step_idx() == 1
    l1, l2 = m.train_step(s, a, R, A), m2.train_step(s, a, R, A)
    assert l1["loss"] == pytest.approx(l2["loss"], rel=1e-6)


def test_gae_batched_rows_bit_exact_and_normalised(tmp_path):
    rng = np.random.RandomState(3)
    R, T = 33, 128
    rew = rng.uniform(0, 1, (R, T))
    val = rng.randn(R, T + 1).astype(np.float32)
    done = np.zeros((R, T)); done[::4, -1] = 1
    raw, ret, adv = utils.compute_gae_batched(rew, val, done, 0.99, 0.95, normalize=True)
    for r in range(R):
        ref = po.compute_gae(list(rew[r]), list(val[r, :T]), val[r, T], list(done[r].astype(bool)), 0.99, 0.95)
        assert np.array_equal(raw[r], ref)
        rr, aa = po.returns_and_normalized_advantages(ref.copy(), val[r, :T].astype(np.float64))
        assert np.array_equal(ret[r], rr) and np.allclose(adv[r], aa, rtol=1e-12, atol=1e-12)
    assert utils.compute_gae([], [], 0.0, [], 0.99, 0.95).shape == (0,)


@pytest.mark.parametrize("M", [32, 77, 256, 2048])
def test_fused_step_gradients_update_and_cache_match_oracle(tmp_path, M):
    """The PPO minibatch step at the reference's minibatch (32), at a ragged size, at the largest fused size (256) and at the synthetic
    replay's per-GPU minibatch (2048: above 256 rows the tiled fp32 MFMA kernels run instead of the fused ones).  Against the oracle:
    (1) the gradient form (data-parallel path: gradients written, then all-reduce + Adam): five loss scalars 1e-4, all 13 gradients 2e-4 of
    the tensor max; (2) the single-rank form (one C call, Adam applied by the blocks that produce each gradient tile): parameters after the
    step equal the oracle's TF-Adam update on ITS gradients wherever the gradient is not at the 1e-8 epsilon scale; (3) with the cached
    log pi_old (mi_ppo_logp_old, the old policy's forward pass skipped) the step is identical to (2)."""
    o, m = make_pair(tmp_path)
    rng = np.random.RandomState(5 + M)
    for k in o.params:                                   # theta != theta_old: ratio != 1, some samples clipped
        o.params[k] = o.params[k] + (0.02 * rng.standard_normal(o.params[k].shape)).astype(np.float32)
    m.dev.load_params(o.params)                          # theta only; theta_old stays the initial copy (as after update_old_policy + some steps)
    s = (0.5 * rng.standard_normal((M, 67))).astype(np.float32)
    a = np.stack([rng.uniform(-1, 1, M), rng.uniform(0, 1, M)], axis=1).astype(np.float32)
    R, A = rng.randn(M).astype(np.float32), rng.randn(M).astype(np.float32)
    scal, grads = o.loss_and_grads(s, a, R, A)
    d = m.dev
    sd, ad, Rd, Ad = m._to_dev(s, (M, 67)), m._to_dev(a, (M, 2)), m._to_dev(R, (M,)), m._to_dev(A, (M,))
    d.forward_backward(sd, ad, Rd, Ad, M, 1.0 / M, 1.0)
    L = d.losses.cpu().numpy()
    for got, key in zip(L, ("policy_loss", "value_loss", "entropy_loss", "loss", "ratio_mean")):
        assert got == pytest.approx(scal[key], rel=1e-4, abs=1e-6), (key, got, scal[key])
    g = d.export_grads()
    bad = {k: rel_err(g[k], grads[k]) for k in grads if rel_err(g[k], grads[k]) > 2e-4}
    assert not bad, bad
    # two runs of the gradient pass are bitwise equal at every size: one wave per tile over the whole minibatch (M <= 256), per-chunk slabs added in a fixed order above
    # (round 4: the row chunks of the large-minibatch form met in fp32 atomics before); garbage in the gradient buffer beforehand (the pass stores)
    first = d.grads.clone()
    d.grads.fill_(4.25)
    d.forward_backward(sd, ad, Rd, Ad, M, 1.0 / M, 1.0)
    import torch
    used = torch.zeros_like(first, dtype=torch.bool)
    for name, (o_, s_) in d.layout.items():
        used[o_:o_ + s_] = True
    assert torch.equal(d.grads[used], first[used])
    d.grads.zero_()
    # (2) one-call step vs the oracle's Adam on its own gradients
    from oracle import vae_oracle as vo
    before = {k: v.copy() for k, v in o.params.items()}
    want = {k: v.copy() for k, v in before.items()}
    adam = vo.AdamTF({k: v.shape for k, v in before.items()})
    adam.step(want, grads, 1e-4)
    from ppo import _adam_alpha
    alpha = _adam_alpha(1e-4, 0.9, 0.999)
    d.train_step(sd, ad, Rd, Ad, M, 1.0 / M, 1.0, alpha)
    got = d.export_params()
    for k in want:
        sig = np.abs(grads[k]) > 1e-6                     # where Adam's first step is +-lr regardless of rounding
        assert np.allclose(got[k][sig], want[k][sig], rtol=0, atol=2e-6), k
        assert np.abs(got[k] - before[k]).max() <= 1.01e-4 + 1e-9
    if M <= 256:
        L2 = d.losses.cpu().numpy()
        assert np.allclose(L2[:5], L[:5], rtol=1e-6, atol=1e-7)
        assert np.allclose(L2[5:7], d.action_mean[:M].cpu().numpy().mean(0), rtol=1e-5)                                  # mean over the minibatch of action_mean
        assert np.allclose(L2[7:9], np.exp(before["policy/action_logstd"]), rtol=1e-6)                                     # std = exp(logstd) before the step
        # (3) cached log pi_old: same update from the same starting point
        import torch
        d.load_params(before)
        d.adam_m.zero_(); d.adam_v.zero_()
        lp = torch.empty(M, device=d.device)
        d.logp_old(sd, ad, M, lp)
        d.train_step(sd, ad, Rd, Ad, M, 1.0 / M, 1.0, alpha, logp_old=lp)
        got2 = d.export_params()
        for k in got:
            assert np.allclose(got2[k], got[k], rtol=0, atol=1e-7), k


def test_rollout_step_one_call_matches_encode_then_predict(tmp_path):
    """mi_rollout_step (raw uint8 frame + measurements -> action, value, z in one call) against the oracle's
    encode([frame / 255])[0] -> np.append(z, measurements) -> predict(state), fp32, 1e-4; sampled and greedy; and against the drop-in's own
    two-call path (VAE.encode + PPO.predict)."""
    from oracle import vae_oracle as vo
    from rollout import RolloutStep
    from vae.models import ConvVAE
    rng = np.random.RandomState(21)
    vparams = vo.init_vae_params(3)
    for k in vparams:
        if k.endswith("bias"):
            vparams[k] = (0.05 * rng.standard_normal(vparams[k].shape)).astype(np.float32)
    for precision in ("fp32", "bf16"):                    # the rollout path computes in exact fp32 on the master weights either way
        vae = ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=str(tmp_path / ("vae_" + precision)), precision=precision, training=False)
        vae.set_weights(vparams)
        vae.init_session(init_logging=False)
        o, m = make_pair(tmp_path / ("ppo_" + precision))
        ovae = vo.OracleVAE(params=vparams, training=False)
        for i in range(3):
            step = RolloutStep(vae, m, io="device" if i == 1 else "pinned")
            frame = rng.randint(0, 256, (80, 160, 3), dtype=np.uint8)
            meas = [float(rng.uniform(-1, 1)), float(rng.uniform(0, 1)), float(rng.uniform(0, 30))]
            noise = rng.standard_normal(2).astype(np.float32)
            z_o = ovae.encode([frame.astype(np.float32) / 255.0])[0]
            state_o = np.append(z_o, meas)
            for greedy in (False, True):
                a_o, v_o = o.predict(state_o, greedy=greedy, noise=None if greedy else noise[None])
                a, v, state = step(frame, meas, greedy=greedy, noise=noise)
                assert state.shape == (67,) and state.dtype == np.float64 and np.array_equal(state[64:], np.asarray(meas))
                assert rel_err(state[:64], z_o) < 1e-4, (precision, i, rel_err(state[:64], z_o))
                assert np.allclose(a, np.asarray(a_o).reshape(-1), rtol=1e-4, atol=1e-5) and v == pytest.approx(float(np.asarray(v_o).reshape(-1)[0]), rel=1e-4, abs=1e-5)
        if precision == "fp32":                           # the two-call surface of the same engines
            z2 = vae.encode([frame.astype(np.float32) / 255.0])[0]
            a2, v2 = m.predict(np.append(z2, meas), greedy=True)
            a, v, state = step(frame, meas, greedy=True)
            assert rel_err(state[:64], z2) < 1e-5 and np.allclose(a, a2, atol=1e-5) and v == pytest.approx(float(v2), rel=1e-5, abs=1e-6)


@pytest.mark.parametrize("M", [32, 77, 300])
def test_step_with_fused_minibatch_gather_equals_gathered_step(tmp_path, M):
    """mi_ppo_train_step_idx (the reference's `states[mb_idx]` ... `advantages[mb_idx]` fancy-indexing of train.py:199-204 done inside the step's kernels)
    against mi_ppo_train_step on the rows gathered beforehand: same parameters after the step (bit for bit at M <= 256, where both run the same
    summation order; M > 256 meets in fp32 atomics: 1e-6), same loss scalars -- with and without the cached log pi_old."""
    import torch
    T = 512
    rng = np.random.RandomState(M)
    s = (0.5 * rng.standard_normal((T, 67))).astype(np.float32)
    a = rng.uniform(-1, 1, (T, 2)).astype(np.float32)
    R, A = rng.randn(T).astype(np.float32), rng.randn(T).astype(np.float32)
    rows = rng.permutation(T)[:M].astype(np.int32)
    outs = []
    for fused in (False, True):
        for cached in (False, True):
            o, m = make_pair(tmp_path / ("f%d%d" % (fused, cached)))
            for k in o.params:
                o.params[k] = o.params[k] + (0.02 * np.random.RandomState(3).standard_normal(o.params[k].shape)).astype(np.float32)
            m.set_weights(o.params)                      # theta != theta_old: ratio != 1
            d = m.dev
            d.ensure_batch(T)
            sd, ad, Rd, Ad = (m._to_dev(x, x.shape) for x in (s, a, R, A))
            lp = None
            if cached:
                lp = torch.empty(T, device=d.device)
                d.logp_old(sd, ad, T, lp)
            rd = torch.from_numpy(rows).to(d.device)
            alpha = 1e-4
            if fused:
                d.train_step_idx(sd, ad, Rd, Ad, lp, rd, M, 1.0 / M, 1.0, alpha)
            else:
                mb = rd.long()
                d.train_step(sd[mb].contiguous(), ad[mb].contiguous(), Rd[mb].contiguous(), Ad[mb].contiguous(), M, 1.0 / M, 1.0, alpha,
                             logp_old=None if lp is None else lp[mb].contiguous())
            outs.append((fused, cached, d.params.cpu().numpy().copy(), d.losses.cpu().numpy()[:5].copy()))
    ref = {c: (p, l) for f, c, p, l in outs if not f}
    for f, c, p, l in outs:
        if not f:
            continue
        p0, l0 = ref[c]
        if M <= 256:
            assert np.array_equal(p, p0), (M, c, float(np.abs(p - p0).max()))
        else:
            assert np.abs(p - p0).max() <= 1e-6 * max(1.0, np.abs(p0).max()), (M, c)
        assert np.allclose(l, l0, rtol=1e-6, atol=1e-7), (M, c, l, l0)


@pytest.mark.parametrize("hidden", [(64, 64), (96, 32)])
def test_fused_gather_with_narrow_hidden_layers(hidden):
    """with H1 < 128 the layer-1 waves whose column tile lies past H1 return early -- the gathered minibatch they leave for the layer-1 filter
    gradient must be complete all the same.  mi_ppo_train_step_idx against mi_ppo_train_step on host-gathered rows for hidden sizes the reference does not
    use: identical parameters after the step (same kernels, same summation order), and out-of-range row values are clamped instead of read past the tables."""
    import torch
    from mi355.ppo_device import PpoDevice
    from mi355.init import init_ppo
    T, M = 256, 48
    rng = np.random.RandomState(5)
    s = (0.5 * rng.standard_normal((T, 67))).astype(np.float32)
    a = rng.uniform(-1, 1, (T, 2)).astype(np.float32)
    R, A = rng.randn(T).astype(np.float32), rng.randn(T).astype(np.float32)
    rows = rng.permutation(T)[:M].astype(np.int32)
    res = []
    for fused in (False, True):
        d = PpoDevice(67, 2, [-1.0, 0.0], [1.0, 1.0], 0.2, 1.0, 0.01, hidden=hidden, max_batch=T)
        assert d.fused_ok()
        vals = init_ppo(3, 67, 2, 1.0, hidden=hidden)
        for i, k in enumerate(vals):
            if k.endswith("bias"):
                vals[k] = (0.05 * np.random.RandomState(i).standard_normal(vals[k].shape)).astype(np.float32)
        old = {k.replace("policy/", "policy_old/", 1): (v + 0.02 * np.random.RandomState(9).standard_normal(v.shape)).astype(np.float32) for k, v in vals.items()}
        d.load_params(vals, old)
        dev = d.device
        sd, ad, Rd, Ad = (torch.from_numpy(x).to(dev) for x in (s, a, R, A))
        rd = torch.from_numpy(rows).to(dev)
        if fused:
            d.train_step_idx(sd, ad, Rd, Ad, None, rd, M, 1.0 / M, 1.0, 1e-4)
        else:
            mb = rd.long()
            d.train_step(sd[mb].contiguous(), ad[mb].contiguous(), Rd[mb].contiguous(), Ad[mb].contiguous(), M, 1.0 / M, 1.0, 1e-4)
        res.append((d.params.cpu().numpy().copy(), d.losses.cpu().numpy()[:5].copy()))
        if fused:                                           # rows outside the table: clamped (row -7 -> 0, row T + 5 -> T - 1), no fault, finite result
            bad_rows = rows.copy(); bad_rows[0] = -7; bad_rows[1] = T + 5
            d.train_step_idx(sd, ad, Rd, Ad, None, torch.from_numpy(bad_rows).to(dev), M, 1.0 / M, 1.0, 1e-4)
            torch.cuda.synchronize()
            assert np.isfinite(d.params.cpu().numpy()).all()
    (p0, l0), (p1, l1) = res
    assert np.array_equal(p0, p1), float(np.abs(p0 - p1).max())
    assert np.allclose(l0, l1, rtol=1e-6, atol=1e-7)
    assert not np.array_equal(p0[:64], np.zeros(64, np.float32))


@pytest.mark.parametrize("M", [32, 77])
def test_mlp_policy_op_level_forward_backward(M):
    """SURVEY 8b names `mi_mlp_policy_fwd / _bwd`: the build_mlp trunk (utils.py:25-28; dense 500 ReLU, dense 300 ReLU: ppo.py:42-44) as op-level calls of the C ABI, against
    the float64 statement of the same two layers: activations 1e-5 of max, the four gradient tensors 2e-5 of max, accumulated into non-zero buffers; bad shapes are refused."""
    import torch
    from mi355 import lib as milib
    L = milib.get()
    rng = np.random.RandomState(M)
    din, H1, H2 = 68, 500, 300                                # the state padded to 68 columns as in the engine (column 67 is zero)
    x = (0.5 * rng.standard_normal((M, din))).astype(np.float32); x[:, 67] = 0
    W1 = (rng.standard_normal((din, H1)) / np.sqrt(din)).astype(np.float32); b1 = (0.1 * rng.standard_normal(H1)).astype(np.float32)
    W2 = (rng.standard_normal((H1, H2)) / np.sqrt(H1)).astype(np.float32); b2 = (0.1 * rng.standard_normal(H2)).astype(np.float32)
    g2 = rng.standard_normal((M, H2)).astype(np.float32)
    d = lambda a: torch.from_numpy(a).cuda()                  # noqa: E731
    xd, W1d, b1d, W2d, b2d, g2d = d(x), d(W1), d(b1), d(W2), d(b2), d(g2)
    h1, h2 = torch.empty(M, H1, device="cuda"), torch.empty(M, H2, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    L.mi_mlp_policy_fwd(st, xd.data_ptr(), M, din, W1d.data_ptr(), b1d.data_ptr(), H1, W2d.data_ptr(), b2d.data_ptr(), H2, h1.data_ptr(), h2.data_ptr())
    t = lambda a: torch.from_numpy(np.asarray(a, np.float64)).requires_grad_(True)      # noqa: E731
    x64, W164, b164, W264, b264 = t(x), t(W1), t(b1), t(W2), t(b2)
    r1 = torch.relu(x64 @ W164 + b164); r2 = torch.relu(r1 @ W264 + b264)
    (r2 * torch.from_numpy(g2.astype(np.float64))).sum().backward()
    assert rel_err(h1.cpu().numpy(), r1.detach().numpy()) < 1e-5 and rel_err(h2.cpu().numpy(), r2.detach().numpy()) < 1e-5
    dW1, db1 = torch.full((din, H1), 0.5, device="cuda"), torch.full((H1,), -1.0, device="cuda")
    dW2, db2 = torch.full((H1, H2), 0.25, device="cuda"), torch.full((H2,), 2.0, device="cuda")
    scratch = torch.empty(M * (H1 + H2), device="cuda")
    L.mi_mlp_policy_bwd(st, xd.data_ptr(), M, din, W2d.data_ptr(), H1, H2, h1.data_ptr(), h2.data_ptr(), g2d.data_ptr(), dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(), scratch.data_ptr())
    torch.cuda.synchronize()
    for got, off, ref, name in ((dW1, 0.5, W164.grad, "dW1"), (db1, -1.0, b164.grad, "db1"), (dW2, 0.25, W264.grad, "dW2"), (db2, 2.0, b264.grad, "db2")):
        assert rel_err(got.cpu().numpy() - off, ref.numpy()) < 2e-5, name
    with pytest.raises(milib.MiError):
        L.mi_mlp_policy_fwd(st, xd.data_ptr(), M, 67, W1d.data_ptr(), b1d.data_ptr(), H1, W2d.data_ptr(), b2d.data_ptr(), H2, h1.data_ptr(), h2.data_ptr())
