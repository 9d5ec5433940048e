"""The HIP path (through the C ABI, fp32 mode) against the reference's OWN serialized TensorFlow graphs, with no restatement in
between: tests/golden/ref_graph_*.json.gz (decoded from the MetaGraphDef files the reference ships; tests/golden/make_graph_fixture.py)
executed in float64 by oracle/tf_graph.py on the same seeded inputs.

Tolerances: the fp32 engine against the float64 graph -- losses 1e-4 relative, outputs 1e-4 of the tensor's max, every gradient
tensor 2e-4 of its max; after three Adam steps the parameter UPDATES agree within 2 % of one step size on > 99.9 % of the entries
(Adam's first steps are sign-like, so entries whose gradient is ~0 legitimately differ)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ppo_oracle as po  # noqa: E402
from oracle import vae_oracle as vo  # noqa: E402
from ppo import PPO  # noqa: E402
from ref_graph_helpers import PPO_EPS, VAE_EPS, adam_nodes, init_variables, load_graph  # noqa: E402
from vae.models import ConvVAE  # noqa: E402


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("which,target_c", [("vae_rgb", 3), ("vae_seg", 1)])
def test_hip_vae_step_matches_reference_graph(tmp_path, which, target_c):
    g, _ = load_graph(which)
    params = vo.init_vae_params(11, 64, (80, 160, 3), (80, 160, target_c))
    rng = np.random.RandomState(12)
    for k in params:
        if k.endswith("bias"):
            params[k] = (0.05 * rng.standard_normal(params[k].shape)).astype(np.float32)
    init_variables(g, params)
    B = 4
    src = (rng.randint(0, 256, (B, 80, 160, 3)) / 255.0).astype(np.float32)
    tgt = src if target_c == 3 else (rng.randint(0, 13, (B, 80, 160, 1)) / 12.0).astype(np.float32)
    eps = rng.standard_normal((B, 64)).astype(np.float32)
    feed = {"vae/source_state_placeholder": src, "vae/target_state_placeholder": tgt, VAE_EPS: eps[None]}
    nodes = adam_nodes(g)
    fetched = g.run(["vae/mean/BiasAdd", "vae/reconstructed_states", "vae/Mean_1", "vae/Mean_2"] + [grad for *_, grad in nodes], feed)
    mean, rec, recon, kl = fetched[:4]
    grads = {var: x for (_, var, *_), x in zip(nodes, fetched[4:])}

    m = ConvVAE(np.array([80, 160, 3]), np.array([80, 160, target_c]), z_dim=64, model_dir=str(tmp_path), precision="fp32")
    m.set_weights(params)
    m.init_session(init_logging=False)
    assert rel_err(m.encode(src), mean) < 1e-4
    if target_c == 3:                                           # reconstruct() reshapes with the SOURCE shape, as the reference does
        assert rel_err(np.stack([r.reshape(-1) for r in m.reconstruct(src, eps=eps)]), rec) < 1e-4
    s = m._frames(src, 38400, "src")
    t = s if target_c == 3 else m._frames(tgt, m.dev.P, "tgt")
    e = m._eps(B, eps)
    m.dev.forward(s, t, None, B, 1.0 / B, e, 1, 1)
    got = m.dev.losses.cpu().numpy()
    assert got[0] == pytest.approx(recon, rel=1e-4) and got[1] == pytest.approx(kl, rel=1e-4)
    m.dev.backward(s, None, e, 1.0 / B, 0)
    dg = m.dev.export_grads()
    worst = {k: rel_err(dg[k], grads[k]) for k in grads}
    assert max(worst.values()) < 2e-4, worst

    # three sess.run(train_step) of the reference graph vs three train_step() of the drop-in
    m2 = ConvVAE(np.array([80, 160, 3]), np.array([80, 160, target_c]), z_dim=64, model_dir=str(tmp_path / "b"), precision="fp32")
    m2.set_weights(params)
    m2.init_session(init_logging=False)
    for step in range(3):
        ee = np.random.RandomState(100 + step).standard_normal((B, 64)).astype(np.float32)
        r_ref, k_ref = g.run(["vae/Mean_1", "vae/Mean_2", "vae/Adam"], dict(feed, **{VAE_EPS: ee[None]}))[:2]
        r_got, k_got = m2.train_step(src, tgt, eps=ee)
        assert r_got == pytest.approx(r_ref, rel=1e-4) and k_got == pytest.approx(k_ref, rel=2e-4), step
    after = m2.dev.export_params()
    for k in params:
        upd_ref, upd_got = g.vars[k] - params[k], after[k] - params[k]
        assert np.mean(np.abs(upd_got - upd_ref) > 0.02 * 3e-4) < 1e-3, k
    assert m2.beta1_power == pytest.approx(float(g.vars["vae/beta1_power"]), rel=1e-6)


def test_hip_inference_vae_matches_the_agent_graph(tmp_path):
    """training=False, as run_eval / train.py build the VAE next to the agent: encode, mean-fed reconstruction, generate_from_latent."""
    g, _ = load_graph("ppo")
    params = vo.init_vae_params(7, 64, (80, 160, 3), (80, 160, 1))                          # the shipped agent was trained on the seg VAE
    init_variables(g, params)
    src = (np.random.RandomState(8).randint(0, 256, (2, 80, 160, 3)) / 255.0).astype(np.float32)
    m = ConvVAE(np.array([80, 160, 3]), np.array([80, 160, 1]), z_dim=64, model_dir=str(tmp_path), precision="fp32", training=False)
    m.set_weights(params)
    m.init_session(init_logging=False)
    feed = {"vae/source_state_placeholder": src}
    mean = g.run("vae/mean/BiasAdd", feed)
    assert rel_err(m.encode(src), mean) < 1e-4
    assert rel_err(m.generate_from_latent(mean), g.run("vae/reconstructed_states", feed)) < 1e-4          # the graph decodes the mean
    z = np.random.RandomState(9).standard_normal((2, 64)).astype(np.float32)
    assert rel_err(m.generate_from_latent(z), g.run("vae/reconstructed_states", {"vae/mean/BiasAdd": z})) < 1e-4


def test_hip_ppo_minibatch_matches_reference_graph(tmp_path):
    g, _ = load_graph("ppo")
    space = po.ActionSpace()
    params = po.init_ppo_params(seed=21, initial_std=1.0)
    rng = np.random.RandomState(22)
    old = {k.replace("policy/", "policy_old/", 1): (v + 0.02 * rng.standard_normal(v.shape)).astype(np.float32) for k, v in params.items()}
    init_variables(g, dict(params, **old))
    # hyper-parameters = the constants of the shipped agent's graph (checked in tests/test_ref_graph.py::test_ppo_graph_constants)
    m = PPO(np.array([67]), space, learning_rate=1e-4, lr_decay=1.0, epsilon=0.2, value_scale=1.0, entropy_scale=0.01, initial_std=1.0, model_dir=str(tmp_path))
    m.init_session(init_logging=False)
    m.dev.load_params(params, old)
    M = 32
    s = (0.5 * rng.standard_normal((M, 67))).astype(np.float32)
    a = rng.uniform(-1, 1, (M, 2)).astype(np.float32)
    R, A = rng.standard_normal(M).astype(np.float32), rng.standard_normal(M).astype(np.float32)
    feed = {"input_state_placeholder": s, "taken_action_placeholder": a, "returns_placeholder": R, "advantage_placeholder": A}
    nodes = adam_nodes(g)
    fetched = g.run(["Mean", "mul_2", "mul_3", "sub_1", "Exp"] + [grad for *_, grad in nodes], feed)
    pol, val, ent, loss, ratio = fetched[:5]
    grads = {var: x for (_, var, *_), x in zip(nodes, fetched[5:])}
    d = m.dev
    d.forward_backward(m._to_dev(s, (M, 67)), m._to_dev(a, (M, 2)), m._to_dev(R, (M,)), m._to_dev(A, (M,)), M, 1.0 / M, 1.0)
    L = d.losses.cpu().numpy()
    for got, want in zip(L, (pol, val, ent, loss, float(np.mean(ratio)))):
        assert got == pytest.approx(float(want), rel=1e-4, abs=1e-6)
    dg = d.export_grads()
    worst = {k: rel_err(dg[k], grads[k]) for k in grads}
    assert max(worst.values()) < 2e-4, worst

    noise = rng.standard_normal((M, 2)).astype(np.float32)
    act_ref, v_ref, mean_ref = g.run(["policy/clip_by_value", "policy/Squeeze", "policy/add_1"], dict(feed, **{PPO_EPS: noise[None]}))
    act, v = m.predict(s, noise=noise)
    act_g, _ = m.predict(s, greedy=True)
    assert np.abs(act - act_ref).max() < 1e-4 and np.abs(v - v_ref).max() < 1e-4 and np.abs(act_g - mean_ref).max() < 1e-4

    # update_old_policy (the graph's group_deps assigns) + three minibatch steps, on a fresh engine (the gradient buffer above was
    # filled without an Adam step to clear it)
    m2 = PPO(np.array([67]), space, learning_rate=1e-4, lr_decay=1.0, epsilon=0.2, value_scale=1.0, entropy_scale=0.01, initial_std=1.0, model_dir=str(tmp_path / "b"))
    m2.init_session(init_logging=False)
    m2.dev.load_params(params, old)
    g.run("group_deps")
    m2.update_old_policy()
    for step in range(3):
        g.run(["Adam", "Assign"], feed)
        m2.train(s, a, R, A)
    after = m2.dev.export_params()
    for k in params:
        upd_ref, upd_got = g.vars[k] - params[k], after[k] - params[k]
        assert np.mean(np.abs(upd_got - upd_ref) > 0.02 * 3e-4) < 1e-3, k
    old_after = m2.dev.export_old()
    for k in params:
        assert np.array_equal(old_after[k.replace("policy/", "policy_old/", 1)], params[k])
    assert m2.train_step_counter == 3 == int(g.vars["train_step_counter"])
