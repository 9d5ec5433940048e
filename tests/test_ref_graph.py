"""The CPU oracle against the reference's OWN serialized TensorFlow graphs (tests/golden/ref_graph_*.json.gz, decoded from the
MetaGraphDef `.meta` files the reference ships next to its checkpoints; generator: tests/golden/make_graph_fixture.py).

oracle/tf_graph.py executes those node lists -- forward pass, losses, the `gradients/` sub-graph tf.gradients built, ApplyAdam -- in
float64, and the oracle restatements run in float64 on the same inputs, so the tolerances below are rounding-only:
    forward / losses / every gradient tensor   1e-9 relative (measured ~1e-15; constants the graph holds as float32, e.g. the
                                               entropy's 0.5*log(2*pi*e) and entropy_scale, bound it at ~1e-9)
    parameters after 3 Adam steps              2e-5 of each tensor's max (the oracle's Adam is float32 like TensorFlow's kernel)
This is what pins the oracle to the reference rather than to a reading of its Python source; what stays unpinned is the float32
rounding inside TensorFlow's kernels and its random streams (noise is fed in)."""
import numpy as np
import pytest
import torch

from oracle import ppo_oracle as po
from oracle import vae_oracle as vo
from ref_graph_helpers import PPO_EPS, VAE_EPS, adam_nodes, init_variables, load_graph


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def vae_case(which, seed=3, B=3):
    tc = 3 if which == "vae_rgb" else 1
    g, doc = load_graph(which)
    params = vo.init_vae_params(seed, 64, (80, 160, 3), (80, 160, tc))
    rng = np.random.RandomState(seed + 1)
    for k in params:                                           # non-zero biases: every path of the graph carries signal
        if k.endswith("bias"):
            params[k] = (0.05 * rng.standard_normal(params[k].shape)).astype(np.float32)
    init_variables(g, params)
    src = (rng.randint(0, 256, (B, 80, 160, 3)) / 255.0).astype(np.float32)
    tgt = src if tc == 3 else (rng.randint(0, 13, (B, 80, 160, 1)) / 12.0).astype(np.float32)
    eps = rng.standard_normal((B, 64)).astype(np.float32)
    feed = {"vae/source_state_placeholder": src, "vae/target_state_placeholder": tgt, VAE_EPS: eps[None]}
    return g, params, src, tgt, eps, feed


# ---------------------------------------------------------------------------------------------------- ConvVAE
def test_fixture_provenance_and_variable_tables(golden_dir):
    import json
    import os
    ref_vars = json.load(open(os.path.join(golden_dir, "ref_variables.json")))
    for which, key in (("vae_rgb", "vae_rgb"), ("vae_seg", "vae_seg"), ("ppo", "ppo_agent")):
        g, doc = load_graph(which)
        assert doc["source"].endswith(".meta") and doc["generator"] == "tests/golden/make_graph_fixture.py"
        table = ref_vars[key]
        graph_vars = {n: list(g.variable_shape(n)) for n in g.variables()}
        saved = {k: v["shape"] for k, v in table.items()}
        # the checkpoint index (ref_variables.json) and the graph describe the same variables (the VAE's metric accumulators are local, unsaved)
        assert {k: v for k, v in graph_vars.items() if k in saved} == saved
        assert all("mean_" in k or "metrics" in k for k in graph_vars if k not in saved), [k for k in graph_vars if k not in saved]


def test_vae_graph_constants_match_the_product():
    from vae import models as vm
    g, _ = load_graph("vae_rgb", np.float32)
    assert g.const("vae/Adam/beta1") == np.float32(vm.ADAM_BETA1) and g.const("vae/Adam/beta2") == np.float32(vm.ADAM_BETA2)
    assert g.const("vae/Adam/epsilon") == np.float32(vm.ADAM_EPSILON) and g.const("vae/Adam/learning_rate") == np.float32(1e-4)
    assert g.const("vae/beta1_power/initial_value") == np.float32(0.9) and g.const("vae/beta2_power/initial_value") == np.float32(0.999)
    nodes = adam_nodes(g)
    assert len(nodes) == 22
    for n, *_ in nodes:                                        # constant learning rate: ExponentialDecay is built but NOT wired into Adam
        assert g.nodes[n]["input"][5] == "vae/Adam/learning_rate"
        assert g.attr(g.nodes[n], "use_nesterov", "b", False) is False
    for i, cout in enumerate((32, 64, 128, 256)):
        n = g.nodes["vae/encoder/conv%d/Conv2D" % (i + 1)]
        assert g.attr(n, "strides", "list")["i"] == [1, 2, 2, 1] and g.attr(n, "padding", "s") == "VALID" and g.attr(n, "data_format", "s") == "NHWC"
        assert g.variable_shape("vae/encoder/conv%d/kernel" % (i + 1))[::3] == (4, cout)
    for i, k in enumerate((4, 4, 5, 4)):
        n = g.nodes["vae/decoder/deconv%d/conv2d_transpose" % (i + 1)]
        assert n["op"] == "Conv2DBackpropInput" and g.attr(n, "strides", "list")["i"] == [1, 2, 2, 1] and g.attr(n, "padding", "s") == "VALID"
        assert g.variable_shape("vae/decoder/deconv%d/kernel" % (i + 1))[:2] == (k, k)
    assert g.const("vae/kl_divergence/mul/x") == np.float32(-0.5) and g.const("vae/mul_1/x") == np.float32(1.0)       # beta = 1
    assert vm.adam_alpha(1e-4, np.float32(0.9), np.float32(0.999)) == pytest.approx(1e-4 * np.sqrt(1 - 0.999) / (1 - 0.9), rel=2e-5)       # float32 (1 - beta2_power)


def test_initializers_match_reference_graph_constants():
    """tf.layers' default initializers leave their limits in the graph as constants: the product's and the oracle's samplers use the same."""
    from mi355 import init as mi
    for which in ("vae_rgb", "vae_seg", "ppo"):
        g, _ = load_graph(which, np.float32)
        seen = 0
        for name in g.variables():
            hi = name + "/Initializer/random_uniform/max"
            if hi in g.nodes:
                shape = g.variable_shape(name)
                assert mi.glorot_limit(shape) == pytest.approx(float(g.const(hi)), rel=1e-6), name
                assert float(g.const(name + "/Initializer/random_uniform/min")) == -float(g.const(hi))
                draw, odraw = mi.glorot_uniform(np.random.RandomState(0), shape), vo.glorot_uniform(np.random.RandomState(0), shape)
                assert np.abs(draw).max() <= float(g.const(hi)) * (1 + 1e-6) and np.abs(draw).max() > 0.9 * float(g.const(hi))
                assert np.array_equal(draw, odraw)
                seen += 1
            elif name.endswith("/bias") and "Adam" not in name:
                assert not np.any(g.run(name + "/Initializer/zeros"))
        # 4 conv + 2 heads + dense1 + 4 deconv; the agent's graph holds the (inference) VAE too, + 5 + 5 dense kernels of policy / policy_old
        assert seen == (11 if which != "ppo" else 21), (which, seen)
    g, _ = load_graph("ppo", np.float32)
    std = float(g.const("policy/action_mean/kernel/Initializer/truncated_normal/stddev"))
    assert mi.truncnormal_stddev((300, 2), 0.1) == pytest.approx(std, rel=1e-6)
    x = mi.variance_scaling_fan_in_truncnormal(np.random.RandomState(0), (300, 2), 0.1)
    assert np.abs(x).max() <= 2 * std * (1 + 1e-6) and np.array_equal(x, po.variance_scaling_truncnormal(np.random.RandomState(0), (300, 2), 0.1))
    assert g.const("policy/action_logstd/initial_value").tolist() == [0.0, 0.0]                   # log(initial_std = 1.0)
    assert mi.init_ppo(0, 67, 2, 1.0)["policy/action_logstd"].tolist() == [0.0, 0.0]


@pytest.mark.parametrize("which", ["vae_rgb", "vae_seg"])
def test_vae_oracle_forward_losses_gradients_match_reference_graph(which):
    g, params, src, tgt, eps, feed = vae_case(which)
    (recon, kl, loss), grads, fw = vo.vae_loss_and_grads(params, src, tgt, eps, beta=1.0, dtype=torch.float64)
    mean, logvar, z, logits, g_recon, g_kl, g_loss = g.run(["vae/mean/BiasAdd", "vae/logstd_sqare/BiasAdd", "vae/Squeeze", "vae/reconstructed_logits/Reshape",
                                                            "vae/Mean_1", "vae/Mean_2", "vae/add_1"], feed)
    assert rel_err(mean, fw["mean"].numpy()) < 1e-9 and rel_err(logvar, fw["logvar"].numpy()) < 1e-9
    assert rel_err(z, fw["z"].numpy()) < 1e-9 and rel_err(logits, fw["logits"].numpy()) < 1e-9
    assert g_recon == pytest.approx(recon, rel=1e-10) and g_kl == pytest.approx(kl, rel=1e-10) and g_loss == pytest.approx(loss, rel=1e-10)
    nodes = adam_nodes(g)
    assert [v for _, v, *_ in nodes] == list(params)                                   # TF creation order == the oracle's variable order
    got = g.run([grad for *_, grad in nodes], feed)
    worst = {var: rel_err(x, grads[var]) for (_, var, *_), x in zip(nodes, got)}
    assert max(worst.values()) < 1e-9, worst


def test_vae_inference_surface_matches_reference_graph():
    g, params, src, tgt, eps, feed = vae_case("vae_rgb")
    o = vo.OracleVAE(params=params, dtype=torch.float64)
    assert rel_err(g.run("vae/mean/BiasAdd", feed), o.encode(src)) < 1e-6                       # encode() returns the mean
    rec = g.run("vae/reconstructed_states", feed)                                             # sigmoid(logits), [B, 38400]
    assert rel_err(rec, np.stack([r.reshape(-1) for r in o.reconstruct(src, eps=eps)])) < 1e-6
    bad = dict(feed)
    bad["vae/source_state_placeholder"] = src + 1.0                                           # verify_range's tf.Assert is in the graph
    with pytest.raises(AssertionError):
        g.run("vae/mean/BiasAdd", bad)
    with pytest.raises(Exception):
        vo.verify_range(src + 1.0)


def test_inference_mode_vae_inside_the_agent_graph():
    """The shipped agent's graph holds the VAE as run_eval / train.py build it (training=False): the decoder is fed the MEAN, no sampling."""
    g, _ = load_graph("ppo")
    assert g.nodes["vae/decoder/dense1/MatMul"]["input"][0] == "vae/mean/BiasAdd"
    assert not [n for n in g.order if n.startswith("vae/") and g.nodes[n]["op"] in ("RandomStandardNormal", "ApplyAdam")]
    params = vo.init_vae_params(7, 64, (80, 160, 3), (80, 160, 1))                          # the shipped agent was trained on the seg VAE
    init_variables(g, params)
    src = (np.random.RandomState(8).randint(0, 256, (2, 80, 160, 3)) / 255.0).astype(np.float32)
    feed = {"vae/source_state_placeholder": src}
    o = vo.OracleVAE(target_shape=(80, 160, 1), params=params, training=False, dtype=torch.float64)
    mean = g.run("vae/mean/BiasAdd", feed)
    assert rel_err(mean, o.encode(src)) < 1e-6                                               # what train.py / run_eval.py feed the agent
    rec = g.run("vae/reconstructed_states", feed)                                            # [2, 12800]: decoder of the mean
    assert rel_err(rec, np.stack([np.asarray(r).reshape(-1) for r in o.generate_from_latent(mean)])) < 1e-6
    z = np.random.RandomState(9).standard_normal((2, 64)).astype(np.float32)               # generate_from_latent: feed the sample tensor
    got = g.run("vae/reconstructed_states", {"vae/mean/BiasAdd": z})
    assert rel_err(got, np.stack([np.asarray(r).reshape(-1) for r in o.generate_from_latent(z)])) < 1e-6


def test_vae_adam_trajectory_matches_reference_graph():
    g, params, src, tgt, eps, feed = vae_case("vae_rgb", seed=5, B=2)
    o = vo.OracleVAE(params=params, dtype=torch.float64)
    for step in range(3):
        g.run("vae/Adam", feed)                                  # sess.run(train_step)
        o.train_step(src, tgt, eps)
        for k in params:
            assert rel_err(g.vars[k], o.params[k]) < 2e-5, (step, k)
            assert rel_err(g.vars["vae/" + k + "/Adam"], o.adam.m[k]) < 2e-5 and rel_err(g.vars["vae/" + k + "/Adam_1"], o.adam.v[k]) < 2e-5, (step, k)
        assert float(g.vars["vae/beta1_power"]) == pytest.approx(float(o.adam.beta1_power), rel=1e-6)
        assert float(g.vars["vae/beta2_power"]) == pytest.approx(float(o.adam.beta2_power), rel=1e-6)
    moved = max(rel_err(o.params[k], params[k]) for k in params if np.abs(params[k]).max() > 0)
    assert moved > 1e-3                                          # the 2e-5 above is small against what three steps change


def test_vae_epoch_loops_and_metrics_replayed_on_the_reference_graph():
    """train_one_epoch / evaluate (vae/models.py:207-231) replayed sess.run by sess.run on the reference graph -- legacy-numpy shuffle,
    N // batch_size minibatches, tf.metrics.mean accumulators (local variables reset per epoch), inc_step_idx once per epoch."""
    g, params, src, tgt, eps, feed = vae_case("vae_rgb", seed=13, B=5)
    o = vo.OracleVAE(params=params, dtype=torch.float64)
    noise = np.random.RandomState(14).standard_normal((8, 2, 64)).astype(np.float32)
    local = [n for n in g.variables() if n.startswith(("vae/mean_3/", "vae/mean_4/"))]     # tf.local_variables_initializer()

    def graph_epoch(train, bs, draws):
        indices = np.arange(len(src))
        np.random.shuffle(indices)
        for n in local:
            g.set_variable(n, 0.0)
        ops = (["vae/Adam"] if train else []) + ["vae/mean_3/update_op", "vae/mean_4/update_op"]
        for i in range(len(src) // bs):
            mb = indices[i * bs:(i + 1) * bs]
            g.run(ops, {"vae/source_state_placeholder": src[mb], "vae/target_state_placeholder": tgt[mb], VAE_EPS: draws[i][None]})
        if train:
            g.run("vae/Assign")                                  # inc_step_idx
        return g.run(["vae/mean_4/value", "vae/mean_3/value"])   # [mean_reconstruction_loss, mean_kl_loss]

    for epoch in range(2):
        it = iter(noise[4 * epoch:])
        np.random.seed(100 + epoch)
        want_val = o.evaluate(src, tgt, 2, lambda n: next(it))
        want_train = o.train_one_epoch(src, tgt, 2, lambda n: next(it))
        np.random.seed(100 + epoch)
        got_val = graph_epoch(False, 2, noise[4 * epoch:])
        got_train = graph_epoch(True, 2, noise[4 * epoch + 2:])
        assert got_val == pytest.approx(want_val, rel=1e-5) and got_train == pytest.approx(want_train, rel=1e-5), epoch
    assert int(g.vars["vae/step_idx"]) == 2 == o.step_idx
    assert int(g.vars["vae/mean_3/count"]) == 2                    # 5 // 2 minibatches, the remainder is dropped


# ---------------------------------------------------------------------------------------------------- PPO
def ppo_case(seed=5, M=32):
    g, _ = load_graph("ppo")
    space = po.ActionSpace()
    params = po.init_ppo_params(seed=seed, initial_std=1.0)
    rng = np.random.RandomState(seed + 1)
    old = {k.replace("policy/", "policy_old/", 1): (v + 0.01 * rng.standard_normal(v.shape)).astype(np.float32) for k, v in params.items()}
    # the shipped agent's hyper-parameters (train.py defaults it was trained with): they are constants of its graph, checked below
    o = po.OraclePPO(np.array([67]), space, learning_rate=1e-4, lr_decay=1.0, epsilon=0.2, value_scale=1.0, entropy_scale=0.01, initial_std=1.0,
                     params=params, dtype=torch.float64)
    o.params_old = {k: v.copy() for k, v in old.items()}
    init_variables(g, dict(params, **old))
    s = (0.5 * rng.standard_normal((M, 67))).astype(np.float32)
    a = rng.uniform(-1, 1, (M, 2)).astype(np.float32)
    R, A = rng.standard_normal(M).astype(np.float32), rng.standard_normal(M).astype(np.float32)
    feed = {"input_state_placeholder": s, "taken_action_placeholder": a, "returns_placeholder": R, "advantage_placeholder": A}
    return g, o, params, old, (s, a, R, A), feed, rng


def test_ppo_graph_constants():
    g, _ = load_graph("ppo", np.float32)
    assert g.const("clip_by_value/Minimum/y") == np.float32(1.2) and g.const("clip_by_value/y") == np.float32(0.8)         # 1 +- epsilon
    assert g.const("mul_2/y") == np.float32(1.0) and g.const("mul_3/y") == np.float32(0.01)                                # value_scale, entropy_scale
    assert g.const("policy/mul/y").tolist() == [2.0, 1.0] and g.const("policy/add_1/x").tolist() == [-1.0, 0.0]           # action range of CarlaEnv
    assert g.const("policy/clip_by_value/Minimum/y").tolist() == [1.0, 1.0] and g.const("policy/clip_by_value/y").tolist() == [-1.0, 0.0]
    assert g.const("Adam/beta1") == np.float32(0.9) and g.const("Adam/beta2") == np.float32(0.999) and g.const("Adam/epsilon") == np.float32(1e-8)
    nodes = adam_nodes(g)
    assert [v for _, v, *_ in nodes] == list(po.ppo_variable_specs())               # 13 trainable tensors, policy scope only
    assert all(g.nodes[n]["input"][5] == "ExponentialDecay" for n, *_ in nodes)     # decayed learning rate IS wired in here (unlike the VAE)
    assert g.nodes["ExponentialDecay/Cast_2"]["input"] == ["episode_counter/read"]  # ... and decays per EPISODE
    assert g.attr(g.nodes["policy/Sum"], "keep_dims", "b") is True                  # log-prob sum keeps [M, 1]: ratio*advantage stays [M, 1]
    upd = g.nodes["group_deps"]["input"]                                            # update_old_policy: 13 assigns policy -> policy_old
    assert len(upd) == 13
    for ref in upd:
        tgt, src = g.nodes[ref.lstrip("^")]["input"]
        assert tgt.startswith("policy_old/") and src == tgt.replace("policy_old/", "policy/", 1) + "/read"


def test_ppo_oracle_losses_gradients_predict_match_reference_graph():
    g, o, params, old, (s, a, R, A), feed, rng = ppo_case()
    scal, grads = o.loss_and_grads(s, a, R, A)
    pol, val, ent, loss, ratio = g.run(["Mean", "mul_2", "mul_3", "sub_1", "Exp"], feed)
    assert ratio.shape == (32, 1) and float(np.mean(ratio)) == pytest.approx(scal["ratio_mean"], rel=1e-9)
    assert pol == pytest.approx(scal["policy_loss"], rel=1e-9) and val == pytest.approx(scal["value_loss"], rel=1e-9)
    assert ent == pytest.approx(scal["entropy_loss"], rel=1e-7) and loss == pytest.approx(scal["loss"], rel=1e-8)
    assert np.any(ratio > 1.2) or np.any(ratio < 0.8) or np.abs(ratio - 1).max() > 1e-3         # theta != theta_old: the clip can act
    nodes = adam_nodes(g)
    got = g.run([grad for *_, grad in nodes], feed)
    worst = {var: rel_err(x, grads[var]) for (_, var, *_), x in zip(nodes, got)}
    assert max(worst.values()) < 1e-8, worst
    noise = rng.standard_normal((32, 2)).astype(np.float32)
    act, value, mean = g.run(["policy/clip_by_value", "policy/Squeeze", "policy/add_1"], dict(feed, **{PPO_EPS: noise[None]}))
    a_s, v_s = o.predict(s, noise=noise)
    a_g, _ = o.predict(s, greedy=True)
    assert np.abs(act - a_s).max() < 1e-6 and np.abs(value - v_s).max() < 1e-6 and np.abs(mean - a_g).max() < 1e-6
    assert (act[:, 0] >= -1).all() and (act[:, 0] <= 1).all() and (act[:, 1] >= 0).all() and (act[:, 1] <= 1).all()


def test_ppo_update_old_policy_and_adam_trajectory_match_reference_graph():
    g, o, params, old, (s, a, R, A), feed, rng = ppo_case(seed=9)
    for step in range(3):
        if step == 1:                                            # train.py:192: theta_old <- theta before each update
            g.run("group_deps")
            o.update_old_policy()
            for k in params:
                assert np.array_equal(g.vars[k.replace("policy/", "policy_old/", 1)], g.vars[k])
        g.run(["Adam", "Assign"], feed)
        o.train(s, a, R, A)
        for k in params:
            assert rel_err(g.vars[k], o.params[k]) < 2e-5, (step, k)
            assert rel_err(g.vars[k + "/Adam"], o.adam.m[k]) < 2e-5 and rel_err(g.vars[k + "/Adam_1"], o.adam.v[k]) < 2e-5, (step, k)
    assert int(g.vars["train_step_counter"]) == 3 == o.train_step_counter
    assert float(g.vars["beta1_power"]) == pytest.approx(0.9 ** 4, rel=1e-6)
