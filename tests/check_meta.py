#!/usr/bin/env python3
"""Check the oracle -- and, with --gpu, the HIP path -- against ANY graph the reference serialized.

    python tests/check_meta.py <model_dir>/checkpoints/model.ckpt-N.meta [--gpu] [--batch 4] [--seed 0]

Every tf.train.Saver checkpoint of the reference comes with the MetaGraphDef of the graph that wrote it.  This decodes it (no TensorFlow
needed: oracle/tf_meta.py), executes it in float64 (oracle/tf_graph.py) on random weights and inputs, and reports how far the oracle's
restatement -- and the drop-in classes on the GPU -- are from it: forward outputs, loss terms, every gradient tensor.  The committed
tests (tests/test_ref_graph*.py) do this for the three graphs the reference ships; this is the same check for a user's own model
directory (other z_dim / beta / target depth / PPO hyper-parameters: they are read from the graph's shapes and constants).
Lives under tests/ because it runs the oracle (test infrastructure, never the product)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import ppo_oracle as po  # noqa: E402
from oracle import tf_graph as tg  # noqa: E402
from oracle import tf_meta  # noqa: E402
from oracle import vae_oracle as vo  # noqa: E402
from ref_graph_helpers import PPO_EPS, VAE_EPS, adam_nodes, init_variables  # noqa: E402


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def report(title, rows):
    print(title)
    for k, v in rows:
        print("    %-44s %.3e" % (k, v))
    return max(v for _, v in rows)


def check_vae(g, args):
    src_shape = tuple(g.attr(g.nodes["vae/source_state_placeholder"], "shape", "shape")[1:])
    tgt_shape = tuple(g.attr(g.nodes["vae/target_state_placeholder"], "shape", "shape")[1:])
    z_dim = g.variable_shape("vae/mean/kernel")[1]
    if "vae/encoder/conv1/kernel" not in g.nodes:
        raise SystemExit("this VAE graph is not the ConvVAE (MlpVAE graphs: compare through tests/test_f_mlp_vae_gpu.py's oracle instead)")
    training = any(n["op"] == "ApplyAdam" for n in g.nodes.values())
    beta = float(g.const("vae/mul_1/x")) if "vae/mul_1/x" in g.nodes else 1.0
    print("ConvVAE graph: source %s target %s z_dim %d beta %g  (%s)" % (src_shape, tgt_shape, z_dim, beta, "training" if training else "inference"))
    params = vo.init_vae_params(args.seed, z_dim, src_shape, tgt_shape)
    rng = np.random.RandomState(args.seed + 1)
    for k in params:
        if k.endswith("bias"):
            params[k] = (0.05 * rng.standard_normal(params[k].shape)).astype(np.float32)
    init_variables(g, params)
    B = args.batch
    src = (rng.randint(0, 256, (B,) + src_shape) / 255.0).astype(np.float32)
    tgt = src if tgt_shape == src_shape else rng.uniform(0, 1, (B,) + tgt_shape).astype(np.float32)
    eps = rng.standard_normal((B, z_dim)).astype(np.float32)
    feed = {"vae/source_state_placeholder": src, "vae/target_state_placeholder": tgt}
    worst = 0.0
    if not training:
        o = vo.OracleVAE(src_shape, tgt_shape, z_dim, params=params, training=False, dtype=torch.float64)
        mean = g.run("vae/mean/BiasAdd", feed)
        rec = g.run("vae/reconstructed_states", feed)
        worst = report("oracle vs graph (inference)", [("encode (mean)", rel_err(o.encode(src), mean)),
                                                        ("decode of the mean", rel_err(np.stack([np.asarray(r).reshape(-1) for r in o.generate_from_latent(mean)]), rec))])
        return worst, (params, src, tgt, eps, feed, None, None)
    feed[VAE_EPS] = eps[None]
    nodes = adam_nodes(g)
    fetched = g.run(["vae/mean/BiasAdd", "vae/reconstructed_logits/Reshape", "vae/Mean_1", "vae/Mean_2"] + [grad for *_, grad in nodes], feed)
    mean, logits, recon, kl = fetched[:4]
    grads = {var: x for (_, var, *_), x in zip(nodes, fetched[4:])}
    (o_recon, o_kl, _), o_grads, fw = vo.vae_loss_and_grads(params, src, tgt, eps, beta=beta, dtype=torch.float64)
    rows = [("mean", rel_err(fw["mean"].numpy(), mean)), ("logits", rel_err(fw["logits"].numpy(), logits)),
            ("reconstruction loss", abs(o_recon / recon - 1)), ("kl loss", abs(o_kl / kl - 1))]
    rows += [("d " + k, rel_err(o_grads[k], grads[k])) for k in grads]
    worst = report("oracle (float64) vs graph (float64): relative to each tensor's max", rows)
    return worst, (params, src, tgt, eps, feed, (mean, recon, kl), grads)


def check_vae_gpu(g, args, case):
    from vae.models import ConvVAE
    import tempfile
    params, src, tgt, eps, feed, ref, grads = case
    if ref is None:
        raise SystemExit("--gpu on an inference graph: use tests/test_ref_graph_gpu.py::test_hip_inference_vae_matches_the_agent_graph as the template")
    mean, recon, kl = ref
    m = ConvVAE(np.array(src.shape[1:]), np.array(tgt.shape[1:]), z_dim=eps.shape[1], model_dir=tempfile.mkdtemp(), precision="fp32")
    m.set_weights(params)
    m.init_session(init_logging=False)
    B = len(src)
    s = m._frames(src, int(np.prod(src.shape[1:])), "src")
    t = s if tgt is src else m._frames(tgt, m.dev.P, "tgt")
    e = m._eps(B, eps)
    m.dev.forward(s, t, None, B, 1.0 / B, e, 1, 1)
    got = m.dev.losses.cpu().numpy()
    m.dev.backward(s, None, e, 1.0 / B, 0)
    dg = m.dev.export_grads()
    rows = [("encode (mean)", rel_err(m.encode(src), mean)), ("reconstruction loss", abs(got[0] / recon - 1)), ("kl loss", abs(got[1] / kl - 1))]
    rows += [("d " + k, rel_err(dg[k], grads[k])) for k in grads]
    return report("HIP path (fp32 mode, through the C ABI) vs graph (float64)", rows)


def check_ppo(g, args):
    input_dim, hidden1 = g.variable_shape("policy/dense/kernel")
    num_actions = g.variable_shape("policy/action_logstd")[0]
    hi_clip, lo_clip = float(g.const("clip_by_value/Minimum/y")), float(g.const("clip_by_value/y"))
    epsilon = (hi_clip - lo_clip) / 2
    value_scale, entropy_scale = float(g.const("mul_2/y")), float(g.const("mul_3/y"))
    scale, low = g.const("policy/mul/y"), g.const("policy/add_1/x")
    space = po.ActionSpace(low=tuple(low.tolist()), high=tuple((low + scale).tolist()))
    print("PPO graph: input %d, %d actions in [%s, %s], epsilon %.3g value_scale %g entropy_scale %g" %
          (input_dim, num_actions, space.low.tolist(), space.high.tolist(), epsilon, value_scale, entropy_scale))
    params = po.init_ppo_params(seed=args.seed, input_dim=input_dim, num_actions=num_actions, initial_std=1.0)
    rng = np.random.RandomState(args.seed + 1)
    old = {k.replace("policy/", "policy_old/", 1): (v + 0.01 * rng.standard_normal(v.shape)).astype(np.float32) for k, v in params.items()}
    init_variables(g, dict(params, **old))                        # (the agent's graph also holds the inference VAE: not evaluated here)
    o = po.OraclePPO(np.array([input_dim]), space, learning_rate=1e-4, lr_decay=1.0, epsilon=epsilon, value_scale=value_scale, entropy_scale=entropy_scale,
                     initial_std=1.0, params=params, dtype=torch.float64)
    o.params_old = {k: v.copy() for k, v in old.items()}
    M = max(args.batch, 8)
    s = (0.5 * rng.standard_normal((M, input_dim))).astype(np.float32)
    a = rng.uniform(space.low, space.high, (M, num_actions)).astype(np.float32)
    R, A = rng.standard_normal(M).astype(np.float32), rng.standard_normal(M).astype(np.float32)
    feed = {"input_state_placeholder": s, "taken_action_placeholder": a, "returns_placeholder": R, "advantage_placeholder": A}
    nodes = adam_nodes(g)
    fetched = g.run(["Mean", "mul_2", "mul_3", "sub_1"] + [grad for *_, grad in nodes], feed)
    pol, val, ent, loss = fetched[:4]
    grads = {var: x for (_, var, *_), x in zip(nodes, fetched[4:])}
    scal, o_grads = o.loss_and_grads(s, a, R, A)
    noise = rng.standard_normal((M, num_actions)).astype(np.float32)
    act, v = g.run(["policy/clip_by_value", "policy/Squeeze"], dict(feed, **{PPO_EPS: noise[None]}))
    a_o, v_o = o.predict(s, noise=noise)
    rows = [("policy objective", abs(scal["policy_loss"] / pol - 1)), ("value loss", abs(scal["value_loss"] / val - 1)), ("entropy term", abs(scal["entropy_loss"] / ent - 1)),
            ("loss", abs(scal["loss"] / loss - 1)), ("sampled + clipped action", rel_err(a_o, act)), ("value", rel_err(v_o, v))]
    rows += [("d " + k, rel_err(o_grads[k], grads[k])) for k in grads]
    worst = report("oracle (float64) vs graph (float64): relative to each tensor's max (constants held as float32 bound this at ~1e-8)", rows)
    return worst, (params, old, space, (epsilon, value_scale, entropy_scale), (s, a, R, A), (pol, val, ent, loss), grads)


def check_ppo_gpu(g, args, case):
    from ppo import PPO
    import tempfile
    params, old, space, (epsilon, value_scale, entropy_scale), (s, a, R, A), (pol, val, ent, loss), grads = case
    m = PPO(np.array([s.shape[1]]), space, learning_rate=1e-4, lr_decay=1.0, epsilon=epsilon, value_scale=value_scale, entropy_scale=entropy_scale,
            initial_std=1.0, model_dir=tempfile.mkdtemp())
    m.init_session(init_logging=False)
    m.dev.load_params(params, old)
    M = len(s)
    d = m.dev
    d.forward_backward(m._to_dev(s, s.shape), m._to_dev(a, a.shape), m._to_dev(R, (M,)), m._to_dev(A, (M,)), M, 1.0 / M, 1.0)
    L = d.losses.cpu().numpy()
    dg = d.export_grads()
    rows = [("policy objective", abs(L[0] / pol - 1)), ("value loss", abs(L[1] / val - 1)), ("entropy term", abs(L[2] / ent - 1)), ("loss", abs(L[3] / loss - 1))]
    rows += [("d " + k, rel_err(dg[k], grads[k])) for k in grads]
    return report("HIP path (fp32, through the C ABI) vs graph (float64)", rows)


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("meta")
    ap.add_argument("--gpu", action="store_true", help="also run the drop-in classes on cuda:0 against the graph")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    nodes = tf_meta.prune(tf_meta.graph_nodes(args.meta))
    g = tg.Graph(nodes)
    print("%s: %d nodes after dropping Saver / summary plumbing" % (args.meta, len(nodes)))
    if "policy/dense/kernel" in g.nodes:
        worst, case = check_ppo(g, args)
        tol, tol_gpu = 1e-7, 2e-4
        worst_gpu = check_ppo_gpu(g, args, case) if args.gpu else None
    elif "vae/mean/kernel" in g.nodes:
        worst, case = check_vae(g, args)
        tol, tol_gpu = 1e-7, 2e-4
        worst_gpu = check_vae_gpu(g, args, case) if args.gpu else None
    else:
        raise SystemExit("neither a VAE nor a PPO graph of the reference (no vae/mean/kernel, no policy/dense/kernel)")
    ok = worst < tol and (worst_gpu is None or worst_gpu < tol_gpu)
    print("worst oracle deviation %.3e (tolerance %.0e)%s -> %s" % (worst, tol, "" if worst_gpu is None else "; worst HIP deviation %.3e (tolerance %.0e)" % (worst_gpu, tol_gpu),
                                                                    "OK" if ok else "MISMATCH"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
