"""CPU oracle for the ConvVAE + PPO hot path of bitsauce/Carla-ppo.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it — as the checker / the timed CPU baseline, never as the
thing shipped.  The product (`carla-ppo_amd/`) never imports `oracle` and has no CPU fallback.

It is a line-by-line restatement, in torch-CPU fp32 (fp64 on request) and numpy/scipy f64, of

    vae/models.py:7-30,85-142,188-231,249-266   ppo.py:38-66,119-147,218-251,275-276
    utils.py:25-28,45-50                          train.py:171-207
    vae/train_vae.py:15-18,41-45,70-75,138-161    vae_common.py:18-23,47-59

with TensorFlow's variable names and layouts (HWIO conv kernels, [kh,kw,out,in] transposed-conv
kernels, [in,out] dense kernels), TF-form Adam, tfp's Normal log_prob/entropy formulas, injected
reparameterisation noise and the legacy numpy RNG for minibatch permutations.

PARITY PINNING.  The reference is TensorFlow 1.13 graph code; tensorflow / tensorflow_probability are not
installable here, so the reference's Python cannot be run, and it ships no tests or golden vectors for this
path.  It DOES ship its serialized graphs: the MetaGraphDef `.meta` file next to every checkpoint holds the
node list of the graph it trained with (forward, losses, the gradients/ sub-graph, ApplyAdam + constants).
  * PINNED to those graphs: tests/golden/make_graph_fixture.py decodes them (rgb VAE, seg VAE, PPO agent)
    into tests/golden/ref_graph_*.json.gz and oracle/tf_graph.py executes them in float64 from the ops'
    definitions; tests/test_ref_graph.py holds this oracle's forward pass, losses, all 22 + 13 gradient
    tensors (1e-9), inference outputs and 3-step Adam trajectories (2e-5, the oracle's Adam is float32) to
    the graphs, and tests/test_ref_graph_gpu.py compares the HIP path with the graphs directly.
  * NOT pinned (nothing here can): the float32 rounding inside TensorFlow's kernels and its random streams
    (noise is injected), and the Python-side loops around sess.run (minibatch schedules, GAE, epoch metrics),
    which are pinned as before by
      - the variable names / shapes / parameter counts of the shipped checkpoints (tests/golden/ref_variables.json),
      - the untrained-model validation losses logged by the reference's own runs (tests/golden/ref_event_scalars.json),
      - analytic known-answers and fp64 finite-difference gradient checks (tests/test_oracle_*.py),
      - oracle/gae_ref.c, an independent plain-C restatement of the GAE / advantage-normalise / TF-Adam
        recurrences, compared bit-for-bit (GAE) with the scipy form used by the reference.
"""
