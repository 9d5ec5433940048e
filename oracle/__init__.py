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

PARITY UNPINNED.  The reference is TensorFlow 1.13 graph code; tensorflow / tensorflow_probability
are not installable here, so the reference itself cannot be run, and it ships no tests or golden
vectors for this path.  The oracle is therefore pinned only by
  * the variable names / shapes / parameter counts of the reference's shipped checkpoints
    (tests/golden/ref_variables.json, parsed from the `.index` files),
  * the untrained-model validation losses logged by the reference's own runs
    (tests/golden/ref_event_scalars.json),
  * analytic known-answers and fp64 finite-difference gradient checks (tests/test_oracle_*.py),
  * oracle/gae_ref.c — an independent plain-C restatement of the GAE / advantage-normalise / TF-Adam
    recurrences, compared bit-for-bit (GAE) with the scipy form used by the reference.
"""
