"""BASELINE configs[4] / SURVEY 8c "C5" at parity-test size: the synthetic replay (replay.replay_update: VAE encode -> values -> GAE +
per-trajectory normalisation -> PPO minibatch SGD, all device resident) against the same pipeline built from the oracle.

Tolerances (fp32 engines vs fp32 oracle, north_star's 1e-4): encode / value outputs 1e-4 of the tensor's max; GAE, returns and
normalised advantages bit-exact / 1e-12 against the oracle's fp64 statements on the SAME (device-produced) values; per-minibatch total loss,
value loss and mean probability ratio 1e-4 relative; the policy term -mean(min(r A, clip(r) A)) is a cancelling sum over NORMALISED
advantages (mean 0, std 1 per row), so its error is bounded relative to mean|A| = O(1): 1e-4 absolute.  Every stage is compared on the
stage's own inputs as the device produced them (the oracle's SGD consumes the device's states / returns / advantages), so the stage
tolerances do not compound."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ppo_oracle as po  # noqa: E402
from oracle import vae_oracle as vo  # noqa: E402
from ppo import PPO  # noqa: E402
from vae.models import ConvVAE  # noqa: E402
import replay  # noqa: E402


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


# (4, 16, 24, 2): small, partial last minibatch.  (64, 128, 2048, 1): BASELINE configs[4]'s own shape -- horizon 128, global minibatch 2048 (the
# large-minibatch form of the fused step: row chunks of 256, each chunk's gradients in its own slab, one ordered slab sum, flat Adam -- no atomics) -- at 1/16 of its 1024 trajectories, which is what the CPU
# oracle encodes in seconds (8,256 frames); train.py:171-207 end to end.
@pytest.mark.parametrize("R,T,batch,epochs", [(4, 16, 24, 2), (64, 128, 2048, 1)])
def test_synthetic_replay_matches_oracle_pipeline(tmp_path, R, T, batch, epochs):
    gamma, lam = 0.99, 0.95
    rng = np.random.RandomState(11)
    frames = rng.randint(0, 256, (R, T + 1, 80, 160, 3), dtype=np.uint8)
    meas = np.stack([rng.uniform(-1, 1, (R, T + 1)), rng.uniform(0, 1, (R, T + 1)), rng.uniform(0, 30, (R, T + 1))], axis=-1).astype(np.float32)
    actions = np.stack([rng.uniform(-1, 1, (R, T)), rng.uniform(0, 1, (R, T))], axis=-1).astype(np.float32)
    rewards = rng.uniform(0, 1, (R, T))
    dones = np.zeros((R, T)); dones[1, -1] = 1

    vparams = vo.init_vae_params(3)
    vae = ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=str(tmp_path / "vae"), precision="fp32", training=False)
    vae.set_weights(vparams)
    vae.init_session(init_logging=False)
    space = po.ActionSpace()
    hp = dict(learning_rate=1e-4, lr_decay=1.0, epsilon=0.2, value_scale=1.0, entropy_scale=0.01, initial_std=1.0)
    o = po.OraclePPO([67], space, seed=2, **hp)
    m = PPO(np.array([67]), space, model_dir=str(tmp_path / "ppo"), seed=2, **hp)
    m.set_weights(o.params)
    m.init_session(init_logging=False)

    np.random.seed(5)
    out = replay.replay_update(vae, m, frames, meas, actions, rewards, dones, gamma, lam, num_epochs=epochs, batch_size=batch, return_z=True)
    n_steps = epochs * -(-R * T // batch)                                           # (4, 16, 24, 2): 64 samples = 24 + 24 + 16 per epoch
    assert out["rows"] == (0, R) and out["samples_per_rank"] == R * T and len(out["losses"]) == n_steps

    # the same pipeline from the oracle's pieces
    f32 = frames.reshape(-1, 80, 160, 3).astype(np.float32) / 255.0
    ovae = vo.OracleVAE(params=vparams, training=False)
    z = np.concatenate([ovae.encode(f32[i:i + 512]) for i in range(0, len(f32), 512)])
    assert rel_err(out["z"], z) < 1e-4                                              # stage 1: encode (fp32 engine vs fp32 oracle)
    states = np.concatenate([out["z"], meas.reshape(-1, 3)], axis=1).astype(np.float32).reshape(R, T + 1, 67)   # the device's own states
    _, v_o = o.predict(states.reshape(-1, 67), greedy=True)
    assert rel_err(out["values"], v_o.reshape(R, T + 1)) < 1e-4                     # stage 2: value estimates

    # GAE / returns / normalisation: fp64, on the device-produced values -> bit-exact with the reference's numpy / scipy statements
    vals = out["values"]
    for r in range(R):
        ref = po.compute_gae(list(rewards[r]), list(vals[r, :T]), vals[r, T], list(dones[r].astype(bool)), gamma, lam)
        rr, aa = po.returns_and_normalized_advantages(ref.copy(), vals[r, :T].astype(np.float64))
        assert np.array_equal(out["returns"][r], rr)
        assert np.allclose(out["advantages"][r], aa, rtol=1e-12, atol=1e-12)

    # minibatch SGD: same legacy-numpy permutations, same minibatch boundaries (last one partial)
    s_flat, a_flat = states[:, :T].reshape(R * T, 67), actions.reshape(R * T, 2)
    ret_flat, adv_flat = out["returns"].reshape(-1), out["advantages"].reshape(-1)
    np.random.seed(5)
    o.update_old_policy()
    logs = [o.train(s_flat[mb], a_flat[mb], ret_flat[mb], adv_flat[mb]) for mb in po.minibatch_schedule(R * T, batch, epochs)]
    assert len(logs) == len(out["losses"])
    for i, (want, got) in enumerate(zip(logs, out["losses"])):
        assert got["loss"] == pytest.approx(want["loss"], rel=1e-4, abs=1e-4), (i, got, want)      # contains the policy term (abs 1e-4)
        assert got["value_loss"] == pytest.approx(want["value_loss"], rel=1e-4), (i, got, want)
        assert got["policy_loss"] == pytest.approx(want["policy_loss"], abs=1e-4), (i, got, want)
        assert got["prob_ratio"] == pytest.approx(want["ratio_mean"], rel=1e-4), (i, got, want)
    assert out["losses"][0]["prob_ratio"] == pytest.approx(1.0, abs=1e-5)              # theta_old == theta at the first step
    assert m.get_train_step_idx() == n_steps


def test_replay_reads_a_device_resident_frame_table_where_it_lies(tmp_path):
    """Round 4: `replay_update(frames=<cuda uint8 tensor>)` -- the recording already in HBM, no PCIe in the call -- gives exactly the numbers of the host-array
    form (same kernels on the same bytes); the fp32 engine refuses a uint8 device table (it reads float frames) instead of reinterpreting it."""
    import torch
    R, T = 3, 8
    rng = np.random.RandomState(4)
    frames = rng.randint(0, 256, (R, T + 1, 80, 160, 3), dtype=np.uint8)
    meas = rng.uniform(0, 1, (R, T + 1, 3)).astype(np.float32)
    actions = rng.uniform(0, 1, (R, T, 2)).astype(np.float32)
    rewards, dones = rng.uniform(0, 1, (R, T)), np.zeros((R, T))
    outs = []
    for resident in (False, True):
        vae = ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=str(tmp_path / ("v%d" % resident)), precision="bf16", training=False, seed=0)
        vae.set_weights(vo.init_vae_params(3))
        vae.init_session(init_logging=False)
        hp = dict(learning_rate=1e-4, lr_decay=1.0, epsilon=0.2, value_scale=1.0, entropy_scale=0.01, initial_std=1.0)
        m = PPO(np.array([67]), po.ActionSpace(), model_dir=str(tmp_path / ("p%d" % resident)), seed=2, **hp)
        m.init_session(init_logging=False)
        np.random.seed(5)
        f = torch.from_numpy(frames).to("cuda") if resident else frames
        outs.append(replay.replay_update(vae, m, f, meas, actions, rewards, dones, 0.99, 0.95, num_epochs=1, batch_size=8, return_z=True))
    assert np.array_equal(outs[0]["z"], outs[1]["z"]) and np.array_equal(outs[0]["returns"], outs[1]["returns"])
    assert [l["loss"] for l in outs[0]["losses"]] == [l["loss"] for l in outs[1]["losses"]]
    v32 = ConvVAE(np.array([80, 160, 3]), z_dim=64, model_dir=str(tmp_path / "v32"), precision="fp32", training=False, seed=0)
    v32.init_session(init_logging=False)
    with pytest.raises(ValueError, match="bf16 engine"):
        replay.encode_resident(v32, torch.from_numpy(frames.reshape(-1, 38400)).to("cuda"))
