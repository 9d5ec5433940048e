"""Host logic of replay.replay_update (BASELINE configs[4]) on CPU: the orchestration -- row sharding, state assembly, value pass, per-row GAE,
flattening, legacy-numpy minibatch schedule with a partial last minibatch, loss records -- is run with STAND-INS for the two device
engines and for utils.compute_gae_batched that compute with the oracle, and compared with the reference trainer's own sequence
(train.py:171-207) written out from the oracle's pieces.  The device arithmetic itself is covered by tests/test_e_c5_replay_gpu.py."""
import sys
import types

import numpy as np
import pytest
import torch

from oracle import ppo_oracle as po


class FakeVaeDev:
    device = torch.device("cpu")

    def __init__(self, proj):
        self.proj = proj

    def encode(self, src, idx, n, out):                       # any deterministic frames -> z map will do for the orchestration
        assert idx is None and src.shape == (n, self.proj.shape[0])
        out.copy_(src @ self.proj)


class FakeVae:
    z_dim = 64

    def __init__(self, feat):
        self.dev = FakeVaeDev(torch.from_numpy((np.random.RandomState(1).standard_normal((feat, 64)) / np.sqrt(feat)).astype(np.float32)))
        self.feat = feat

    def _need_dev(self):
        return self.dev

    def _src_feat(self):
        return self.feat

    def _frames(self, arr, feat, what, keep_u8_ok=False):
        assert arr.dtype == np.uint8
        return torch.from_numpy(arr.reshape(len(arr), -1).astype(np.float32) / np.float32(255.0))


class FakePpoDev:
    device = torch.device("cpu")

    def __init__(self, oracle):
        self.o, self.losses = oracle, torch.zeros(5)

    def predict(self, states, m, noise, greedy, action, value):
        assert greedy and noise is None and states.shape[0] == m
        a, v = self.o.predict(states.numpy(), greedy=True)
        action.copy_(torch.from_numpy(np.atleast_2d(a)))
        value.copy_(torch.from_numpy(np.atleast_1d(v)))


class FakePpo:
    input_dim, num_actions = 67, 2

    def __init__(self, oracle):
        self.o, self.dev, self.train_step_counter, self.calls = oracle, FakePpoDev(oracle), 0, []

    def _need_dev(self):
        return self.dev

    def update_old_policy(self):
        self.o.update_old_policy()

    def _step_resident(self, s, a, r, adv, m_local, m_global):
        assert s.is_contiguous() and s.shape == (m_local, 67) and a.shape == (m_local, 2) and r.shape == adv.shape == (m_local,) and m_global == m_local
        L = self.o.train(s.numpy(), a.numpy(), r.numpy(), adv.numpy())
        self.calls.append(m_local)
        self.dev.losses = torch.tensor([L["policy_loss"], L["value_loss"], L["entropy_loss"], L["loss"], L["ratio_mean"]], dtype=torch.float32)


def fake_utils():
    mod = types.ModuleType("utils")

    def compute_gae_batched(rewards, values, terminals, gamma, lam, normalize=False):
        raw, ret, adv = [], [], []
        for r in range(len(rewards)):
            g = po.compute_gae(list(rewards[r]), list(values[r, :-1]), values[r, -1], list(np.asarray(terminals[r]).astype(bool)), gamma, lam)
            rr, aa = po.returns_and_normalized_advantages(g.copy(), values[r, :-1].astype(np.float64))
            raw.append(g), ret.append(rr), adv.append(aa)
        return np.array(raw), np.array(ret), np.array(adv)
    mod.compute_gae_batched = compute_gae_batched

    def gae_resident(rewards, values, terminals, gamma, lam):          # the device-resident form replay_update calls: tensors in, fp64 tensors out
        raw, ret, adv = compute_gae_batched(rewards, values.numpy(), terminals, gamma, lam, normalize=True)
        return torch.from_numpy(raw), torch.from_numpy(ret), torch.from_numpy(adv)
    mod.gae_resident = gae_resident
    return mod


def test_replay_update_orchestration_matches_the_trainer_sequence(monkeypatch):
    import replay
    monkeypatch.setitem(sys.modules, "utils", fake_utils())
    R, T, H, W = 3, 10, 4, 6
    rng = np.random.RandomState(11)
    frames = rng.randint(0, 256, (R, T + 1, H, W, 3), dtype=np.uint8)
    meas = rng.uniform(-1, 1, (R, T + 1, 3)).astype(np.float32)
    actions = np.stack([rng.uniform(-1, 1, (R, T)), rng.uniform(0, 1, (R, T))], axis=-1).astype(np.float32)
    rewards, dones = rng.uniform(0, 1, (R, T)), np.zeros((R, T))
    dones[2, -1] = 1
    hp = dict(learning_rate=1e-4, lr_decay=1.0, epsilon=0.2, value_scale=1.0, entropy_scale=0.01, initial_std=1.0)
    vae = FakeVae(H * W * 3)
    ppo = FakePpo(po.OraclePPO([67], po.ActionSpace(), seed=2, **hp))
    np.random.seed(5)
    out = replay.replay_update(vae, ppo, frames, meas, actions, rewards, dones, 0.99, 0.95, num_epochs=2, batch_size=8, encode_chunk=7)
    assert out["rows"] == (0, R) and out["samples_per_rank"] == 30
    assert ppo.calls == [8, 8, 8, 6] * 2 and ppo.train_step_counter == 8 and len(out["losses"]) == 8          # 30 samples: 8 + 8 + 8 + 6, twice

    # the reference trainer's sequence per trajectory (train.py:171-177), then its flattened minibatch loop (:193-204)
    o = po.OraclePPO([67], po.ActionSpace(), seed=2, **hp)
    z = (frames.reshape(R * (T + 1), -1).astype(np.float32) / np.float32(255.0)) @ vae.dev.proj.numpy()
    states = np.concatenate([z, meas.reshape(-1, 3)], axis=1).reshape(R, T + 1, 67)
    _, v = o.predict(states.reshape(-1, 67), greedy=True)
    v = v.reshape(R, T + 1)
    assert np.allclose(out["values"], v, rtol=1e-6, atol=1e-7)
    ret, adv = [], []
    for r in range(R):
        g = po.compute_gae(list(rewards[r]), list(v[r, :T]), v[r, T], list(dones[r].astype(bool)), 0.99, 0.95)
        rr, aa = po.returns_and_normalized_advantages(g, v[r, :T].astype(np.float64))
        ret.append(rr), adv.append(aa)
    assert np.allclose(out["returns"], ret, rtol=1e-6) and np.allclose(out["advantages"], adv, rtol=1e-5, atol=1e-6)
    np.random.seed(5)
    o.update_old_policy()
    s_flat, a_flat = states[:, :T].reshape(R * T, 67).astype(np.float32), actions.reshape(R * T, 2)
    logs = [o.train(s_flat[mb], a_flat[mb], np.concatenate(ret)[mb], np.concatenate(adv)[mb]) for mb in po.minibatch_schedule(R * T, 8, 2)]
    for want, got in zip(logs, out["losses"]):
        assert got["loss"] == pytest.approx(want["loss"], rel=1e-4, abs=1e-5) and got["prob_ratio"] == pytest.approx(want["ratio_mean"], rel=1e-4)


def test_replay_update_rejects_bad_shapes_and_uneven_shards(monkeypatch):
    import replay
    monkeypatch.setitem(sys.modules, "utils", fake_utils())
    ppo = FakePpo(po.OraclePPO([67], po.ActionSpace(), seed=2))
    vae = FakeVae(12)
    f = np.zeros((2, 4, 2, 2, 3), np.uint8)
    with pytest.raises(ValueError):
        replay.replay_update(vae, ppo, f, np.zeros((2, 3, 3)), np.zeros((2, 3, 2)), np.zeros((2, 3)), np.zeros((2, 3)))        # measurements need T + 1 steps
    with pytest.raises(ValueError):
        replay.replay_update(vae, ppo, f, np.zeros((2, 4, 2)), np.zeros((2, 3, 2)), np.zeros((2, 3)), np.zeros((2, 3)))        # 64 + 2 != 67 inputs
    from mi355 import dist as midist
    monkeypatch.setattr(midist, "world_size", lambda: 2)
    monkeypatch.setattr(midist, "rank", lambda: 0)
    with pytest.raises(ValueError):
        replay.replay_update(vae, ppo, np.zeros((3, 4, 2, 2, 3), np.uint8), np.zeros((3, 4, 3)), np.zeros((3, 3, 2)), np.zeros((3, 3)), np.zeros((3, 3)))
