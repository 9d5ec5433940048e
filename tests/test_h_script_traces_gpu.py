"""The recorded call traces of the reference's driver scripts REPLAYED against the real drop-in on the GPU (round 5; fixtures
tests/golden/train_vae_calls.json / train_py_calls.json written from the REAL /root/reference/vae/train_vae.py and train.py by tests/golden/make_script_traces.py):
every call the unchanged scripts make is issued, with the recorded positional / keyword layout, on `vae.models.ConvVAE / MlpVAE`, `ppo.PPO` and `utils.compute_gae`,
and what comes back is what the scripts' own code needs next (the unpacking at vae/train_vae.py:147, the asserts at train.py:186-189, the counters their loops test)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_script_traces import materialise  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _frames_like(d, rng):
    a = materialise(d)
    if isinstance(a, np.ndarray) and a.ndim == 4:
        hi = 256 if a.shape[-1] == 3 else 13
        return (rng.randint(0, hi, a.shape).astype(np.float32) / (hi - 1.0)).astype(a.dtype)
    return a


@pytest.mark.parametrize("case", ["rgb_cnn_restart", "seg_mlp_continue"])
def test_replay_of_the_train_vae_trace_on_the_dropin(tmp_path, case, monkeypatch):
    from vae import models as vm
    trace = json.load(open(os.path.join(GOLDEN, "train_vae_calls.json")))[case]
    rng = np.random.RandomState(1)
    monkeypatch.chdir(tmp_path)                          # the script's model_dir is relative ("models/<name>", vae/train_vae.py:116)
    np.random.seed(0)                                    # vae/train_vae.py:70
    vae, cache, n_saves, last_eval = None, {}, 0, None
    for c in trace:
        name = c["call"]
        kw = {k: materialise(v) for k, v in c["kwargs"].items()}
        if name in ("ConvVAE", "MlpVAE"):
            kw["loss_fn"] = getattr(vm, kw["loss_fn"])   # the token the script picked from the module (:98-100)
            vae = getattr(vm, name)(*[materialise(a) for a in c["args"]], **kw)
            assert isinstance(vae, getattr(vm, name)) and int(vae.z_dim) == int(kw["z_dim"])
            for d in vae.dirs:                           # the constructor creates the directories (vae/models.py:156-159); :130-132 re-creates them after rmtree
                assert os.path.isdir(d)
            assert vae.model_dir.startswith("models" + os.sep)
            continue
        # frame tables: one array per recorded (shape, dtype), as the script passes the same arrays every epoch
        args = []
        for a in c["args"]:
            key = json.dumps(a, sort_keys=True)
            if key not in cache:
                cache[key] = _frames_like(a, rng)
            args.append(cache[key])
        out = getattr(vae, name)(*args, **kw)
        if name == "get_step_idx":
            assert isinstance(out, int) and out == c["returns"]          # the epoch counter the script prints and the recording stand-in counted the same way
        elif name == "evaluate":
            val_loss, kl = out                                             # `val_loss, _ = vae.evaluate(...)` (:147)
            assert np.isfinite(val_loss) and np.isfinite(kl) and val_loss > 0
            last_eval = val_loss
        elif name == "save":
            n_saves += 1
            assert os.path.exists(os.path.join(vae.checkpoint_dir, "checkpoint"))
        elif name == "load_latest_checkpoint":
            assert out in (True, False, None)                              # nothing saved yet in a fresh directory: the script ignores the result (:136)
    assert n_saves == 3 and last_eval is not None
    # a continued run finds what this one saved (vae/train_vae.py:135-136 on the next invocation)
    again = type(vae)(source_shape=vae.source_shape, target_shape=vae.target_shape, z_dim=int(vae.z_dim), model_dir=vae.model_dir,
                      **({"encoder_sizes": vae.encoder_sizes, "decoder_sizes": vae.decoder_sizes} if case.startswith("seg_mlp") else {}))
    again.init_session(init_logging=False)
    assert again.load_latest_checkpoint() is True and again.get_step_idx() >= 1


def test_replay_of_the_train_py_trace_on_the_dropin(tmp_path, monkeypatch):
    import ppo as ppo_mod
    import utils as utils_mod
    from vae import models as vm
    g = json.load(open(os.path.join(GOLDEN, "train_py_calls.json")))
    p, trace = g["params"], g["calls"]
    monkeypatch.chdir(tmp_path)
    np.random.seed(p["seed"])

    class Box:                                           # env.action_space as train.py hands it over (gym.spaces.Box: .shape / .low / .high are all ppo.py reads)
        shape, low, high = (2,), np.array([-1.0, 0.0], np.float32), np.array([1.0, 1.0], np.float32)
    rng = np.random.RandomState(2)
    vae = model = None
    n_train = n_pred = 0
    horizon_batch = None
    for c in trace:
        obj, name = c["obj"], c["call"]
        args, kw = [materialise(a) for a in c["args"]], {k: materialise(v) for k, v in c["kwargs"].items()}
        if obj in ("env", "script"):
            continue
        if obj == "vae":
            if name == "ConvVAE":
                kw["model_dir"] = str(tmp_path / "vae_seg")               # (the trained VAE the script loads lives outside this test: a fresh one of the same shape)
                vae = vm.ConvVAE(*args, **kw)
                assert vae.training is False
            elif name == "init_session":
                vae.init_session(**kw)
            elif name == "load_latest_checkpoint":
                assert vae.load_latest_checkpoint() in (False, None)      # empty directory: vae_common.py:25-26 would raise "Failed to load VAE" -- the contract it relies on
            elif name == "encode":
                frame = rng.randint(0, 256, (80, 160, 3)).astype(np.float32) / 255.0
                z = vae.encode([frame])                                   # vae_common.py:48: a LIST holding one frame
                assert z.shape == (1, 64) and z.dtype == np.float32
            continue
        if obj == "utils":
            adv = utils_mod.compute_gae(*args, **kw)
            want = np.array(c["returns"]["ndarray"], np.float64)
            assert isinstance(adv, np.ndarray) and adv.dtype == np.float64 and np.array_equal(adv, want)      # bit-exact with the REFERENCE's own compute_gae (scipy.signal.lfilter) on the recorded inputs
            rewards, values = np.asarray(args[0], np.float64), np.asarray([float(v) for v in args[1]])
            returns = adv + values                                        # train.py:176-177
            advn = (adv - adv.mean()) / (adv.std() + 1e-8)
            horizon_batch = (returns, advn)
            continue
        # ---- ppo ----
        if name == "PPO":
            args[1] = Box()
            model = ppo_mod.PPO(*args, **kw)
            for d in (model.checkpoint_dir, model.log_dir, model.video_dir):
                assert os.path.isdir(d)
        elif name == "predict":
            state = np.concatenate([rng.standard_normal(64) * 0.5, [-0.25, 0.5, 12.5]])      # np.append(z, measurements): float64 [67] (vae_common.py:59)
            action, value = model.predict(state, **kw)
            assert np.shape(action) == (2,) and np.ndim(value) == 0 and Box.low[0] <= action[0] <= Box.high[0] and Box.low[1] <= action[1] <= Box.high[1]
            n_pred += 1
        elif name == "train":
            m = len(c["rows"])
            T = len(horizon_batch[0])
            rows = np.array(c["rows"]) % T
            states = rng.standard_normal((T, 67)) * 0.5
            actions = np.tile(np.array([0.1, 0.6], np.float32), (T, 1))
            assert [a.shape if hasattr(a, "shape") else None for a in args][:2] == [(m, 67), (m, 2)]       # what the script passed
            out = model.train(states[rows], actions[rows], horizon_batch[0][rows], horizon_batch[1][rows])
            assert out is None
            n_train += 1
            assert model.get_train_step_idx() == n_train
        elif name == "write_episodic_summaries":
            before = model.get_episode_idx()
            model.write_episodic_summaries()
            assert model.get_episode_idx() == before + 1                   # the loop condition of train.py:113
        elif name == "save":
            model.save()
            assert os.path.exists(os.path.join(model.checkpoint_dir, "checkpoint"))
        elif name == "write_dict_to_summary":
            model.write_dict_to_summary(args[0], {k: p[k] for k in args[1]}, args[2])      # (the trace keeps the dictionary's keys: train.py:111 passes its params)
        else:
            getattr(model, name)(*args, **kw)
    assert n_train == 10 and n_pred == g["episode_steps"] + 2               # 13 simulator steps + the two bootstrap values (train.py:172)
    losses = model.dev.losses.cpu().numpy()
    assert np.all(np.isfinite(losses))
