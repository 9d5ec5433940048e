"""Script-level pins on the CPU (round 5; tests/ref_script_traces.py): the call traces of the reference's own vae/train_vae.py and train.py are
(1) what the REAL files produce where the checkout exists, (2) reproduced call for call by the drop-in's vae/train_vae.py, (3) bindable, call by call, to the drop-in
classes' signatures, and (4) the shipped `tensorflow` stub is all the scripts touch."""
import inspect
import json
import os
import tempfile

import numpy as np
import pytest

import ref_script_traces as rs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def _norm(x):
    return json.loads(json.dumps(x))


@pytest.mark.skipif(not os.path.exists(rs.REFERENCE), reason="needs the reference checkout (this container only)")
def test_goldens_are_what_the_real_scripts_do():
    g = _golden("train_vae_calls.json")
    for case in rs.TRAIN_VAE_CASES:
        with tempfile.TemporaryDirectory() as t:
            assert _norm(rs.trace_train_vae("reference", case, t)) == g[case], case
    gp = _golden("train_py_calls.json")
    with tempfile.TemporaryDirectory() as t:
        assert _norm(rs.trace_train_py(t)) == gp["calls"]
    assert gp["params"] == _norm(rs.TRAIN_PARAMS) and gp["episode_steps"] == rs.EPISODE_STEPS


@pytest.mark.parametrize("case", sorted(rs.TRAIN_VAE_CASES))
def test_dropin_train_vae_script_issues_the_reference_call_sequence(case):
    """carla-ppo_amd/vae/train_vae.py's main() against the recording `vae.models`, same argv and dataset as the reference run: the SAME trace -- model naming,
    constructor keywords (numpy shapes, loss token, hyper-parameters), restart / continue handling, and the early-stopping loop call for call (13 evaluations, 3 saves,
    12 epochs for the scripted validation losses)."""
    g = _golden("train_vae_calls.json")[case]
    with tempfile.TemporaryDirectory() as t:
        tr = _norm(rs.trace_train_vae("dropin", case, t))
    assert tr == g
    calls = [c["call"] for c in g]
    assert calls.count("evaluate") == 13 and calls.count("save") == 3 and calls.count("train_one_epoch") == 12
    assert ("load_latest_checkpoint" in calls) == ("-restart" not in rs.TRAIN_VAE_CASES[case])


def materialise(d):
    """A description (ref_call_chain._desc) back into a value of that kind."""
    if isinstance(d, dict):
        if "ndarray" in d:
            return np.array(d["ndarray"], dtype=d["dtype"])
        if "ndarray_shape" in d:
            return np.zeros(d["ndarray_shape"], dtype=d["dtype"])
        if "list" in d:
            return [materialise(x) for x in d["list"]]
        if "scalar" in d:
            return np.dtype(d["dtype"]).type(d["scalar"])
        return None
    return d


def test_every_traced_call_binds_to_the_dropin_signatures():
    """Each call the real scripts make -- positional / keyword layout exactly as recorded -- is accepted by the signature of the drop-in class / function of the same
    name (no GPU: inspect.signature(...).bind): a renamed keyword or a missing default in the mirror would break an unchanged reference script."""
    import ppo as ppo_mod
    import utils as utils_mod
    from vae import models as vm
    targets = {("vae", "ConvVAE"): vm.ConvVAE.__init__, ("vae", "MlpVAE"): vm.MlpVAE.__init__, ("ppo", "PPO"): ppo_mod.PPO.__init__, ("utils", "compute_gae"): utils_mod.compute_gae}
    seen = set()
    traces = list(_golden("train_vae_calls.json").values()) + [_golden("train_py_calls.json")["calls"]] + [v["calls"] for v in _golden("vae_common_calls.json").values()]
    for tr in traces:
        vae_cls = None
        for c in tr:
            obj, name = c.get("obj", "vae"), c["call"]
            args, kwargs = [materialise(a) for a in c["args"]], {k: materialise(v) for k, v in c["kwargs"].items()}
            if obj in ("env", "script"):
                continue
            if (obj, name) in targets:
                fn = targets[(obj, name)]
                if obj == "vae":
                    vae_cls = getattr(vm, name)
            elif obj == "vae":
                fn = getattr(vae_cls or vm.ConvVAE, name)
            elif obj == "ppo":
                fn = getattr(ppo_mod.PPO, name)
            else:
                raise AssertionError((obj, name))
            sig = inspect.signature(fn)
            if obj == "utils":
                sig.bind(*args, **kwargs)
            else:
                sig.bind(None, *args, **kwargs)        # (self)
            seen.add((obj, name))
    for must in (("vae", "ConvVAE"), ("vae", "MlpVAE"), ("vae", "evaluate"), ("vae", "train_one_epoch"), ("vae", "save"), ("vae", "get_step_idx"), ("vae", "encode"),
                 ("ppo", "PPO"), ("ppo", "predict"), ("ppo", "train"), ("ppo", "update_old_policy"), ("ppo", "write_value_to_summary"), ("ppo", "write_dict_to_summary"),
                 ("ppo", "write_episodic_summaries"), ("ppo", "save"), ("utils", "compute_gae")):
        assert must in seen, must


def test_minibatch_schedule_of_the_traced_update_block():
    """train.py:192-207 as recorded: per horizon chunk one update_old_policy, then num_epochs x ceil(T / batch) minibatches that partition the chunk's steps (the last one
    partial, kept), in the order of the legacy-numpy shuffle seeded by train.py:52 -- the rule carla-ppo_amd/replay.py applies per epoch (np.random.shuffle(arange(n)),
    slices of batch_size, the partial tail kept)."""
    g = _golden("train_py_calls.json")
    p, calls = g["params"], g["calls"]
    chunks, cur = [], None
    for c in calls:
        if c["call"] == "update_old_policy":
            cur = []
            chunks.append(cur)
        elif c["call"] == "train":
            cur.append(c["rows"])
    assert len(chunks) == 2
    np.random.seed(p["seed"])                          # train.py:52; nothing else on the traced path draws from numpy's global generator
    first = 0
    for T, mbs in zip((p["horizon"], g["episode_steps"] - p["horizon"]), chunks):
        want = []
        for _ in range(p["num_epochs"]):
            idx = np.arange(T)
            np.random.shuffle(idx)
            n_steps = int(np.ceil(T / p["batch_size"]))
            want += [[int(first + j) for j in idx[i * p["batch_size"]:(i + 1) * p["batch_size"]]] for i in range(n_steps)]
        assert mbs == want
        first += T
    src = inspect.getsource(__import__("replay").replay_update)
    assert "np.random.shuffle(indices)" in src and "np.ceil(n_loc / mb_loc)" in src


def test_tensorflow_stub_is_exactly_what_the_scripts_touch():
    import importlib
    import sys
    saved = sys.modules.pop("tensorflow", None)
    try:
        tf = importlib.import_module("tensorflow")     # carla-ppo_amd/ is ahead on sys.path (tests/conftest.py), as INTEGRATION.md section 2 puts it on PYTHONPATH
        assert "carla-ppo_amd" in tf.__file__
        tf.random.set_random_seed(7)
        assert tf.get_graph_seed() == 7
        # ... and it is consumed: a model built without an explicit seed takes it, whatever numpy's global state is (a script that seeds only TensorFlow)
        from mi355.init import seed_from_numpy_state
        st = np.random.get_state()
        try:
            np.random.seed(1); a = seed_from_numpy_state()
            np.random.seed(2); b = seed_from_numpy_state()
            tf.random.set_random_seed(8); c = seed_from_numpy_state()
            tf.random.set_random_seed(None); np.random.seed(1); d = seed_from_numpy_state(); np.random.seed(2); e = seed_from_numpy_state()
        finally:
            np.random.set_state(st)
        assert a == b and c != a and d != e
        assert tf.reset_default_graph() is None
        with pytest.raises(AttributeError, match="stub"):
            tf.layers
    finally:
        sys.modules.pop("tensorflow", None)
        if saved is not None:
            sys.modules["tensorflow"] = saved
    if os.path.exists(rs.REFERENCE):                   # the scripts' own uses, line by line
        import re
        for f in ("train.py", "run_eval.py", "vae/train_vae.py"):
            uses = set(re.findall(r"\btf\.[\w.]+", open(os.path.join(rs.REFERENCE, f)).read()))
            assert uses <= {"tf.random.set_random_seed", "tf.reset_default_graph"}, (f, uses)
