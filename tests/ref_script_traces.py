"""Script-level pins (round 5): what the reference's own DRIVER SCRIPTS do to the classes of the hot path, recorded as call traces --

  * vae/train_vae.py:47-161   the whole `__main__` block: argument parsing, dataset split, model naming, constructor keywords, restart handling,
                              and the early-stopping loop (get_step_idx / evaluate / save / train_one_epoch);
  * train.py:22-216           `train(params, ...)`: load_vae + encode_state (the real vae_common.py), PPO constructor, predict per simulator step, the update block
                              :171-207 (bootstrap value, compute_gae, returns / normalisation, update_old_policy, shuffled minibatches of model.train) and the summaries.

The REAL files are executed here (runpy / importlib) against RECORDING stand-ins for `models` / `vae.models` / `ppo` / the CARLA environment; TensorFlow is the
shipped stub (carla-ppo_amd/tensorflow: the three symbols the scripts touch).  `tests/golden/make_script_traces.py` writes the traces to
tests/golden/train_vae_calls.json and tests/golden/train_py_calls.json; they travel to the GPU box, where /root/reference does not exist:

  * CPU: golden == trace(reference) where the checkout exists; golden == trace(the DROP-IN's carla-ppo_amd/vae/train_vae.py run against the same stand-ins);
    every traced call BINDS to the drop-in classes' signatures (inspect.signature: names, keywords, arity).
  * GPU: the traces are REPLAYED against the real drop-in objects (tests/test_h_script_traces_gpu.py).
Test infrastructure: nothing here is imported by the product.
"""
import importlib.util
import os
import runpy
import sys
import types

import numpy as np

from ref_call_chain import _desc

REFERENCE = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "carla-ppo_amd")

VAL_LOSSES = [5.0, 4.0, 3.0] + [3.5] * 10          # scripted validation losses: three improvements (three saves), then ten epochs without one -> early stop


def _rec(trace, obj, call, args, kwargs, **extra):
    e = {"obj": obj, "call": call, "args": [_desc(a) for a in args], "kwargs": {k: _desc(v) for k, v in sorted(kwargs.items())}}
    e.update(extra)
    trace.append(e)


class _Token:
    """A loss_fn token of the models module (vae/train_vae.py:98-100 passes the function object itself)."""
    def __init__(self, name):
        self.__name__ = name


def recording_models(trace, tmp_dir, name="vae.models"):
    """Stand-in for the `models` / `vae.models` module: ConvVAE / MlpVAE record constructor keywords and every method call; evaluate follows VAL_LOSSES;
    encode returns a vector filled with the running count of encode calls (so that a state's first element names the simulator step it came from)."""
    mod = types.ModuleType(name)
    mod.bce_loss, mod.bce_loss_v2, mod.mse_loss = _Token("bce_loss"), _Token("bce_loss_v2"), _Token("mse_loss")

    def make(cls_name):
        class _VAE:
            def __init__(self, *args, **kwargs):
                kw = dict(kwargs)
                if "loss_fn" in kw:
                    kw["loss_fn"] = getattr(kw["loss_fn"], "__name__", str(kw["loss_fn"]))
                _rec(trace, "vae", cls_name, args, kw)
                self.z_dim = kwargs.get("z_dim")
                self.model_dir = os.path.join(tmp_dir, "vae_model")
                self.checkpoint_dir, self.log_dir = os.path.join(self.model_dir, "checkpoints"), os.path.join(self.model_dir, "logs")
                self.dirs = [self.checkpoint_dir, self.log_dir]
                for d in self.dirs:
                    os.makedirs(d, exist_ok=True)
                self._epoch, self._evals, self._encodes = 0, 0, 0

            def init_session(self, sess=None, init_logging=True):
                _rec(trace, "vae", "init_session", (), {"init_logging": init_logging})

            def load_latest_checkpoint(self, *args, **kwargs):
                _rec(trace, "vae", "load_latest_checkpoint", args, kwargs)
                return True

            def get_step_idx(self):
                _rec(trace, "vae", "get_step_idx", (), {}, returns=self._epoch)
                return self._epoch

            def evaluate(self, *args, **kwargs):
                v = VAL_LOSSES[min(self._evals, len(VAL_LOSSES) - 1)]
                self._evals += 1
                _rec(trace, "vae", "evaluate", args, kwargs, returns=[v, 0.25])
                return [v, 0.25]

            def train_one_epoch(self, *args, **kwargs):
                _rec(trace, "vae", "train_one_epoch", args, kwargs)
                self._epoch += 1

            def save(self, *args, **kwargs):
                _rec(trace, "vae", "save", args, kwargs)

            def encode(self, *args, **kwargs):
                _rec(trace, "vae", "encode", args, kwargs)
                self._encodes += 1
                return np.full((len(args[0]), self.z_dim), float(self._encodes - 1), np.float32)
        _VAE.__name__ = cls_name
        return _VAE
    mod.ConvVAE, mod.MlpVAE = make("ConvVAE"), make("MlpVAE")
    return mod


# ---------------------------------------------------------------------------------------------------------------------------
# vae/train_vae.py
def make_dataset(root, n=40):
    """<root>/rgb/*.png and <root>/segmentation/*.png: n RGBA frames of the reference's 80 x 160 size (CarlaEnv/collect_data.py:192-197 writes RGBA PNGs)."""
    from PIL import Image
    rng = np.random.RandomState(9)
    for sub, hi in (("rgb", 256), ("segmentation", 13)):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
        for i in range(n):
            a = rng.randint(0, hi, (80, 160, 4), dtype=np.uint8)
            Image.fromarray(a, "RGBA").save(os.path.join(root, sub, "%d.png" % i))
    return root


TRAIN_VAE_CASES = {
    "rgb_cnn_restart": ["--z_dim", "64", "--batch_size", "4", "-restart"],
    "seg_mlp_continue": ["--use_segmentation_as_target", "1", "--model_type", "mlp", "--loss_type", "bce_v2", "--z_dim", "16", "--beta", "4", "--kl_tolerance", "0.5",
                         "--learning_rate", "0.0002", "--lr_decay", "0.98", "--batch_size", "2"],
}


def _with_modules(mods, fn):
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        return fn()
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _tf_stub():
    spec = importlib.util.spec_from_file_location("tensorflow", os.path.join(DROPIN, "tensorflow", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def trace_train_vae(which, case, tmp_dir):
    """which = "reference": runs /root/reference/vae/train_vae.py as __main__; "dropin": carla-ppo_amd/vae/train_vae.py's main() -- both against the recording module,
    the same argv and the same dataset.  The drop-in is run with --host_float_frames (the reference's host-side /255; without it the RGB tables stay uint8 until the device)."""
    trace = []
    data = make_dataset(os.path.join(tmp_dir, "data"))
    argv = ["--dataset", data] + TRAIN_VAE_CASES[case]
    models = recording_models(trace, tmp_dir, "models" if which == "reference" else "vae.models")
    cwd, old_argv = os.getcwd(), sys.argv
    os.chdir(tmp_dir)
    try:
        if which == "reference":
            path = os.path.join(REFERENCE, "vae", "train_vae.py")
            if not os.path.exists(path):
                raise FileNotFoundError(path)
            sys.argv = [path] + argv
            plt = types.ModuleType("matplotlib.pyplot")
            _with_modules({"models": models, "tensorflow": _tf_stub(), "matplotlib.pyplot": plt}, lambda: runpy.run_path(path, run_name="__main__"))
        else:
            path = os.path.join(DROPIN, "vae", "train_vae.py")
            pkg = types.ModuleType("vae")
            pkg.models = models
            pkg.__path__ = [os.path.join(DROPIN, "vae")]

            def run():
                spec = importlib.util.spec_from_file_location("_dropin_train_vae", path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                mod.main(argv + ["--host_float_frames"])
            _with_modules({"vae": pkg, "vae.models": models}, run)
    finally:
        os.chdir(cwd)
        sys.argv = old_argv
    return _scrub(trace, tmp_dir)


def _scrub(trace, tmp_dir):
    """Paths under the temporary directory are not part of the pin."""
    def fix(v):
        if isinstance(v, str):
            return v.replace(tmp_dir, "<tmp>")
        if isinstance(v, dict):
            return {k: fix(x) for k, x in v.items()}
        if isinstance(v, list):
            return [fix(x) for x in v]
        return v
    return fix(trace)


# ---------------------------------------------------------------------------------------------------------------------------
# train.py
class ActionSpace:
    shape, low, high = (2,), np.array([-1.0, 0.0], np.float32), np.array([1.0, 1.0], np.float32)      # CarlaEnv/carla_lap_env.py:136


TRAIN_PARAMS = dict(learning_rate=1e-4, lr_decay=1.0, discount_factor=0.99, gae_lambda=0.95, ppo_epsilon=0.2, initial_std=1.0, value_scale=1.0, entropy_scale=0.01,
                    horizon=8, num_epochs=2, num_episodes=1, batch_size=3, vae_model="vae/models/seg_bce_cnn_zdim64_beta1_kl_tolerance0.0_data/", vae_model_type=None,
                    vae_z_dim=None, synchronous=True, fps=30, action_smoothing=0.0, model_name="agent", reward_fn="reward_speed_centering_angle_multiply", seed=0,
                    eval_interval=5, record_eval=True)
EPISODE_STEPS = 13                                  # the stub environment ends the episode after 13 steps: one full horizon of 8, then 5 steps with the terminal state


def trace_train_py(tmp_dir):
    """Runs the REAL /root/reference/train.py::train(params, start_carla=False, restart=True) for one episode against recording stand-ins."""
    path = os.path.join(REFERENCE, "train.py")
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    trace = []
    models = recording_models(trace, tmp_dir)

    class RecordingPPO:
        def __init__(self, *args, **kwargs):
            a = list(args)
            if len(a) > 1:
                a[1] = "action_space"
            _rec(trace, "ppo", "PPO", a, kwargs)
            self.model_dir = os.path.join(tmp_dir, "ppo_model")
            self.checkpoint_dir, self.log_dir, self.video_dir = (os.path.join(self.model_dir, d) for d in ("checkpoints", "logs", "videos"))
            self.dirs = [self.checkpoint_dir, self.log_dir, self.video_dir]
            for d in self.dirs:
                os.makedirs(d, exist_ok=True)
            self._episode, self._train_steps, self._predicts = 0, 0, 0

        def _m(name, ret=None):
            def f(self, *args, **kwargs):
                _rec(trace, "ppo", name, args, kwargs)
                return ret
            return f
        init_session, load_latest_checkpoint, save, update_old_policy = _m("init_session"), _m("load_latest_checkpoint", True), _m("save"), _m("update_old_policy")
        write_value_to_summary = _m("write_value_to_summary")

        def write_dict_to_summary(self, name, d, step):
            _rec(trace, "ppo", "write_dict_to_summary", (name, sorted(d.keys()), step), {})

        def write_episodic_summaries(self):
            _rec(trace, "ppo", "write_episodic_summaries", (), {})
            self._episode += 1

        def get_episode_idx(self):
            return self._episode

        def get_train_step_idx(self):
            return self._train_steps

        def predict(self, *args, **kwargs):
            _rec(trace, "ppo", "predict", args, kwargs)
            self._predicts += 1
            return np.array([0.1, 0.6], np.float32), np.float32(0.5 + 0.01 * self._predicts)

        def train(self, *args, **kwargs):
            _rec(trace, "ppo", "train", args, kwargs, rows=[int(x) for x in np.asarray(args[0])[:, 0]])       # which simulator steps make up this minibatch
            self._train_steps += 1

    class Env:
        """CarlaLapEnv as train.py uses it (CarlaEnv/carla_lap_env.py): reset / step / render / seed, action_space, the episode statistics, and what encode_state reads."""
        action_space = ActionSpace()

        class _Vehicle:
            class _Control:
                steer, throttle = -0.25, 0.5
            control = _Control()

            def get_speed(self):
                return 12.5
        vehicle = _Vehicle()

        def __init__(self, **kwargs):
            kw = {k: (v if isinstance(v, (bool, int, float, str, tuple)) or v is None else type(v).__name__) for k, v in kwargs.items()}
            _rec(trace, "env", "CarlaEnv", (), kw)
            self.encode_state_fn, self.extra_info = kwargs["encode_state_fn"], []
            self.distance_traveled, self.speed_accum, self.step_count, self.center_lane_deviation = 100.0, 50.0, 0, 2.0
            self.observation = np.random.RandomState(3).randint(0, 256, (80, 160, 3), dtype=np.uint8)

        def seed(self, s):
            _rec(trace, "env", "seed", (s,), {})

        def reset(self):
            self.step_count = 0
            return self.encode_state_fn(self)

        def step(self, action):
            self.step_count += 1
            return self.encode_state_fn(self), 1.0 + 0.1 * self.step_count, self.step_count >= EPISODE_STEPS, {"closed": False}

        def render(self):
            pass
    lap = types.ModuleType("CarlaEnv.carla_lap_env")
    lap.CarlaLapEnv = Env
    wrappers = types.ModuleType("CarlaEnv.wrappers")
    wrappers.vector = lambda v: np.array([v[0], v[1], v[2]])
    pkg_c, pkg_v = types.ModuleType("CarlaEnv"), types.ModuleType("vae")
    pkg_c.wrappers, pkg_c.carla_lap_env, pkg_v.models = wrappers, lap, models
    ppo_mod = types.ModuleType("ppo")
    ppo_mod.PPO = RecordingPPO
    rf = types.ModuleType("reward_functions")
    rf.reward_functions = {"reward_speed_centering_angle_multiply": "reward_fn"}
    re_mod = types.ModuleType("run_eval")

    def run_eval(env, model, video_filename=None):
        _rec(trace, "script", "run_eval", (), {"video_filename": os.path.basename(video_filename)})
        env.step_count = 7                           # (the real run_eval drives an evaluation episode: the statistics train.py logs afterwards are non-zero)
        return 1.5
    re_mod.run_eval = run_eval
    mods = {"tensorflow": _tf_stub(), "cv2": types.ModuleType("cv2"), "CarlaEnv": pkg_c, "CarlaEnv.wrappers": wrappers, "CarlaEnv.carla_lap_env": lap, "vae": pkg_v,
            "vae.models": models, "ppo": ppo_mod, "reward_functions": rf, "run_eval": re_mod}

    def run():
        # the REAL utils.py (compute_gae :45-50) and vae_common.py of the reference, loaded under their own names
        class _AnyTf(types.ModuleType):              # the reference's utils.py names tf.tanh in a default argument (:25): while THAT file loads, any tf symbol resolves
            def __getattr__(self, name):
                return None
        strict_tf = sys.modules["tensorflow"]
        for name in ("utils", "vae_common"):
            spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE, name + ".py"))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            sys.modules["tensorflow"] = _AnyTf("tensorflow") if name == "utils" else strict_tf
            spec.loader.exec_module(mod)
        sys.modules["tensorflow"] = strict_tf          # train.py itself runs against the SHIPPED stub
        real_gae = sys.modules["utils"].compute_gae

        def compute_gae(*args, **kwargs):
            out = real_gae(*args, **kwargs)
            _rec(trace, "utils", "compute_gae", args, kwargs, returns=_desc(np.asarray(out)))
            return out
        sys.modules["utils"].compute_gae = compute_gae
        spec = importlib.util.spec_from_file_location("_ref_train", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.train(dict(TRAIN_PARAMS), start_carla=False, restart=True)
    saved_extra = {k: sys.modules.get(k) for k in ("utils", "vae_common")}
    cwd = os.getcwd()
    os.chdir(tmp_dir)
    try:
        _with_modules(mods, run)
    finally:
        os.chdir(cwd)
        for k, v in saved_extra.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return _scrub(trace, tmp_dir)
