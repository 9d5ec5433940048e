"""the call chain of the reference's vae_common.py (load_vae :6-27, create_encode_state_fn / encode_state :33-62) as a
recorded TRACE, so that the drop-in can be driven exactly the way the unchanged reference scripts drive it -- on the GPU box, where
/root/reference does not exist.

  * RecordingModels: a stand-in for the `vae.models` module that records every constructor keyword and method call it receives.
  * trace_reference(): imports the REAL /root/reference/vae_common.py (with `vae.models` -> RecordingModels and a stub for
    `CarlaEnv.wrappers`, which needs the carla package) and records what it does.  Only available where the reference checkout is.
  * restated_load_vae / restated_encode_state: the same chain restated here (test infrastructure, cites the reference lines).
  * tests/golden/vae_common_calls.json (written by tests/golden/make_call_trace.py from trace_reference()) pins the restatement to the
    reference: test_host_logic.py checks golden == trace(restatement) everywhere and golden == trace(reference) where the checkout exists;
    test_b_c1_epoch_gpu.py then runs the restatement against the real drop-in on the GPU and compares with the oracle.
"""
import importlib.util
import os
import re
import sys
import types

import numpy as np

REFERENCE = "/root/reference"


def _desc(v):
    """JSON-able description of an argument: values for small things, shape / dtype for frames."""
    if isinstance(v, np.ndarray):
        if v.size <= 8:
            return {"ndarray": v.tolist(), "dtype": str(v.dtype)}
        return {"ndarray_shape": list(v.shape), "dtype": str(v.dtype)}
    if isinstance(v, (list, tuple)):
        return {"list": [_desc(x) for x in v]}
    if isinstance(v, (bool, int, float, str)) or v is None:
        return v
    if isinstance(v, np.generic):                       # numpy scalars (a value estimate, a reward): the value and its dtype
        return {"scalar": v.item(), "dtype": str(v.dtype)}
    return {"type": type(v).__name__}


class RecordingModels(types.ModuleType):
    """Fake `vae.models`: ConvVAE / MlpVAE record how they are used; encode returns a fixed ramp so the consumer's indexing is visible."""

    def __init__(self):
        super().__init__("vae.models")
        self.trace = []
        trace = self.trace

        def make(cls_name):
            class _VAE:
                def __init__(self, *args, **kwargs):
                    trace.append({"call": cls_name, "args": [_desc(a) for a in args], "kwargs": {k: _desc(v) for k, v in sorted(kwargs.items())}})
                    self.z_dim = kwargs.get("z_dim")

                def init_session(self, *args, **kwargs):
                    trace.append({"call": "init_session", "args": [_desc(a) for a in args], "kwargs": {k: _desc(v) for k, v in sorted(kwargs.items())}})

                def load_latest_checkpoint(self, *args, **kwargs):
                    trace.append({"call": "load_latest_checkpoint", "args": [_desc(a) for a in args], "kwargs": {k: _desc(v) for k, v in sorted(kwargs.items())}})
                    return True

                def encode(self, *args, **kwargs):
                    trace.append({"call": "encode", "args": [_desc(a) for a in args], "kwargs": {k: _desc(v) for k, v in sorted(kwargs.items())}})
                    n = len(args[0])
                    return np.tile(np.arange(self.z_dim, dtype=np.float32), (n, 1))
            _VAE.__name__ = cls_name
            return _VAE
        self.ConvVAE, self.MlpVAE = make("ConvVAE"), make("MlpVAE")


class StubEnv:
    """What encode_state reads from the CarlaEnv (vae_common.py:47-56): observation (uint8 camera frame) and the vehicle's control / speed."""

    class _Vehicle:
        class _Control:
            steer, throttle = -0.25, 0.5
        control = _Control()

        def get_speed(self):
            return 12.5

        def get_forward_vector(self):
            return (1.0, 0.0, 0.0)
    vehicle = _Vehicle()

    def __init__(self, frame_u8):
        self.observation = frame_u8


# ---------------------------------------------------------------------------------------------------------------------------
# restatement of /root/reference/vae_common.py (test infrastructure)
def restated_load_vae(models, model_dir, z_dim=None, model_type=None):
    """vae_common.py:6-27."""
    if z_dim is None:
        z_dim = int(re.findall(r"zdim(\d+)", model_dir)[0])                                   # :12
    if model_type is None:
        model_type = "mlp" if "mlp" in model_dir else "cnn"                                   # :13
    VAEClass = models.MlpVAE if model_type == "mlp" else models.ConvVAE                       # :14
    target_depth = 1 if "seg_" in model_dir else 3                                            # :15
    vae = VAEClass(source_shape=np.array([80, 160, 3]), target_shape=np.array([80, 160, target_depth]),
                   z_dim=z_dim, models_dir="vae", model_dir=model_dir, training=False)        # :18-23
    vae.init_session(init_logging=False)                                                      # :24
    if not vae.load_latest_checkpoint():                                                      # :25-26
        raise Exception("Failed to load VAE")
    return vae


def restated_encode_state(vae, env, measurements_to_include=("steer", "throttle", "speed")):
    """vae_common.py:29-31 (preprocess_frame) and :45-61 (encode_state)."""
    frame = env.observation.astype(np.float32) / 255.0                                        # :30
    encoded_state = vae.encode([frame])[0]                                                    # :48
    measurements = []
    if "steer" in measurements_to_include:
        measurements.append(env.vehicle.control.steer)                                        # :52
    if "throttle" in measurements_to_include:
        measurements.append(env.vehicle.control.throttle)                                     # :53
    if "speed" in measurements_to_include:
        measurements.append(env.vehicle.get_speed())                                          # :54
    return np.append(encoded_state, measurements)                                             # :59


# ---------------------------------------------------------------------------------------------------------------------------
def _frame():
    return np.random.RandomState(3).randint(0, 256, (80, 160, 3), dtype=np.uint8)


def _finish(models, state):
    return {"calls": models.trace, "state": {"shape": list(state.shape), "dtype": str(state.dtype), "head": [float(x) for x in state[:3]],
                                              "tail": [float(x) for x in state[-3:]]}}


def trace_restatement(model_dir):
    models = RecordingModels()
    vae = restated_load_vae(models, model_dir)
    return _finish(models, restated_encode_state(vae, StubEnv(_frame())))


def trace_reference(model_dir):
    """Runs the REAL vae_common.py of the reference checkout against the recording module (container with /root/reference only)."""
    path = os.path.join(REFERENCE, "vae_common.py")
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    models = RecordingModels()
    wrappers = types.ModuleType("CarlaEnv.wrappers")
    wrappers.vector = lambda v: np.array([v[0], v[1], v[2]])       # CarlaEnv/wrappers.py needs `carla`; only `vector` is imported (vae_common.py:3)
    pkg_c, pkg_v = types.ModuleType("CarlaEnv"), types.ModuleType("vae")
    pkg_c.wrappers, pkg_v.models = wrappers, models
    saved = {k: sys.modules.get(k) for k in ("CarlaEnv", "CarlaEnv.wrappers", "vae", "vae.models")}
    sys.modules.update({"CarlaEnv": pkg_c, "CarlaEnv.wrappers": wrappers, "vae": pkg_v, "vae.models": models})
    try:
        spec = importlib.util.spec_from_file_location("_ref_vae_common", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        vae = mod.load_vae(model_dir)
        state = mod.create_encode_state_fn(vae, ["steer", "throttle", "speed"])(StubEnv(_frame()))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return _finish(models, state)


MODEL_DIRS = ["vae/models/seg_bce_cnn_zdim64_beta1_kl_tolerance0.0_data", "vae/models/bce_cnn_zdim64_beta1_kl_tolerance0.0_data",
              "vae/models/bce_mlp_zdim10_beta1_kl_tolerance0.0_data"]
