#!/usr/bin/env python3
"""Writes tests/golden/train_vae_calls.json and tests/golden/train_py_calls.json: what the reference's own driver scripts (vae/train_vae.py's __main__ block,
train.py's train()) do to the classes of the hot path, recorded by running the REAL files against recording stand-ins (tests/ref_script_traces.py).
Run in the container that holds the reference checkout; the fixtures travel to the GPU box."""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "carla-ppo_amd"))
import ref_script_traces as rs  # noqa: E402

out = {}
for case in rs.TRAIN_VAE_CASES:
    with tempfile.TemporaryDirectory() as t:
        out[case] = rs.trace_train_vae("reference", case, t)
json.dump(out, open(os.path.join(HERE, "train_vae_calls.json"), "w"), indent=1, sort_keys=True)
with tempfile.TemporaryDirectory() as t:
    tr = rs.trace_train_py(t)
json.dump({"params": {k: v for k, v in rs.TRAIN_PARAMS.items()}, "episode_steps": rs.EPISODE_STEPS, "calls": tr}, open(os.path.join(HERE, "train_py_calls.json"), "w"), indent=1, sort_keys=True)
print("train_vae:", {k: len(v) for k, v in out.items()}, " train.py:", len(tr), "calls")
