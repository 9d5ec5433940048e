#!/usr/bin/env python3
"""Writes tests/golden/vae_common_calls.json: what the reference's own vae_common.py (load_vae + encode_state) does to the `vae.models`
classes, recorded by running /root/reference/vae_common.py against a recording stand-in (tests/ref_call_chain.py).  Run in the container
that holds the reference checkout; the fixture travels to the GPU box."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ref_call_chain as rc  # noqa: E402

out = {d: rc.trace_reference(d) for d in rc.MODEL_DIRS}
json.dump(out, open(os.path.join(HERE, "vae_common_calls.json"), "w"), indent=1, sort_keys=True)
print("wrote", len(out), "traces")
