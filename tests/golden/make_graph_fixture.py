#!/usr/bin/env python3
"""Dump the reference's OWN serialized TensorFlow graphs into small fixtures (authoring container only: needs /root/reference).

    python tests/golden/make_graph_fixture.py [/root/reference]

The reference ships, next to every checkpoint, the `.meta` file tf.train.Saver wrote: a MetaGraphDef whose GraphDef holds every node
of the graph the reference trained with (forward pass, losses, the `gradients/` sub-graph tf.gradients built, the ApplyAdam nodes and
their hyper-parameter constants).  TensorFlow 1.13 cannot run here, but a GraphDef is plain protobuf, so it is decoded by hand (wire
format only, no schema files) into

    tests/golden/ref_graph_vae_rgb.json.gz    vae/models/rgb_bce_cnn_zdim64_beta1_kl_tolerance0.0_data  (ConvVAE, rgb target)
    tests/golden/ref_graph_vae_seg.json.gz    vae/models/seg_bce_cnn_zdim64_beta1_kl_tolerance0.0_data  (ConvVAE, seg target)
    tests/golden/ref_graph_ppo.json.gz        models/pretrained_agent                                   (PPO agent)

= {"source": ..., "nodes": [{"name", "op", "input": [...], "attr": {...decoded AttrValues...}}, ...]} minus the Saver / summary /
initializer plumbing.  oracle/tf_graph.py EXECUTES these node lists (tests/test_ref_graph.py), which pins the CPU oracle to the
reference's graph itself rather than to a reading of its Python source.

The protobuf decoding lives in oracle/tf_meta.py.
"""
import gzip
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.tf_meta import graph_nodes, prune  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def latest_meta(ckpt_dir):
    metas = [f for f in os.listdir(ckpt_dir) if f.endswith(".meta")]
    return os.path.join(ckpt_dir, max(metas, key=lambda f: int(f.split("-")[1].split(".")[0])))


def main():
    jobs = [("ref_graph_vae_rgb.json.gz", "vae/models/rgb_bce_cnn_zdim64_beta1_kl_tolerance0.0_data/checkpoints"),
            ("ref_graph_vae_seg.json.gz", "vae/models/seg_bce_cnn_zdim64_beta1_kl_tolerance0.0_data/checkpoints"),
            ("ref_graph_ppo.json.gz", "models/pretrained_agent/checkpoints")]
    for out, rel in jobs:
        meta = latest_meta(os.path.join(REF, rel))
        nodes = prune(graph_nodes(meta))
        doc = {"source": os.path.relpath(meta, REF), "generator": "tests/golden/make_graph_fixture.py", "nodes": nodes}
        raw = json.dumps(doc, separators=(",", ":"), sort_keys=True).encode()
        with open(os.path.join(OUT, out), "wb") as f:             # mtime=0: the fixture bytes depend on the graph only
            with gzip.GzipFile(fileobj=f, mode="wb", mtime=0) as g:
                g.write(raw)
        print("%s: %d nodes, %d B json, %d B gz (from %s)" % (out, len(nodes), len(raw), os.path.getsize(os.path.join(OUT, out)), doc["source"]))


if __name__ == "__main__":
    main()
