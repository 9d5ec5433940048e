"""Checkpoint I/O with the reference's directory layout and manifest (tf.train.Saver defaults):

    <model_dir>/checkpoints/checkpoint                 text manifest, same keys TF writes
    <model_dir>/checkpoints/model.ckpt-<global_step>.npz                      one file per save (default format), and / or
    <model_dir>/checkpoints/model.ckpt-<global_step>.{index,data-00000-of-00001}   the reference's own TensorFlow bundle files

Either form holds EVERY global variable under its TF name (weights, `<var>/Adam`, `<var>/Adam_1`, beta powers, counters) —
reference vae/models.py:154,172-186 and ppo.py:184,202-216.  max_to_keep = 5 like tf.train.Saver().  `load` reads whichever
exists, so a checkpoint directory written by the reference (TensorFlow bundles, mi355/tf_bundle.py) restores directly;
MI355_CKPT_FORMAT = npz (default) | tf | both selects what `save` writes.
"""
import os
import re

import numpy as np

MAX_TO_KEEP = 5


def _manifest(ckpt_dir):
    return os.path.join(ckpt_dir, "checkpoint")


def read_manifest(ckpt_dir):
    path = _manifest(ckpt_dir)
    if not os.path.exists(path):
        return None, []
    latest, allp = None, []
    for line in open(path):
        m = re.match(r'\s*(model_checkpoint_path|all_model_checkpoint_paths):\s*"(.*)"', line)
        if m:
            if m.group(1) == "model_checkpoint_path":
                latest = m.group(2)
            else:
                allp.append(m.group(2))
    return latest, allp


def save(ckpt_dir, global_step, variables, fmt=None):
    fmt = (fmt or os.environ.get("MI355_CKPT_FORMAT", "npz")).lower()
    if fmt not in ("npz", "tf", "both"):
        raise ValueError("checkpoint format %r (npz | tf | both)" % fmt)
    os.makedirs(ckpt_dir, exist_ok=True)
    name = "model.ckpt-%d" % int(global_step)
    if fmt in ("npz", "both"):
        tmp = os.path.join(ckpt_dir, name + ".tmp.npz")
        np.savez(tmp, **{k.replace("/", "|"): np.asarray(v) for k, v in variables.items()})
        os.replace(tmp, os.path.join(ckpt_dir, name + ".npz"))
    if fmt in ("tf", "both"):
        from . import tf_bundle
        tf_bundle.write_bundle(os.path.join(ckpt_dir, name), variables)
    _, allp = read_manifest(ckpt_dir)
    allp = [p for p in allp if p != name] + [name]
    while len(allp) > MAX_TO_KEEP:
        old = allp.pop(0)
        for ext in (".npz", ".index", ".data-00000-of-00001", ".meta"):
            try:
                os.remove(os.path.join(ckpt_dir, os.path.basename(old) + ext))
            except OSError:
                pass
    with open(_manifest(ckpt_dir), "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % name)
        for p in allp:
            f.write('all_model_checkpoint_paths: "%s"\n' % p)
    return os.path.join(ckpt_dir, "model.ckpt")


def latest(ckpt_dir):
    """tf.train.latest_checkpoint: path prefix of the newest checkpoint or None."""
    name, _ = read_manifest(ckpt_dir)
    if not name:
        return None
    return os.path.join(ckpt_dir, os.path.basename(name))


def load(prefix):
    path = prefix + ".npz"
    if os.path.exists(path):
        with np.load(path) as z:
            return {k.replace("|", "/"): z[k] for k in z.files}
    if os.path.exists(prefix + ".index"):                 # a checkpoint written by the reference (or by save(fmt="tf"))
        from . import tf_bundle
        return tf_bundle.read_bundle(prefix)
    raise FileNotFoundError("%s(.npz | .index): no such checkpoint" % prefix)
