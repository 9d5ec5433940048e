"""Shared helpers of the model-level GPU parity tests (tests/test_*_gpu.py): seeded synthetic frames, the drop-in ConvVAE on the HIP path, deviation measures.

Tolerances (stated per north_star): fp32 mode -- losses / outputs / gradients within 1e-4 relative of the oracle;
bf16 mode -- compared with the oracle's bf16-storage emulation (same rounding points, fp32 accumulate): losses 2e-3,
gradients 3e-2 of each tensor's max (bf16 has 8 mantissa bits; deviations are reported, not hidden).
Index work (minibatch permutations) is the reference's own legacy-numpy shuffle, hence bit-exact by construction."""
import numpy as np

from oracle import vae_oracle as vo
from vae.models import ConvVAE


def synth_frames(n, seed=1234):
    return np.random.RandomState(seed).randint(0, 256, (n, 80, 160, 3), dtype=np.uint8).astype(np.float32) / 255.0


def make(tmp_path, precision, target_c=3, params=None, **kw):
    m = ConvVAE(np.array([80, 160, 3]), np.array([80, 160, target_c]), z_dim=64, model_dir=str(tmp_path), precision=precision, **kw)
    if params is not None:
        m.set_weights(params)
    m.init_session(init_logging=False)
    return m


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def trained_like_params(seed=0, target_c=3):
    """Glorot weights + small random biases so that ReLU masks / biases carry signal on every path."""
    p = vo.init_vae_params(seed, 64, (80, 160, 3), (80, 160, target_c))
    rng = np.random.RandomState(seed + 1)
    for k in p:
        if k.endswith("bias"):
            p[k] = (0.05 * rng.standard_normal(p[k].shape)).astype(np.float32)
    return p


def _dev_table(title, rows):
    """Printed (pytest -s / captured on failure, and written next to the test run): the measured deviations are part of the parity statement."""
    print("\n" + title)
    for k, v in rows:
        print("  %-38s %s" % (k, v))


# ----------------------------------------------------------------------------------------------------------------------
# MlpVAE (reference vae/models.py:271-299) against the oracle's restatement
def _mlp_params(seed, src_shape, tgt_shape, enc, dec, scale=1.0):
    p = vo.init_mlp_vae_params(seed, z_dim=64, source_shape=src_shape, target_shape=tgt_shape, encoder_sizes=enc, decoder_sizes=dec)
    rng = np.random.RandomState(seed + 1)
    for k in p:
        if k.endswith("bias"):
            p[k] = (0.05 * rng.standard_normal(p[k].shape)).astype(np.float32)       # non-zero biases: the bias paths are exercised
        else:
            p[k] = (scale * p[k]).astype(np.float32)
    return p
