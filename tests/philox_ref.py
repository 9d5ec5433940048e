"""numpy restatement of the engine-side noise source (csrc/elementwise.hip: philox4x32_10 + Box-Muller), test infrastructure.
Philox4x32-10: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11); multipliers 0xD2511F53 / 0xCD9E8D57,
Weyl constants 0x9E3779B9 / 0xBB67AE85.  The reference draws its noise from TensorFlow's unseeded stateful RNG (tfp Normal.sample,
vae/models.py:101-105, ppo.py:58-60): there is no stream to match, only the distribution -- this file pins OUR stream so that it stays
reproducible across launch geometries, ranks and graph replays."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: uint32 [..., 4], key: uint32 [..., 2] -> uint32 [..., 4]."""
    c = [ctr[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = key[..., 0].astype(np.uint64), key[..., 1].astype(np.uint64)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ k0) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & MASK
        n3 = p0 & MASK
        c = [n0, n1, n2, n3]
        k0, k1 = (k0 + np.uint64(W0)) & MASK, (k1 + np.uint64(W1)) & MASK
    return np.stack(c, axis=-1).astype(np.uint32)


def normal(seed, offset, n):
    """Element i = N(0,1) draw (offset + i) of stream `seed`, as philox_normal() on the device computes it (float32 Box-Muller)."""
    idx = np.uint64(offset) + np.arange(n, dtype=np.uint64)
    ctr = np.zeros((n, 4), np.uint32)
    ctr[:, 0] = (idx & MASK).astype(np.uint32)
    ctr[:, 1] = (idx >> np.uint64(32)).astype(np.uint32)
    key = np.zeros((n, 2), np.uint32)
    key[:, 0] = np.uint32(seed & 0xFFFFFFFF)
    key[:, 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    r = philox4x32_10(ctr, key)
    u1 = ((r[:, 0] >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(2.0 ** -24)
    u2 = (r[:, 1] >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return (np.sqrt(np.float32(-2.0) * np.log(u1), dtype=np.float32) * np.cos(np.float32(2.0 * np.pi) * u2.astype(np.float64)).astype(np.float32)).astype(np.float32)
