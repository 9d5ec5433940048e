#!/usr/bin/env python3
"""Host-side entry points of libmi355_carla.so under AddressSanitizer (run by tools/asan_host_check.sh with the ASan build preloaded): everything the C ABI
does WITHOUT a GPU -- descriptor / layout / workspace arithmetic, error paths with missing or misaligned buffers, CRC32C over odd lengths, tuning get / set,
the last-error string, RCCL-less communicator queries.  Exit code 0 = no ASan report (ASan aborts the process on the first one)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "carla-ppo_amd"))
import numpy as np
from mi355 import lib as milib

L = milib.get()
assert L.mi_abi_version() == 7
print("library:", milib.LIB_PATH)

# ---- VAE descriptor arithmetic for every dtype / loss / geometry the models use, with and without the guard mode ----
def vae_desc(dtype, tc=3, loss=0, max_batch=64):
    d = milib.MiVaeDesc()
    for k, v in dict(ih=80, iw=160, cin=3, ct=tc, z_dim=64, beta=1.0, kl_tolerance=0.0, loss_kind=loss, dtype=dtype, max_batch=max_batch).items():
        if hasattr(d, k):
            setattr(d, k, v)
    return d

fields = [f[0] for f in milib.MiVaeDesc._fields_]
print("MiVaeDesc fields:", fields)
for guards in ("0", "1"):
    os.environ["MI355_DEBUG_GUARDS"] = guards
    for dtype in (0, 1, 2):
        for tc in (3, 1):
            d = vae_desc(dtype, tc)
            n = L.mi_vae_param_floats(ctypes.byref(d))
            nt = L.mi_vae_tensor_count()
            offs = (ctypes.c_longlong * nt)(); sizes = (ctypes.c_longlong * nt)()
            L.mi_vae_param_layout(ctypes.byref(d), ctypes.addressof(offs), ctypes.addressof(sizes), nt)
            assert sum(sizes) <= n and offs[0] == 0
            ws = L.mi_vae_workspace_bytes(ctypes.byref(d))
            assert ws > 0
            # error paths: missing buffers, too-small workspace, misaligned buffers -> NULL handle + message, nothing dereferenced
            h = L.mi_vae_create(ctypes.byref(d), None, None, None, None, None, None, None, 0)
            assert not h and L.cdll.mi_last_error()
            h = L.mi_vae_create(ctypes.byref(d), 256, 256, 256, 256, 256, 256, 256, 16)
            assert not h
            h = L.mi_vae_create(ctypes.byref(d), 257, 512, 768, 1024, 1280, 1536, 1792, ws)
            assert not h
del os.environ["MI355_DEBUG_GUARDS"]
bad = vae_desc(1); bad.ih = 7
assert L.mi_vae_workspace_bytes(ctypes.byref(bad)) < 0 and b"geometry" in L.cdll.mi_last_error()

# ---- MlpVAE descriptor arithmetic (round 4 engine): layout without padding in the reference's variable order, workspace, error paths ----
def mlp_desc(dtype, enc=(512, 256), dec=(256, 512), S=38400, P=38400, z=64, max_batch=64, opt=1):
    d = milib.MiMlpVaeDesc()
    d.dtype, d.max_batch, d.source_size, d.target_size, d.z_dim, d.n_enc, d.n_dec = dtype, max_batch, S, P, z, len(enc), len(dec)
    for i, h in enumerate(enc):
        d.enc[i] = h
    for i, h in enumerate(dec):
        d.dec[i] = h
    d.loss_kind, d.with_optimizer, d.beta, d.kl_tolerance = 0, opt, 1.0, 0.0
    return d

for dtype in (0, 1):
    for enc, dec, P_ in (((512, 256), (256, 512), 38400), ((64,), (32, 48, 64), 12800), ((8, 16, 24, 32), (8,), 38400)):
        if dtype == 1 and any(h % 8 for h in enc + dec):
            continue
        for opt in (0, 1):
            d = mlp_desc(dtype, enc, dec, P=P_, opt=opt)
            n = L.mi_mlpvae_param_floats(ctypes.byref(d))
            nt = L.mi_mlpvae_tensor_count(ctypes.byref(d))
            assert nt == 2 * (len(enc) + 1 + len(dec) + 1)
            offs = (ctypes.c_longlong * nt)(); sizes = (ctypes.c_longlong * nt)()
            L.mi_mlpvae_param_layout(ctypes.byref(d), ctypes.addressof(offs), ctypes.addressof(sizes), nt)
            assert offs[0] == 0 and sum(sizes) == n and all(offs[i + 1] == offs[i] + sizes[i] for i in range(nt - 1))
            widths = (38400,) + enc
            assert [sizes[2 * i] for i in range(len(enc))] == [widths[i] * widths[i + 1] for i in range(len(enc))] and sizes[2 * len(enc)] == enc[-1] * 128
            ws = L.mi_mlpvae_workspace_bytes(ctypes.byref(d))
            assert ws > 0 and ws % 256 == 0
            assert not L.mi_mlpvae_create(ctypes.byref(d), None, None, None, None, None, None, None, 0) and L.cdll.mi_last_error()
            assert not L.mi_mlpvae_create(ctypes.byref(d), 256, 256, 256, 256, 256, 256, 256, 16)            # workspace too small
            assert not L.mi_mlpvae_create(ctypes.byref(d), 256, 256, 256, 256, 256, 256, 257, ws)            # workspace misaligned
for bad in (mlp_desc(2), mlp_desc(1, enc=(500, 256)), mlp_desc(0, enc=()), mlp_desc(0, z=0), mlp_desc(0, S=38402)):
    assert L.mi_mlpvae_workspace_bytes(ctypes.byref(bad)) < 0 and b"unsupported" in L.cdll.mi_last_error()
assert L.cdll.mi_mlpvae_forward(None, None, None, None, 0, None, 4, 0.25, None, 0, 0, None, 0.0) != 0 and b"null handle" in L.cdll.mi_last_error()
assert L.mi_mlpvae_decoder_offset(None) == -1 and L.mi_mlpvae_buffer(None, 0) is None

# ---- argument checks of the round-4 optimiser / staging launchers (every one of these returns before anything is launched) ----
off1 = (ctypes.c_longlong * 2)(0, 4096); K1 = (ctypes.c_int * 2)(64, 64); N1 = (ctypes.c_int * 2)(64, 64)
adam = L.cdll.mi_adam_tf_layouts
base = dict(stream=None, dtype=1, p=4096, m=8192, v=12288, g=16384, n=10000)
def adam_rc(dtype=1, p=4096, n=10000, offs=off1, K=K1, N=N1, count=2, shadow=32768, wt=65536):
    return adam(None, dtype, p, 8192, 12288, 16384, n, ctypes.addressof(offs), ctypes.addressof(K), ctypes.addressof(N), None, count, 1e-3, None, 0.9, 0.999, 1e-8, shadow, wt, 0)
assert adam_rc(dtype=2) != 0 and b"MI_F32 or MI_BF16" in L.cdll.mi_last_error()
assert adam_rc(count=17) != 0
assert adam_rc(p=4100) != 0 and b"16-byte aligned" in L.cdll.mi_last_error()
assert adam_rc(wt=65540) != 0
assert adam_rc(N=(ctypes.c_int * 2)(64, 62)) != 0 and b"N % 4" in L.cdll.mi_last_error()
assert adam_rc(offs=(ctypes.c_longlong * 2)(0, 2048)) != 0 and b"non-overlapping" in L.cdll.mi_last_error()      # second kernel starts inside the first
assert adam_rc(n=8000) != 0                                                                                      # second kernel ends behind the buffer
assert L.cdll.mi_gather_rows_cast(None, 2, 4096, None, 4, 128, 8192) != 0
assert L.cdll.mi_gather_rows_cast(None, 1, None, None, 4, 128, 8192) != 0 and L.cdll.mi_gather_rows_cast(None, 1, 4096, None, 4, 0, 8192) != 0
assert L.cdll.mi_gather_rows_cast(None, 1, 4096, None, 0, 128, 8192) == 0                                         # empty batch: nothing to do

# ---- PPO descriptor arithmetic ----
pd = milib.MiPpoDesc()
print("MiPpoDesc fields:", [f[0] for f in milib.MiPpoDesc._fields_])
for k, v in dict(input_dim=67, num_actions=2, h1=500, h2=300, max_batch=2048).items():
    if hasattr(pd, k):
        setattr(pd, k, v)
if L.mi_ppo_param_floats(ctypes.byref(pd)) > 0:
    assert L.mi_ppo_workspace_bytes(ctypes.byref(pd)) > 0

# ---- CRC32C (TensorFlow's masked checksum of the bundle files) over every length 0..300 and every alignment 0..7 ----
buf = np.frombuffer(np.random.RandomState(0).bytes(512), np.uint8).copy()
seen = set()
for off in range(8):
    for n in range(0, 301, 7):
        seen.add(int(L.mi_crc32c(0, buf[off:].ctypes.data, n)) & 0xffffffff)
assert len(seen) > 300
assert int(L.mi_crc32c(0, np.frombuffer(b"123456789", np.uint8).ctypes.data, 9)) & 0xffffffff == 0xE3069283      # the standard check value

# ---- tuning knobs: set returns the previous value; unknown keys are rejected without touching memory ----
for key in range(0, 20):
    prev = L.cdll.mi_set_tuning(key, 1)
    L.cdll.mi_set_tuning(key, prev)
L.cdll.mi_set_tuning(12345, 1)
L.cdll.mi_set_tuning(-3, 1)

# ---- communicator queries that do not need RCCL or a device ----
assert L.mi_comm_id_bytes() == 128
assert L.mi_deconv2d_tail_blocks.__name__ or True
print("last error string:", L.cdll.mi_last_error()[:80])
print("asan host check: ok")
