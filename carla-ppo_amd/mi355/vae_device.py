"""Device side of the ConvVAE: owns the torch tensors (plumbing: HBM allocations + streams) that the native
engine (csrc/vae_engine.hip) works on, and converts between TensorFlow-named variables and the flat device layout.

No arithmetic happens here — every FLOP is in libmi355_carla.so.  There is no CPU fallback: constructing a
VaeDevice without a visible GPU or without the built library raises.
"""
import ctypes
from collections import OrderedDict

import numpy as np
import torch

from . import lib as milib
from .init import vae_variables

LOSS_KINDS = {"bce": 0, "bce_v2": 1, "mse": 2}
PRECISIONS = {"fp32": milib.MI_F32, "f32": milib.MI_F32, "bf16": milib.MI_BF16, "bf16x3": milib.MI_BF16X3}

# device tensor order inside the flat buffer (see mi_vae_param_layout)
_DEVICE_ORDER = (["vae/encoder/conv%d/%s" % (i, k) for i in (1, 2, 3, 4) for k in ("kernel", "bias")] +
                 ["@heads/kernel", "@heads/bias", "vae/decoder/dense1/kernel", "vae/decoder/dense1/bias"] +
                 ["vae/decoder/deconv%d/%s" % (i, k) for i in (1, 2, 3, 4) for k in ("kernel", "bias")])


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("mi355: no GPU visible — the MI355X HIP path has no CPU fallback "
                           "(the CPU oracle under oracle/ is test infrastructure only)")


class VaeDevice:
    def __init__(self, source_shape, target_shape, z_dim, beta, kl_tolerance, loss_fn, precision, max_batch=128,
                 with_optimizer=True, device=None):
        require_gpu()
        self.L = milib.get()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.source_shape = tuple(int(s) for s in source_shape)
        self.target_shape = tuple(int(s) for s in target_shape)
        if self.source_shape[:2] != self.target_shape[:2]:
            raise ValueError("ConvVAE: source and target must share height/width")
        self.z_dim = int(z_dim)
        self.precision = precision
        self.dtype = PRECISIONS[precision]
        self.loss_kind = LOSS_KINDS[loss_fn]
        self.beta, self.kl_tolerance = float(beta), float(kl_tolerance)
        self.with_optimizer = with_optimizer
        self.variables = vae_variables(self.z_dim, self.source_shape, self.target_shape)
        self.P = int(np.prod(self.target_shape))
        self.handle = None
        self.max_batch = 0
        self._alloc_params()
        self._create(max_batch)

    # ---- buffers ----
    def _desc(self, max_batch):
        return milib.MiVaeDesc(self.dtype, int(max_batch), self.source_shape[0], self.source_shape[1], self.source_shape[2],
                               self.target_shape[2], self.z_dim, self.loss_kind, self.beta, self.kl_tolerance, 0 if self.with_optimizer else 1)

    def _alloc_params(self):
        d = self._desc(1)
        n = self.L.mi_vae_param_floats(ctypes.byref(d))
        if n <= 0:
            raise milib.MiError("mi_vae_param_floats: " + self.L.cdll.mi_last_error().decode())
        self.n_flat = int(n)
        cnt = self.L.mi_vae_tensor_count()
        off, size = np.zeros(cnt, np.int64), np.zeros(cnt, np.int64)
        self.L.mi_vae_param_layout(ctypes.byref(d), off.ctypes.data, size.ctypes.data, cnt)
        self.layout = OrderedDict((name, (int(o), int(s))) for name, o, s in zip(_DEVICE_ORDER, off, size))
        z = lambda dt=torch.float32: torch.zeros(self.n_flat, device=self.device, dtype=dt)   # noqa: E731
        self.params = z()
        self.grads = z() if self.with_optimizer else None
        self.adam_m = z() if self.with_optimizer else None
        self.adam_v = z() if self.with_optimizer else None
        # the weight copies the MFMA kernels read, in the engine's storage type: bf16, or split storage (bf16x3: hi | lo halves in one 32-bit word)
        store = {milib.MI_F32: torch.float32, milib.MI_BF16: torch.bfloat16, milib.MI_BF16X3: torch.int32}[self.dtype]
        self.shadow = z(store) if self.dtype != milib.MI_F32 else None
        self.weights_t = z(store)                                                                # K-contiguous kernel copies
        self.metrics = torch.zeros(3, device=self.device)
        self.decoder_offset = self.layout["vae/decoder/dense1/kernel"][0]   # grads[decoder_offset:] are ready first in backward
        # data-parallel gradient buckets in the order backward completes them: (engine part, first float, one past the last float).
        # decoder (43 % of the parameters) | heads + conv4 (51 %) | conv3..conv1 (6 %): only the last, small bucket is reduced with
        # nothing left to overlap it.
        bk = np.zeros(9, np.int64)
        self.L.mi_vae_dp_buckets(ctypes.byref(d), bk.ctypes.data)          # the library's own table: mi_vae_train_step_dp walks the same one
        self.grad_buckets = [tuple(int(x) for x in bk[3 * i:3 * i + 3]) for i in range(3)]
        c4 = self.layout["vae/encoder/conv4/kernel"][0]
        assert self.grad_buckets == [(1, self.decoder_offset, self.n_flat), (3, c4, self.decoder_offset), (4, 0, c4)]

    def _create(self, max_batch):
        recreated = self.handle is not None
        if self.handle is not None:
            self.L.mi_vae_destroy(self.handle)
            self.handle = None
        d = self._desc(max_batch)
        nbytes = self.L.mi_vae_workspace_bytes(ctypes.byref(d))
        self.workspace = torch.empty(int(nbytes), device=self.device, dtype=torch.uint8)
        p = milib.ptr
        self.handle = self.L.mi_vae_create(ctypes.byref(d), p(self.params), p(self.grads), p(self.adam_m), p(self.adam_v),
                                           p(self.shadow), p(self.weights_t), p(self.workspace), int(nbytes))
        if not self.handle:
            raise milib.MiError("mi_vae_create: " + self.L.cdll.mi_last_error().decode())
        self.max_batch = int(max_batch)
        self.losses = self._view(0, 2)
        if getattr(self, "_seed", None) is not None:        # a re-created engine (larger workspace) moves on to a fresh noise stream
            self._seed += 0x9E3779B97F4A7C15
            self.L.mi_vae_set_seed(self.handle, int(self._seed) & 0xFFFFFFFFFFFFFFFF)
        if recreated:                                       # the derived weight copies that live INSIDE the workspace (the fragment-ordered kernels of the activation-resident
            self.sync_shadow()                              # layers) went with the old one: without this the first step of the new engine runs the general kernels

    def ensure_batch(self, b):
        if b > self.max_batch:
            torch.cuda.synchronize(self.device)
            self._create(b)

    def _view(self, which, n, dtype=torch.float32):
        """Zero-copy torch view of an engine buffer inside the workspace."""
        addr = self.L.mi_vae_buffer(self.handle, which)
        base = self.workspace.data_ptr()
        esz = torch.empty(0, dtype=dtype).element_size()
        off = addr - base
        return self.workspace[off:off + n * esz].view(dtype)

    def close(self):
        if self.handle is not None:
            self.L.mi_vae_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- TF-named variables <-> flat device layout ----
    def _to_flat(self, named):
        flat = np.zeros(self.n_flat, np.float32)
        for name, (o, s) in self.layout.items():
            if name == "@heads/kernel":
                a = np.concatenate([named["vae/mean/kernel"], named["vae/logstd_sqare/kernel"]], axis=1)
            elif name == "@heads/bias":
                a = np.concatenate([named["vae/mean/bias"], named["vae/logstd_sqare/bias"]])
            else:
                a = named[name]
            a = np.asarray(a, np.float32)
            expect = (self.variables[name] if not name.startswith("@") else None)
            if expect is not None and tuple(a.shape) != tuple(expect):
                raise ValueError("%s: shape %s, expected %s" % (name, a.shape, expect))
            if a.size != s:
                raise ValueError("%s: %d elements, expected %d" % (name, a.size, s))
            flat[o:o + s] = a.reshape(-1)
        return flat

    def _from_flat(self, flat):
        out = OrderedDict()
        z = self.z_dim
        for name, shape in self.variables.items():
            if name in ("vae/mean/kernel", "vae/logstd_sqare/kernel"):
                o, s = self.layout["@heads/kernel"]
                k = flat[o:o + s].reshape(-1, 2 * z)
                out[name] = np.ascontiguousarray(k[:, :z] if "mean" in name else k[:, z:])
            elif name in ("vae/mean/bias", "vae/logstd_sqare/bias"):
                o, s = self.layout["@heads/bias"]
                out[name] = flat[o:o + z].copy() if "mean" in name else flat[o + z:o + 2 * z].copy()
            else:
                o, s = self.layout[name]
                out[name] = flat[o:o + s].reshape(shape).copy()
        return out

    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def sync_shadow(self):
        """Refresh the derived weight copies (bf16 shadow, K-contiguous kernels) after self.params was written from outside."""
        self.L.mi_vae_sync_shadow(self.handle, self.stream())

    def load_params(self, named):
        self.params.copy_(torch.from_numpy(self._to_flat(named)))
        self.sync_shadow()

    def load_slots(self, m_named, v_named):
        self.adam_m.copy_(torch.from_numpy(self._to_flat(m_named)))
        self.adam_v.copy_(torch.from_numpy(self._to_flat(v_named)))

    def export_params(self):
        return self._from_flat(self.params.cpu().numpy())

    def export_slots(self):
        return self._from_flat(self.adam_m.cpu().numpy()), self._from_flat(self.adam_v.cpu().numpy())

    def export_grads(self):
        return self._from_flat(self.grads.cpu().numpy())

    def check_guards(self, guard_index=-1):
        """Debug mode (MI355_DEBUG_GUARDS=1 when the engine was created): (guarded regions, overwritten guards, byte offset of guard `guard_index` or -1).
        0 regions = the mode is off.  Synchronises the device."""
        n, bad, off = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_longlong(-1)
        self.L.mi_vae_debug_check_guards(self.handle, ctypes.addressof(n), ctypes.addressof(bad), int(guard_index), ctypes.addressof(off))
        return n.value, bad.value, off.value

    # ---- steps (all asynchronous on the current torch stream) ----
    # Frame tables are float32 [N, feat] in [0, 1] or -- bf16 engine only -- raw uint8 camera frames [N, feat] (normalised to k / 255 inside
    # the kernels that read them: 4x less HBM traffic in conv1 forward / filter gradient and the loss); source and target share the format.
    @property
    def accepts_u8(self):
        return self.dtype == milib.MI_BF16

    @staticmethod
    def _u8(src, tgt=None):
        u8 = src.dtype == torch.uint8
        if tgt is not None and (tgt.dtype == torch.uint8) != u8:
            raise ValueError("source and target frame tables must share their format (both uint8 or both float32)")
        return 1 if u8 else 0

    def set_seed(self, seed):
        """Seed of the engine's own N(0,1) stream (used whenever a sampling pass gets eps=None)."""
        torch.cuda.synchronize(self.device)
        self.L.mi_vae_set_seed(self.handle, int(seed) & 0xFFFFFFFFFFFFFFFF)
        self._seed = int(seed)

    def forward(self, src, tgt, idx, B, inv_batch, eps, sample, want_grad, accumulate_metrics=True):
        self.ensure_batch(B)
        p = milib.ptr
        self.L.mi_vae_forward(self.handle, self.stream(), p(src), p(tgt), self._u8(src, tgt), p(idx), int(B), float(inv_batch), p(eps), int(sample),
                              int(want_grad), p(self.metrics) if accumulate_metrics else None, float(B * inv_batch))

    def backward(self, src, idx, eps, inv_batch, part=0):
        p = milib.ptr
        self.L.mi_vae_backward(self.handle, self.stream(), p(src), p(idx), p(eps), float(inv_batch), int(part))

    def apply_adam(self, alpha, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.L.mi_vae_apply_adam(self.handle, self.stream(), float(alpha), float(beta1), float(beta2), float(epsilon))

    def train_step(self, src, tgt, idx, B, inv_batch, eps, alpha, beta1=0.9, beta2=0.999, epsilon=1e-8, accumulate_metrics=True):
        """One whole SGD step in one C call (single-rank path)."""
        self.ensure_batch(B)
        p = milib.ptr
        self.L.mi_vae_train_step(self.handle, self.stream(), p(src), p(tgt), self._u8(src, tgt), p(idx), int(B), float(inv_batch), p(eps), float(alpha),
                                 float(beta1), float(beta2), float(epsilon), p(self.metrics) if accumulate_metrics else None, float(B * inv_batch))

    def train_step_dp(self, comm_handle, src, tgt, idx, B, inv_batch, eps, alpha, beta1=0.9, beta2=0.999, epsilon=1e-8, accumulate_metrics=True):
        """One whole DATA-PARALLEL SGD step in one C call: forward, backward in bucket order with each bucket's all-reduce queued on the library communicator's
        stream under the next part, join, Adam (mi_vae_train_step_dp).  comm_handle: the C-ABI communicator (mi355/dist.py)."""
        self.ensure_batch(B)
        p = milib.ptr
        self.L.mi_vae_train_step_dp(self.handle, comm_handle, self.stream(), p(src), p(tgt), self._u8(src, tgt), p(idx), int(B), float(inv_batch), p(eps), float(alpha),
                                    float(beta1), float(beta2), float(epsilon), p(self.metrics) if accumulate_metrics else None, float(B * inv_batch))

    def encode(self, src, idx, B, out):
        self.ensure_batch(B)
        self.L.mi_vae_encode(self.handle, self.stream(), milib.ptr(src), self._u8(src), milib.ptr(idx), int(B), milib.ptr(out))

    def decode(self, z, B, out):
        self.ensure_batch(B)
        self.L.mi_vae_decode(self.handle, self.stream(), milib.ptr(z), int(B), milib.ptr(out))

    def reconstruct(self, src, idx, B, eps, sample, out):
        self.ensure_batch(B)
        self.L.mi_vae_reconstruct(self.handle, self.stream(), milib.ptr(src), self._u8(src), milib.ptr(idx), int(B), milib.ptr(eps), int(sample), milib.ptr(out))

    def range_ok(self, t):
        flag = torch.zeros(1, device=self.device, dtype=torch.int32)
        self.L.mi_range_check(self.stream(), t.data_ptr(), t.numel(), 0.0, 1.0, flag.data_ptr())
        return int(flag.item()) == 0
