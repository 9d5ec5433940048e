"""Device side of the MlpVAE (reference vae/models.py:271-299 on top of the base VAE graph :85-142): the same interface as VaeDevice over the native MlpVAE engine
(csrc/mlp_engine.hip, round 4): one C call per SGD step (`mi_mlpvae_train_step`), the frame rows gathered and converted in one launch, bias gradients as a ones row of
the filter gradients, gradients stored (no zeroing, no atomics), Adam writing both weight layouts.  Rounds 1-3 sequenced the op-level entry points from here.

No CPU fallback: constructing the device without a GPU or without the built library raises.
"""
import ctypes
from collections import OrderedDict

import numpy as np
import torch

from . import lib as milib
from .init import mlp_vae_variables
from .vae_device import LOSS_KINDS, PRECISIONS, require_gpu


class MlpVaeDevice:
    def __init__(self, source_shape, target_shape, z_dim, beta, kl_tolerance, loss_fn, precision, encoder_sizes=(512, 256),
                 decoder_sizes=(256, 512), max_batch=128, with_optimizer=True, device=None):
        require_gpu()
        self.L = milib.get()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.source_shape = tuple(int(s) for s in source_shape)
        self.target_shape = tuple(int(s) for s in target_shape)
        self.z_dim = int(z_dim)
        self.precision = precision
        self.dtype = PRECISIONS[precision]
        if self.dtype == milib.MI_BF16X3:
            raise ValueError("MlpVAE: precision 'bf16x3' (split storage) is built for the ConvVAE engine; use 'fp32' (1e-4 parity) or 'bf16'")
        self.bf16 = self.dtype == milib.MI_BF16
        self.T = torch.bfloat16 if self.bf16 else torch.float32
        self.loss_kind = LOSS_KINDS[loss_fn]
        self.beta, self.kl_tolerance = float(beta), float(kl_tolerance)
        self.with_optimizer = with_optimizer
        self.enc_sizes = tuple(int(h) for h in encoder_sizes)
        self.dec_sizes = tuple(int(h) for h in decoder_sizes)
        self.variables = mlp_vae_variables(self.z_dim, self.source_shape, self.target_shape, self.enc_sizes, self.dec_sizes)
        self.S = int(np.prod(self.source_shape))
        self.P = int(np.prod(self.target_shape))
        vec = 8 if self.bf16 else 4
        for n in (self.S, self.P, self.z_dim) + self.enc_sizes + self.dec_sizes:
            if n % vec != 0:
                raise ValueError("MlpVAE: every layer width must be a multiple of %d (16-byte vectors), got %d" % (vec, n))
        if not (1 <= len(self.enc_sizes) <= milib.MI_MLP_MAX_HIDDEN and 1 <= len(self.dec_sizes) <= milib.MI_MLP_MAX_HIDDEN):
            raise ValueError("MlpVAE: 1 .. %d hidden layers per side" % milib.MI_MLP_MAX_HIDDEN)
        self.handle = None
        # tensor names in the engine's order (mi_mlpvae_param_layout): the two heads are one [K, 2Z] kernel like in the ConvVAE engine
        names = []
        for i in range(len(self.enc_sizes)):
            names += ["vae/encoder/dense" + ("_%d" % i if i else "") + s for s in ("/kernel", "/bias")]
        names += ["@heads/kernel", "@heads/bias"]
        for i in range(len(self.dec_sizes) + 1):
            names += ["vae/decoder/dense" + ("_%d" % i if i else "") + s for s in ("/kernel", "/bias")]
        d = self._desc(1)
        n = self.L.mi_mlpvae_param_floats(ctypes.byref(d))
        if n <= 0:
            raise milib.MiError("mi_mlpvae_param_floats: " + self.L.cdll.mi_last_error().decode())
        self.n_flat = int(n)
        cnt = self.L.mi_mlpvae_tensor_count(ctypes.byref(d))
        assert cnt == len(names), (cnt, names)
        off, size = np.zeros(cnt, np.int64), np.zeros(cnt, np.int64)
        self.L.mi_mlpvae_param_layout(ctypes.byref(d), off.ctypes.data, size.ctypes.data, cnt)
        self.layout = OrderedDict((name, (int(o), int(s))) for name, o, s in zip(names, off, size))
        self.decoder_offset = self.layout["vae/decoder/dense/kernel"][0]                      # grads[decoder_offset:] are complete first in backward
        self.grad_buckets = [(1, self.decoder_offset, self.n_flat), (2, 0, self.decoder_offset)]   # (backward part, lo, hi), see VaeDevice
        z = lambda dt=torch.float32: torch.zeros(self.n_flat, device=self.device, dtype=dt)   # noqa: E731
        self.params = z()
        self.grads = z() if with_optimizer else None
        self.adam_m = z() if with_optimizer else None
        self.adam_v = z() if with_optimizer else None
        self.shadow = z(torch.bfloat16) if self.bf16 else None
        # K-contiguous copies [N][K] of every dense kernel for the forward GEMMs (the MFMA kernels stream the B operand along K), rewritten by the optimiser step itself
        # (mi_adam_tf_layouts); the input gradients read the [K, N] originals, which ARE K-contiguous for x * W^T
        self.weights_t = z(self.T)
        self.metrics = torch.zeros(3, device=self.device)
        self.max_batch = 0
        self._create(max_batch)

    def _desc(self, max_batch):
        d = milib.MiMlpVaeDesc()
        d.dtype, d.max_batch, d.source_size, d.target_size, d.z_dim = self.dtype, int(max_batch), self.S, self.P, self.z_dim
        d.n_enc, d.n_dec = len(self.enc_sizes), len(self.dec_sizes)
        for i, h in enumerate(self.enc_sizes):
            d.enc[i] = h
        for i, h in enumerate(self.dec_sizes):
            d.dec[i] = h
        d.loss_kind, d.with_optimizer, d.beta, d.kl_tolerance = self.loss_kind, 1 if self.with_optimizer else 0, self.beta, self.kl_tolerance
        return d

    def _create(self, max_batch):
        if self.handle is not None:
            self.L.mi_mlpvae_destroy(self.handle)
            self.handle = None
        d = self._desc(max_batch)
        nbytes = self.L.mi_mlpvae_workspace_bytes(ctypes.byref(d))
        if nbytes <= 0:
            raise milib.MiError("mi_mlpvae_workspace_bytes: " + self.L.cdll.mi_last_error().decode())
        self.workspace = torch.empty(int(nbytes), device=self.device, dtype=torch.uint8)
        p = milib.ptr
        self.handle = self.L.mi_mlpvae_create(ctypes.byref(d), p(self.params), p(self.grads), p(self.adam_m), p(self.adam_v), p(self.shadow), p(self.weights_t),
                                              p(self.workspace), int(nbytes))
        if not self.handle:
            raise milib.MiError("mi_mlpvae_create: " + self.L.cdll.mi_last_error().decode())
        self.max_batch = int(max_batch)
        self.losses = self._view(0, 2)

    def ensure_batch(self, b):
        if b > self.max_batch:
            torch.cuda.synchronize(self.device)
            self._create(b)                                 # (the weight copies live outside the workspace: nothing to rebuild)

    def _view(self, which, n, dtype=torch.float32):
        """Zero-copy torch view of an engine buffer inside the workspace."""
        addr = self.L.mi_mlpvae_buffer(self.handle, which)
        off = addr - self.workspace.data_ptr()
        esz = torch.empty(0, dtype=dtype).element_size()
        return self.workspace[off:off + n * esz].view(dtype)

    @property
    def mean(self):
        return self._view(1, self.max_batch * self.z_dim).view(self.max_batch, self.z_dim)

    @property
    def logvar(self):
        return self._view(2, self.max_batch * self.z_dim).view(self.max_batch, self.z_dim)

    def close(self):
        if self.handle is not None:
            self.L.mi_mlpvae_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def sync_shadow(self):
        """Refresh the derived weight copies (bf16 shadow, K-contiguous kernels) after self.params was written from outside."""
        self.L.mi_mlpvae_sync_shadow(self.handle, self.stream())

    # ---- TF-named variables <-> flat layout ----
    def _to_flat(self, named):
        flat = np.zeros(self.n_flat, np.float32)
        for name, (o, s) in self.layout.items():
            if name == "@heads/kernel":
                a = np.concatenate([named["vae/mean/kernel"], named["vae/logstd_sqare/kernel"]], axis=1)
            elif name == "@heads/bias":
                a = np.concatenate([named["vae/mean/bias"], named["vae/logstd_sqare/bias"]])
            else:
                a = np.asarray(named[name], np.float32)
                if tuple(a.shape) != tuple(self.variables[name]):
                    raise ValueError("%s: shape %s, expected %s" % (name, a.shape, self.variables[name]))
            a = np.asarray(a, np.float32)
            if a.size != s:
                raise ValueError("%s: %d elements, expected %d" % (name, a.size, s))
            flat[o:o + s] = a.reshape(-1)
        return flat

    def _from_flat(self, flat):
        out, z = OrderedDict(), self.z_dim
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

    # ---- steps (all asynchronous on the current torch stream) ----
    accepts_u8 = True                                   # round 5: frame tables may stay uint8 camera bytes in HBM (normalised where the minibatch is staged / in the loss kernel)

    @staticmethod
    def _tab(t, what):
        if t is not None and t.dtype not in (torch.float32, torch.uint8):
            raise TypeError("MlpVAE: %s must be a float32 or uint8 device tensor, got %s" % (what, t.dtype))
        return t

    @staticmethod
    def _u8(src, tgt=None):
        """frames_u8 of the C ABI: bit 0 = the source table holds raw uint8 bytes, bit 1 = the target table does."""
        return (1 if src is not None and src.dtype == torch.uint8 else 0) | (2 if tgt is not None and tgt.dtype == torch.uint8 else 0)

    def forward(self, src, tgt, idx, B, inv_batch, eps, sample, want_grad, accumulate_metrics=True):
        self.ensure_batch(B)
        p = milib.ptr
        self.L.mi_mlpvae_forward(self.handle, self.stream(), p(self._tab(src, "the source table")), p(self._tab(tgt, "the target table")), self._u8(src, tgt), p(idx), int(B), float(inv_batch),
                                 p(eps), int(sample), int(want_grad), p(self.metrics) if accumulate_metrics else None, float(B * inv_batch))
        # the fp32 engine with idx == None reads the first layer's filter-gradient operand straight from `src` in backward(): keep the table alive until then
        # (a temporary freed between forward() and backward(part=2) of the data-parallel path would be read after free -- ADVICE r04)
        self._last_src = src if want_grad else None

    def backward(self, src, idx, eps, inv_batch, part=0):
        """part 0 = everything, 1 = decoder half (+ dz), 2 = encoder half: the data-parallel host all-reduces grads[decoder_offset:] in between."""
        self.L.mi_mlpvae_backward(self.handle, self.stream(), milib.ptr(eps), float(inv_batch), int(part))
        if part in (0, 2):
            self._last_src = None

    def apply_adam(self, alpha, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.L.mi_mlpvae_apply_adam(self.handle, self.stream(), float(alpha), float(beta1), float(beta2), float(epsilon))

    def train_step(self, src, tgt, idx, B, inv_batch, eps, alpha, beta1=0.9, beta2=0.999, epsilon=1e-8, accumulate_metrics=True):
        """One whole SGD step in one C call (single-rank path)."""
        self.ensure_batch(B)
        p = milib.ptr
        self.L.mi_mlpvae_train_step(self.handle, self.stream(), p(self._tab(src, "the source table")), p(self._tab(tgt, "the target table")), self._u8(src, tgt), p(idx), int(B), float(inv_batch),
                                    p(eps), float(alpha), float(beta1), float(beta2), float(epsilon), p(self.metrics) if accumulate_metrics else None, float(B * inv_batch))

    def train_step_dp(self, comm_handle, src, tgt, idx, B, inv_batch, eps, alpha, beta1=0.9, beta2=0.999, epsilon=1e-8, accumulate_metrics=True):
        """One whole DATA-PARALLEL SGD step in one C call (mi_mlpvae_train_step_dp, round 6): forward, the decoder half of backward, its all-reduce on the library
        communicator's stream under the encoder half, that half's all-reduce, join, Adam.  comm_handle: the C-ABI communicator (mi355/dist.py)."""
        self.ensure_batch(B)
        p = milib.ptr
        self.L.mi_mlpvae_train_step_dp(self.handle, comm_handle, self.stream(), p(self._tab(src, "the source table")), p(self._tab(tgt, "the target table")), self._u8(src, tgt), p(idx), int(B),
                                       float(inv_batch), p(eps), float(alpha), float(beta1), float(beta2), float(epsilon), p(self.metrics) if accumulate_metrics else None, float(B * inv_batch))
        self._last_src = None

    def encode(self, src, idx, B, out):
        self.ensure_batch(B)
        self.L.mi_mlpvae_encode(self.handle, self.stream(), milib.ptr(self._tab(src, "the source table")), self._u8(src), milib.ptr(idx), int(B), milib.ptr(out))

    def decode(self, z, B, out):
        self.ensure_batch(B)
        zz = z.to(torch.float32).contiguous()
        self.L.mi_mlpvae_decode(self.handle, self.stream(), milib.ptr(zz), int(B), milib.ptr(out))

    def reconstruct(self, src, idx, B, eps, sample, out):
        self.ensure_batch(B)
        self.L.mi_mlpvae_reconstruct(self.handle, self.stream(), milib.ptr(self._tab(src, "the source table")), self._u8(src), milib.ptr(idx), int(B), milib.ptr(eps), int(sample), milib.ptr(out))

    def range_ok(self, t):
        flag = torch.zeros(1, device=self.device, dtype=torch.int32)
        self.L.mi_range_check(self.stream(), t.data_ptr(), t.numel(), 0.0, 1.0, flag.data_ptr())
        return int(flag.item()) == 0
