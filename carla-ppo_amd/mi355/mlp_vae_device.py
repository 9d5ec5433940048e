"""Device side of the MlpVAE (reference vae/models.py:271-299 on top of the base VAE graph :85-142): the same interface as
VaeDevice, composed from the op-level entry points of the C ABI (dense forward / input gradient `mi_gemm_bias_act`, filter gradient
`mi_gemm_wgrad`, bias gradient `mi_colsum`, `mi_vae_reparam_kl_fwd/bwd`, `mi_bce_logits_fwd_bwd`, `mi_vae_finalize_losses`,
`mi_adam_tf_flat`).  The ConvVAE has a native engine (csrc/vae_engine.hip) because its 45 launches per step need stream
orchestration; the MLP is seven dense layers in a row, so the sequencing lives here and every FLOP stays in libmi355_carla.so.

No CPU fallback: constructing the device without a GPU or without the built library raises.
"""
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
        self.handle = None                                  # (no native engine object; kept for interface symmetry)
        # dense layers in forward order: (kernel name, K, N); the two heads share one [K, 2Z] GEMM like the ConvVAE engine
        self.enc = [("vae/encoder/dense" + ("_%d" % i if i else ""), k, n) for i, (k, n) in enumerate(zip((self.S,) + self.enc_sizes[:-1], self.enc_sizes))]
        dec_out = self.dec_sizes + (self.P,)
        self.dec = [("vae/decoder/dense" + ("_%d" % i if i else ""), k, n) for i, (k, n) in enumerate(zip((self.z_dim,) + dec_out[:-1], dec_out))]
        self.layout = OrderedDict()
        off = 0
        for name, k, n in self.enc:
            self.layout[name + "/kernel"], off = (off, k * n), off + k * n
            self.layout[name + "/bias"], off = (off, n), off + n
        kh = self.enc_sizes[-1]
        self.layout["@heads/kernel"], off = (off, kh * 2 * self.z_dim), off + kh * 2 * self.z_dim
        self.layout["@heads/bias"], off = (off, 2 * self.z_dim), off + 2 * self.z_dim
        self.decoder_offset = off                            # grads[decoder_offset:] are complete first in backward
        for name, k, n in self.dec:
            self.layout[name + "/kernel"], off = (off, k * n), off + k * n
            self.layout[name + "/bias"], off = (off, n), off + n
        self.n_flat = off
        self.grad_buckets = [(1, self.decoder_offset, self.n_flat), (2, 0, self.decoder_offset)]   # (backward part, lo, hi), see VaeDevice
        z = lambda dt=torch.float32: torch.zeros(self.n_flat, device=self.device, dtype=dt)   # noqa: E731
        self.params = z()
        self.grads = z() if with_optimizer else None
        self.adam_m = z() if with_optimizer else None
        self.adam_v = z() if with_optimizer else None
        self.shadow = z(torch.bfloat16) if self.bf16 else None
        # K-contiguous copies [N][K] of every dense kernel for the forward GEMMs (the MFMA kernels stream the B operand along K): refreshed after
        # every optimiser step by mi_transpose_weights; the input gradients read the [K, N] originals, which ARE K-contiguous for x * W^T
        self.weights_t = z(self.T)
        kern = [(self.layout[name + "/kernel"][0], k, n) for name, k, n in self.enc] + [(self.layout["@heads/kernel"][0], self.enc_sizes[-1], 2 * self.z_dim)] \
            + [(self.layout[name + "/kernel"][0], k, n) for name, k, n in self.dec]
        self._tr_off = np.array([o for o, _, _ in kern], np.int64)
        self._tr_k = np.array([k for _, k, _ in kern], np.int32)
        self._tr_n = np.array([n for _, _, n in kern], np.int32)
        self.slab = None                                    # split-K partial sums of the long-K layers (allocated on first use)
        self.metrics = torch.zeros(3, device=self.device)
        self.losses = torch.zeros(2, device=self.device)
        self.nchunks = int(self.L.mi_recon_loss_chunks(self.P))
        self.max_batch = 0
        self.last_B = 0
        self.ensure_batch(max_batch)

    # ---- buffers ----
    def ensure_batch(self, b):
        if b <= self.max_batch:
            return
        B, dev, T = int(b), self.device, self.T
        e = lambda *s, dt=T: torch.empty(*s, device=dev, dtype=dt)                            # noqa: E731
        self.x = e(B, self.S)
        self.h = [e(B, n) for _, _, n in self.enc]                                            # ReLU outputs of the encoder layers
        self.heads = e(B, 2 * self.z_dim, dt=torch.float32)
        self.mean, self.logvar, self.kl_row = (e(B, self.z_dim, dt=torch.float32), e(B, self.z_dim, dt=torch.float32), e(B, dt=torch.float32))
        self.z = e(B, self.z_dim)
        self.zf32 = e(B, self.z_dim, dt=torch.float32)
        self.d = [e(B, n) for _, _, n in self.dec]                                            # decoder activations; d[-1] = logits
        self.partial = e(B * self.nchunks, dt=torch.float32)
        if self.with_optimizer:
            self.g_d = [e(B, n) for _, _, n in self.dec]                                      # gradients of the decoder activations
            self.dz = e(B, self.z_dim, dt=torch.float32)
            self.dheads = e(B, 2 * self.z_dim)
            self.g_h = [e(B, n) for _, _, n in self.enc]
        self.max_batch = B

    def close(self):
        pass

    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _w(self, name):
        """Device pointer of a kernel in the dtype the GEMMs read (bf16 shadow copy or the fp32 master)."""
        o, _ = self.layout[name]
        return (self.shadow if self.bf16 else self.params)[o:].data_ptr()

    def _p(self, buf, name):
        o, _ = self.layout[name]
        return buf[o:].data_ptr()

    def sync_shadow(self):
        if self.bf16:
            self.L.mi_cast_f32_to_bf16(self.stream(), self.params.data_ptr(), self.shadow.data_ptr(), self.n_flat)
        self._refresh_transposed()

    def _refresh_transposed(self):
        self.L.mi_transpose_weights(self.stream(), self.dtype, self.params.data_ptr(), self.weights_t.data_ptr(), self._tr_off.ctypes.data, self._tr_k.ctypes.data,
                                    self._tr_n.ctypes.data, len(self._tr_off))

    def _wt(self, name):
        o, _ = self.layout[name]
        return self.weights_t[o:].data_ptr()

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

    # ---- building blocks ----
    SPLIT_K = 4096                                          # reductions at least this long are split over the chip (38400 -> 30 slabs) and finished by mi_splitk_finish

    def _dense(self, a, M, K, wname, N, bias, relu, out, out_f32=0, w_layout=0, mask=None):
        """out = mask(act(a W + bias)).  w_layout 0: forward, through the K-contiguous copy; 1: input gradient (x * W^T on the [K, N] original)."""
        M, K, N = int(M), int(K), int(N)
        w = self._wt(wname) if w_layout == 0 else self._w(wname)
        mk = mask.data_ptr() if mask is not None else None
        if K >= self.SPLIT_K and K % 128 == 0:
            ns = max(1, min(32, K // 1280))
            while K % (ns * 128) != 0 and ns > 1:
                ns -= 1
            if ns > 1:
                need = ns * M * N
                if self.slab is None or self.slab.numel() < need:
                    self.slab = torch.empty(need, device=self.device, dtype=torch.float32)
                self.L.mi_gemm_bias_act(self.stream(), self.dtype, a.data_ptr(), M, K, w, 1, N, None, 0, None, self.slab.data_ptr(), 1, ns)
                self.L.mi_splitk_finish(self.stream(), self.dtype, self.slab.data_ptr(), ns, M, N, bias, int(relu), mk, out.data_ptr(), int(out_f32))
                return
        self.L.mi_gemm_bias_act(self.stream(), self.dtype, a.data_ptr(), M, K, w, 1, N, bias, int(relu), mk, out.data_ptr(), int(out_f32), 1)

    def _stage_input(self, src, idx, B):
        rows = src if idx is None else src.index_select(0, idx.to(torch.int64))               # plumbing: row gather of the frame table
        rows = rows[:B].contiguous()
        if self.bf16:
            self.L.mi_cast_f32_to_bf16(self.stream(), rows.data_ptr(), self.x.data_ptr(), B * self.S)
            return self.x
        return rows

    def _encode(self, x, B):
        a = x
        for (name, k, n), h in zip(self.enc, self.h):
            self._dense(a, B, k, name + "/kernel", n, self._p(self.params, name + "/bias"), 1, h)
            a = h
        self._dense(a, B, self.enc_sizes[-1], "@heads/kernel", 2 * self.z_dim, None, 0, self.heads, out_f32=1)
        return a

    def _reparam(self, B, eps, sample):
        ob = self.layout["@heads/bias"][0]
        self.L.mi_vae_reparam_kl_fwd(self.stream(), self.dtype, self.heads.data_ptr(), 1, self.params[ob:].data_ptr(), self.params[ob + self.z_dim:].data_ptr(),
                                     milib.ptr(eps), int(sample), B, self.z_dim, self.mean.data_ptr(), self.logvar.data_ptr(), self.z.data_ptr(), self.kl_row.data_ptr())

    def _decode(self, z, B):
        a = z
        for i, ((name, k, n), d) in enumerate(zip(self.dec, self.d)):
            self._dense(a, B, k, name + "/kernel", n, self._p(self.params, name + "/bias"), 1 if i + 1 < len(self.dec) else 0, d)
            a = d
        return a

    # ---- steps (all asynchronous on the current torch stream) ----
    def forward(self, src, tgt, idx, B, inv_batch, eps, sample, want_grad, accumulate_metrics=True):
        self.ensure_batch(B)
        B = int(B)
        x = self._stage_input(src, idx, B)
        self._last_x = x
        self._encode(x, B)
        self._reparam(B, eps, sample)
        logits = self._decode(self.z, B)
        kl_floor = self.kl_tolerance * self.z_dim if self.kl_tolerance > 0 else 0.0
        self.L.mi_bce_logits_fwd_bwd(self.stream(), self.dtype, logits.data_ptr(), tgt.data_ptr(), milib.ptr(idx), self.P, B, self.P, self.loss_kind,
                                     float(inv_batch), self.g_d[-1].data_ptr() if (want_grad and self.with_optimizer) else None, self.partial.data_ptr())
        self.L.mi_vae_finalize_losses(self.stream(), self.partial.data_ptr(), self.nchunks, self.kl_row.data_ptr(), float(kl_floor), B, float(inv_batch),
                                      self.losses.data_ptr(), self.metrics.data_ptr() if accumulate_metrics else None, float(B * inv_batch))
        self.last_B = B

    def backward(self, src, idx, eps, inv_batch, part=0):
        """part 0 = everything, 1 = decoder half (+ dz), 2 = encoder half: the data-parallel host all-reduces grads[decoder_offset:] in between."""
        B, st, L, dt = self.last_B, self.stream(), self.L, self.dtype
        if B < 1:
            raise milib.MiError("MlpVaeDevice.backward: no forward pass recorded")
        if part in (0, 1):
            for i in range(len(self.dec) - 1, -1, -1):
                name, k, n = self.dec[i]
                gy = self.g_d[i]
                a = self.d[i - 1] if i > 0 else self.z
                L.mi_colsum(st, dt, gy.data_ptr(), B, n, self._p(self.grads, name + "/bias"))
                L.mi_gemm_wgrad(st, dt, a.data_ptr(), gy.data_ptr(), B, k, n, self._p(self.grads, name + "/kernel"))
                if i > 0:       # dx = gy W^T, ReLU-grad mask = the layer input (a ReLU output): W[k, n] read as [N_out = k][K_in = n]
                    self._dense(gy, B, n, name + "/kernel", k, None, 0, self.g_d[i - 1], w_layout=1, mask=self.d[i - 1])
                else:
                    self._dense(gy, B, n, name + "/kernel", k, None, 0, self.dz, out_f32=1, w_layout=1)
        if part in (0, 2):
            kl_floor = self.kl_tolerance * self.z_dim if self.kl_tolerance > 0 else 0.0
            L.mi_vae_reparam_kl_bwd(st, dt, self.dz.data_ptr(), 1, self.mean.data_ptr(), self.logvar.data_ptr(), milib.ptr(eps), self.kl_row.data_ptr(),
                                    self.beta, float(kl_floor), float(inv_batch), B, self.z_dim, self.dheads.data_ptr())
            kh = self.enc_sizes[-1]
            L.mi_colsum(st, dt, self.dheads.data_ptr(), B, 2 * self.z_dim, self._p(self.grads, "@heads/bias"))
            L.mi_gemm_wgrad(st, dt, self.h[-1].data_ptr(), self.dheads.data_ptr(), B, kh, 2 * self.z_dim, self._p(self.grads, "@heads/kernel"))
            self._dense(self.dheads, B, 2 * self.z_dim, "@heads/kernel", kh, None, 0, self.g_h[-1], w_layout=1, mask=self.h[-1])
            for i in range(len(self.enc) - 1, -1, -1):
                name, k, n = self.enc[i]
                gy = self.g_h[i]
                a = self.h[i - 1] if i > 0 else self._last_x
                L.mi_colsum(st, dt, gy.data_ptr(), B, n, self._p(self.grads, name + "/bias"))
                L.mi_gemm_wgrad(st, dt, a.data_ptr(), gy.data_ptr(), B, k, n, self._p(self.grads, name + "/kernel"))
                if i > 0:
                    self._dense(gy, B, n, name + "/kernel", k, None, 0, self.g_h[i - 1], w_layout=1, mask=self.h[i - 1])

    def apply_adam(self, alpha, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.L.mi_adam_tf_flat(self.stream(), self.params.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(), self.grads.data_ptr(), self.n_flat,
                               float(alpha), float(beta1), float(beta2), float(epsilon), self.shadow.data_ptr() if self.bf16 else None, 1)
        self._refresh_transposed()

    def encode(self, src, idx, B, out):
        self.ensure_batch(B)
        self._encode(self._stage_input(src, idx, int(B)), int(B))
        self._reparam(int(B), None, 0)
        out.copy_(self.mean[:int(B)])

    def decode(self, z, B, out):
        self.ensure_batch(B)
        B = int(B)
        zz = z.to(torch.float32).contiguous()
        if self.bf16:
            self.L.mi_cast_f32_to_bf16(self.stream(), zz.data_ptr(), self.z.data_ptr(), B * self.z_dim)
            zin = self.z
        else:
            zin = zz
        logits = self._decode(zin, B)
        self.L.mi_sigmoid(self.stream(), self.dtype, logits.data_ptr(), out.data_ptr(), B * self.P)

    def reconstruct(self, src, idx, B, eps, sample, out):
        self.ensure_batch(B)
        B = int(B)
        self._encode(self._stage_input(src, idx, B), B)
        self._reparam(B, eps, sample)
        logits = self._decode(self.z, B)
        self.L.mi_sigmoid(self.stream(), self.dtype, logits.data_ptr(), out.data_ptr(), B * self.P)

    def range_ok(self, t):
        flag = torch.zeros(1, device=self.device, dtype=torch.int32)
        self.L.mi_range_check(self.stream(), t.data_ptr(), t.numel(), 0.0, 1.0, flag.data_ptr())
        return int(flag.item()) == 0
