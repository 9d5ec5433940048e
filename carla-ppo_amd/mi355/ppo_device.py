"""Device side of the PPO model: HBM buffers (torch tensors as plumbing) + the native engine (csrc/ppo_engine.hip).
No arithmetic here; no CPU fallback."""
import ctypes
from collections import OrderedDict

import numpy as np
import torch

from . import lib as milib
from .init import ppo_variables
from .vae_device import require_gpu


class PpoDevice:
    def __init__(self, input_dim, num_actions, action_low, action_high, clip_eps, value_scale, entropy_scale,
                 hidden=(500, 300), max_batch=256, device=None):
        require_gpu()
        self.L = milib.get()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.input_dim, self.num_actions, self.hidden = int(input_dim), int(num_actions), tuple(hidden)
        self.clip_eps, self.value_scale, self.entropy_scale = float(clip_eps), float(value_scale), float(entropy_scale)
        self.low = np.ascontiguousarray(np.asarray(action_low, np.float32).reshape(-1))
        self.high = np.ascontiguousarray(np.asarray(action_high, np.float32).reshape(-1))
        self.variables = ppo_variables(self.input_dim, self.num_actions, self.hidden)
        self.kin = (self.input_dim + 7) // 8 * 8
        d = self._desc(1)
        n = self.L.mi_ppo_param_floats(ctypes.byref(d))
        if n <= 0:
            raise milib.MiError("mi_ppo_param_floats: " + self.L.cdll.mi_last_error().decode())
        self.n_flat = int(n)
        cnt = self.L.mi_ppo_tensor_count()
        off, size = np.zeros(cnt, np.int64), np.zeros(cnt, np.int64)
        self.L.mi_ppo_param_layout(ctypes.byref(d), off.ctypes.data, size.ctypes.data, cnt)
        self.layout = OrderedDict((name, (int(o), int(s))) for name, o, s in zip(self.variables, off, size))
        z = lambda: torch.zeros(self.n_flat, device=self.device)   # noqa: E731
        self.params, self.params_old, self.grads, self.adam_m, self.adam_v = z(), z(), z(), z(), z()
        self.handle = None
        self.max_batch = 0
        self._create(max_batch)

    def _desc(self, max_batch):
        return milib.MiPpoDesc(int(max_batch), self.input_dim, self.num_actions, self.hidden[0], self.hidden[1],
                               self.clip_eps, self.value_scale, self.entropy_scale)

    def _create(self, max_batch):
        if self.handle is not None:
            self.L.mi_ppo_destroy(self.handle)
            self.handle = None
        d = self._desc(max_batch)
        nbytes = int(self.L.mi_ppo_workspace_bytes(ctypes.byref(d)))
        self.workspace = torch.empty(nbytes, device=self.device, dtype=torch.uint8)
        p = milib.ptr
        self.handle = self.L.mi_ppo_create(ctypes.byref(d), p(self.params), p(self.params_old), p(self.grads), p(self.adam_m), p(self.adam_v),
                                           p(self.workspace), nbytes, self.low.ctypes.data, self.high.ctypes.data)
        if not self.handle:
            raise milib.MiError("mi_ppo_create: " + self.L.cdll.mi_last_error().decode())
        self.max_batch = int(max_batch)
        addr = self.L.mi_ppo_buffer(self.handle, 0)
        o = addr - self.workspace.data_ptr()
        # [policy, value, entropy, total, mean prob ratio, mean action_mean[A], std[A]] of the last minibatch step
        self.losses = self.workspace[o:o + 4 * (5 + 2 * self.num_actions)].view(torch.float32)
        addr = self.L.mi_ppo_buffer(self.handle, 1)
        o = addr - self.workspace.data_ptr()
        self.action_mean = self.workspace[o:o + 4 * int(max_batch) * self.num_actions].view(torch.float32).view(int(max_batch), self.num_actions)

    def ensure_batch(self, m):
        if m > self.max_batch:
            torch.cuda.synchronize(self.device)
            self._create(max(m, 2 * self.max_batch))

    def close(self):
        if self.handle is not None:
            self.L.mi_ppo_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # ---- TF-named variables <-> flat layout (first-layer kernels are zero-padded from input_dim to kin rows) ----
    def _to_flat(self, named, scope="policy"):
        flat = np.zeros(self.n_flat, np.float32)
        for name, (o, s) in self.layout.items():
            a = np.asarray(named[name.replace("policy/", scope + "/", 1)], np.float32)
            if tuple(a.shape) != tuple(self.variables[name]):
                raise ValueError("%s: shape %s, expected %s" % (name, a.shape, self.variables[name]))
            if name.endswith(("dense/kernel", "dense_2/kernel")):
                pad = np.zeros((self.kin, a.shape[1]), np.float32)
                pad[:a.shape[0]] = a
                a = pad
            flat[o:o + s] = a.reshape(-1)
        return flat

    def _from_flat(self, flat, scope="policy"):
        out = OrderedDict()
        for name, shape in self.variables.items():
            o, s = self.layout[name]
            if name.endswith(("dense/kernel", "dense_2/kernel")):
                a = flat[o:o + s].reshape(self.kin, shape[1])[:shape[0]].copy()
            else:
                a = flat[o:o + s].reshape(shape).copy()
            out[name.replace("policy/", scope + "/", 1)] = a
        return out

    def load_params(self, named, old_named=None):
        self.params.copy_(torch.from_numpy(self._to_flat(named)))
        if old_named is not None:
            self.params_old.copy_(torch.from_numpy(self._to_flat(old_named, "policy_old")))

    def load_slots(self, m_named, v_named):
        self.adam_m.copy_(torch.from_numpy(self._to_flat(m_named)))
        self.adam_v.copy_(torch.from_numpy(self._to_flat(v_named)))

    def export_params(self):
        return self._from_flat(self.params.cpu().numpy())

    def export_old(self):
        return self._from_flat(self.params_old.cpu().numpy(), "policy_old")

    def export_slots(self):
        return self._from_flat(self.adam_m.cpu().numpy()), self._from_flat(self.adam_v.cpu().numpy())

    def export_grads(self):
        return self._from_flat(self.grads.cpu().numpy())

    # ---- steps ----
    def update_old(self):
        self.L.mi_ppo_update_old(self.handle, self.stream())

    def predict(self, states, M, noise, greedy, action, value):
        self.ensure_batch(M)
        p = milib.ptr
        self.L.mi_ppo_predict(self.handle, self.stream(), p(states), int(M), p(noise), int(greedy), p(action), p(value))

    def forward_backward(self, states, actions, returns, advantage, M, inv_m, grad_scale):
        self.ensure_batch(M)
        p = milib.ptr
        self.L.mi_ppo_forward_backward(self.handle, self.stream(), p(states), p(actions), p(returns), p(advantage), int(M), float(inv_m), float(grad_scale))

    def train_step(self, states, actions, returns, advantage, M, inv_m, grad_scale, alpha, beta1=0.9, beta2=0.999, epsilon=1e-8, logp_old=None):
        """The whole minibatch step in one C call (single rank): fused forward / losses / backward / Adam (csrc/ppo_fused.hip)."""
        self.ensure_batch(M)
        p = milib.ptr
        self.L.mi_ppo_train_step(self.handle, self.stream(), p(states), p(actions), p(returns), p(advantage), p(logp_old), int(M), float(inv_m), float(grad_scale),
                                 float(alpha), float(beta1), float(beta2), float(epsilon))

    def train_step_idx(self, states, actions, returns, advantage, logp_old, row_idx, M, inv_m, grad_scale, alpha, beta1=0.9, beta2=0.999, epsilon=1e-8):
        """train_step on rows `row_idx` (int32 device tensor [M]) of the horizon-batch tables: the gather happens inside the kernels."""
        self.ensure_batch(M)
        p = milib.ptr
        self.L.mi_ppo_train_step_idx(self.handle, self.stream(), p(states), p(actions), p(returns), p(advantage), p(logp_old), p(row_idx), int(states.shape[0]), int(M),
                                     float(inv_m), float(grad_scale), float(alpha), float(beta1), float(beta2), float(epsilon))

    def train_step_dp(self, comm_handle, states, actions, returns, advantage, logp_old, row_idx, M, inv_m, grad_scale, alpha, beta1=0.9, beta2=0.999, epsilon=1e-8):
        """One DATA-PARALLEL minibatch step in one C call (mi_ppo_train_step_dp, round 6): the fused chain on this rank's M rows, one all-reduce of the flat gradient
        buffer through the library communicator, Adam.  row_idx (int32 device tensor [M]) names rows of the horizon-batch tables (gather inside the kernels) or is None
        (contiguous minibatch tensors)."""
        self.ensure_batch(M)
        p = milib.ptr
        self.L.mi_ppo_train_step_dp(self.handle, comm_handle, self.stream(), p(states), p(actions), p(returns), p(advantage), p(logp_old), p(row_idx),
                                    int(states.shape[0]), int(M), float(inv_m), float(grad_scale), float(alpha), float(beta1), float(beta2), float(epsilon))

    def fused_ok(self):
        """True when the fused kernels (in-kernel minibatch gather, cached log pi_old) take this engine's shape; else only the per-layer path runs."""
        return bool(self.L.mi_ppo_fused_shape_ok(self.handle))

    def logp_old(self, states, actions, M, out):
        """log pi_old(a | s) of M samples under theta_old (computed once per horizon batch; theta_old only changes in update_old())."""
        self.ensure_batch(M)
        p = milib.ptr
        self.L.mi_ppo_logp_old(self.handle, self.stream(), p(states), p(actions), int(M), p(out))

    def apply_adam(self, alpha, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.L.mi_ppo_apply_adam(self.handle, self.stream(), float(alpha), float(beta1), float(beta2), float(epsilon))
