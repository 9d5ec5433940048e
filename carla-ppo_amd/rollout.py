"""One environment step of the rollout loop as ONE device call.

The reference does, per simulator step (vae_common.py:45-61, train.py:142, run_eval.py:54):

    frame = env.observation.astype(np.float32) / 255.0
    state = np.append(vae.encode([frame])[0], [steer, throttle, speed])          # sess.run #1, host round trip
    action, value = model.predict(state, write_to_summary=True)                  # sess.run #2, host round trip

RolloutStep does the same arithmetic in one C-ABI call (mi_rollout_step: raw uint8 frame -> /255 -> conv x 4 -> mean -> [z, measurements] ->
policy / value heads; exact fp32 on the master weights; eight launches).  By default nothing is copied: the frame bytes, the measurements and
the exploration noise sit in one pinned host buffer the first kernel reads over PCIe (38 KB), and the last kernel stores (action, value, z)
into pinned host memory; io="device" stages both through HBM with one copy each way (5 us slower on the measured box).
No CPU fallback: needs the HIP library and a GPU.

    step = RolloutStep(vae, ppo)
    action, value, state = step(env.observation, [steer, throttle, speed])       # state: float64 [z_dim + k], as np.append returns it

The split-K layers accumulate with fp32 atomics, so two calls on the same frame can differ in the last bit (1e-7 relative).
"""
import os

import numpy as np

from mi355 import lib as milib


class RolloutStep:
    def __init__(self, vae, ppo, seed=None, io=None):
        import torch
        self.vae, self.ppo = vae, ppo
        vdev, pdev = vae._need_dev(), ppo._need_dev()
        self.L = vdev.L
        self.device = vdev.device
        self.z_dim, self.A = int(vae.z_dim), int(ppo.num_actions)
        self.n_meas = int(ppo.input_dim) - self.z_dim
        if self.n_meas < 0:
            raise ValueError("the policy takes fewer inputs than the VAE's latent size")
        self.frame_bytes = int(np.prod(vdev.source_shape))
        self._noise_off = (self.frame_bytes + 15) // 16 * 16                         # float region: measurements, then noise
        nbytes = self._noise_off + 4 * (self.n_meas + self.A)
        self.h_in = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        self.d_in = None                                                             # (allocated below for io="device")
        self.h_out = torch.empty(self.A + 1 + self.z_dim, dtype=torch.float32).pin_memory()
        self.io = io or os.environ.get("MI355_ROLLOUT_IO", "pinned")
        if self.io not in ("pinned", "device"):
            raise ValueError("RolloutStep: io must be 'pinned' or 'device'")
        self.d_out = torch.empty(self.A + 1 + self.z_dim, dtype=torch.float32, device=self.device) if self.io == "device" else None
        self.d_in = torch.empty(nbytes, dtype=torch.uint8, device=self.device) if self.io == "device" else None
        self._in_np = self.h_in.numpy()
        self._f_np = self._in_np[self._noise_off:].view(np.float32)
        self._out_np = self.h_out.numpy()
        self._rng = np.random.Generator(np.random.Philox(int(seed if seed is not None else (ppo.seed or 0)) + 0xAC7))

    def __call__(self, frame_u8, measurements, greedy=False, noise=None):
        """frame_u8: uint8 [H, W, 3] camera frame; measurements: the k values appended to the latent.  Returns (action [A], value, state [z + k])."""
        import torch
        f = np.asarray(frame_u8)
        if f.dtype != np.uint8 or f.size != self.frame_bytes:
            raise ValueError("RolloutStep: expected a uint8 frame of %d bytes" % self.frame_bytes)
        meas = np.asarray(measurements, np.float64).reshape(-1)
        if meas.size != self.n_meas:
            raise ValueError("RolloutStep: expected %d measurements" % self.n_meas)
        self._in_np[:self.frame_bytes] = f.reshape(-1)
        self._f_np[:self.n_meas] = meas                                              # f64 -> f32 at the feed, as ppo.py:108-109
        if not greedy:
            self._f_np[self.n_meas:] = self._rng.standard_normal(self.A) if noise is None else np.asarray(noise, np.float32).reshape(self.A)
        st = torch.cuda.current_stream(self.device)
        if self.d_in is not None:
            self.d_in.copy_(self.h_in, non_blocking=True)
        base = (self.h_in if self.d_in is None else self.d_in).data_ptr()
        fptr = base + self._noise_off
        self.L.mi_rollout_step(self.vae.dev.handle, self.ppo.dev.handle, st.cuda_stream, base, fptr, self.n_meas,
                               None if greedy else fptr + 4 * self.n_meas, 1 if greedy else 0, (self.h_out if self.d_out is None else self.d_out).data_ptr())
        if self.d_out is not None:
            self.h_out.copy_(self.d_out, non_blocking=True)
        st.synchronize()
        o = self._out_np
        action, value = o[:self.A].copy(), float(o[self.A])
        state = np.append(o[self.A + 1:].copy(), meas)                               # float64, like np.append(float32[z], python floats)
        return action, value, state
