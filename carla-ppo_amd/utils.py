"""utils — drop-in for the reference's utils.py on the hot path: compute_gae (utils.py:45-50), plus the batched,
device-resident forms used by the synthetic-replay configuration (GAE + per-row advantage normalisation for
many independent trajectories, train.py:175-177 semantics per row).

The scan runs in fp64 on the GPU with the exact rounding sequence of numpy + scipy.signal.lfilter, so the result
is bit-identical to the reference's host computation.  No CPU fallback: raises without a GPU / built library.
"""
import numpy as np


def _dev():
    import torch
    from mi355.vae_device import require_gpu
    require_gpu()
    return torch, torch.device("cuda", torch.cuda.current_device())


def compute_gae_batched(rewards, values, terminals, gamma, lam, normalize=False):
    """rewards [R,T], values [R,T+1] (last column = bootstrap value), terminals [R,T] -> advantages [R,T] (fp64 numpy).
    normalize=True additionally returns (returns, normalised advantages) per row (train.py:176-177)."""
    from mi355 import lib as milib
    torch, device = _dev()
    L = milib.get()
    r = torch.as_tensor(np.ascontiguousarray(np.asarray(rewards, np.float64)), device=device)
    v = torch.as_tensor(np.ascontiguousarray(np.asarray(values, np.float64)), device=device)
    d = torch.as_tensor(np.ascontiguousarray(np.asarray(terminals, np.float64)), device=device)
    R, T = r.shape
    if v.shape != (R, T + 1) or d.shape != (R, T):
        raise ValueError("compute_gae_batched: rewards [R,T], values [R,T+1], terminals [R,T]")
    adv = torch.empty(R, T, device=device, dtype=torch.float64)
    st = torch.cuda.current_stream(device).cuda_stream
    L.mi_gae_scan(st, r.data_ptr(), v.data_ptr(), d.data_ptr(), R, T, float(gamma), float(lam), adv.data_ptr())
    if not normalize:
        return adv.cpu().numpy()
    raw = adv.clone()
    ret = torch.empty(R, T, device=device, dtype=torch.float64)
    L.mi_adv_normalize(st, adv.data_ptr(), v.data_ptr(), R, T, ret.data_ptr())
    return raw.cpu().numpy(), ret.cpu().numpy(), adv.cpu().numpy()


def gae_resident(rewards, values, terminals, gamma, lam):
    """Device-resident form for the replay path: `values` is a device tensor [R,T+1] (fp32 or fp64: the value network's own output, widened
    exactly on the device -- no round trip through the host); rewards / terminals host arrays [R,T].  Returns three fp64 DEVICE tensors
    [R,T]: raw advantages, returns, per-row normalised advantages (train.py:175-177), bit-identical to compute_gae_batched(normalize=True)."""
    from mi355 import lib as milib
    torch, device = _dev()
    L = milib.get()
    v = values.to(device=device, dtype=torch.float64).contiguous()
    R, T = v.shape[0], v.shape[1] - 1
    r = torch.as_tensor(np.ascontiguousarray(np.asarray(rewards, np.float64)), device=device)
    d = torch.as_tensor(np.ascontiguousarray(np.asarray(terminals, np.float64)), device=device)
    if r.shape != (R, T) or d.shape != (R, T):
        raise ValueError("gae_resident: rewards / terminals [R,T], values [R,T+1]")
    adv = torch.empty(R, T, device=device, dtype=torch.float64)
    st = torch.cuda.current_stream(device).cuda_stream
    L.mi_gae_scan(st, r.data_ptr(), v.data_ptr(), d.data_ptr(), R, T, float(gamma), float(lam), adv.data_ptr())
    raw = adv.clone()
    ret = torch.empty(R, T, device=device, dtype=torch.float64)
    L.mi_adv_normalize(st, adv.data_ptr(), v.data_ptr(), R, T, ret.data_ptr())
    return raw, ret, adv


def compute_gae(rewards, values, bootstrap_values, terminals, gamma, lam):
    """Reference signature and semantics (utils.py:45-50): lists of T rewards / values / terminals + one bootstrap
    value -> np.ndarray [T] float64.  (No done-mask inside the recursion, like the reference.)"""
    rewards = np.asarray(rewards, np.float64)
    vals = np.asarray(list(values) + [bootstrap_values], np.float64)     # np.float32 scalars upcast exactly
    term = np.asarray(terminals, np.float64)
    if rewards.ndim != 1:
        raise ValueError("compute_gae expects 1-D sequences")
    if len(rewards) == 0:
        return np.zeros(0, np.float64)
    return compute_gae_batched(rewards[None], vals[None], term[None], gamma, lam)[0]


def normalize_advantages(advantages, values):
    """train.py:176-177 on the device: returns, (A - mean) / (std + 1e-8) with population std (fp64)."""
    from mi355 import lib as milib
    torch, device = _dev()
    L = milib.get()
    a = torch.as_tensor(np.ascontiguousarray(np.asarray(advantages, np.float64))[None], device=device).clone()
    T = a.shape[1]
    v = torch.as_tensor(np.ascontiguousarray(np.asarray(list(values) + [0.0], np.float64))[None], device=device)
    ret = torch.empty(1, T, device=device, dtype=torch.float64)
    L.mi_adv_normalize(torch.cuda.current_stream(device).cuda_stream, a.data_ptr(), v.data_ptr(), 1, T, ret.data_ptr())
    return ret.cpu().numpy()[0], a.cpu().numpy()[0]


class VideoRecorder():
    """Reference utils.py:9-23 wraps cv2.VideoWriter; video I/O is outside the hot path."""

    def __init__(self, filename, frame_size, fps=30):
        try:
            import cv2
        except ImportError as e:
            raise ImportError("VideoRecorder needs OpenCV (cv2); it is not part of the MI355X hot path") from e
        self._cv2 = cv2
        self.video_writer = cv2.VideoWriter(filename, cv2.VideoWriter_fourcc(*"MPEG"), int(fps), (frame_size[1], frame_size[0]))

    def add_frame(self, frame):
        self.video_writer.write(self._cv2.cvtColor(frame, self._cv2.COLOR_RGB2BGR))

    def release(self):
        self.video_writer.release()
