"""Helpers for the GPU parity tests (test infrastructure)."""
import numpy as np
import torch

from mi355 import lib as milib

DT = {"f32": (milib.MI_F32, torch.float32), "bf16": (milib.MI_BF16, torch.bfloat16)}


def stream():
    return torch.cuda.current_stream().cuda_stream


def dev(a, tdtype=torch.float32):
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device="cuda", dtype=tdtype).contiguous()


_KEEP = []


def P(t):
    """data_ptr of a device tensor that is kept alive until the next keep_reset() (async launches must not see the
    caching allocator hand a temporary's block to the next temporary)."""
    _KEEP.append(t)
    return t.data_ptr()


def keep_reset():
    torch.cuda.synchronize()
    _KEEP.clear()


def rounded(a, tdtype):
    """numpy fp32 array rounded through the storage dtype (what the kernel actually reads) as float64."""
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(tdtype)
    return t.to(torch.float64)


def host(t):
    torch.cuda.synchronize()
    return t.detach().to("cpu", torch.float64).numpy()


def assert_close(got, ref, rtol, atol, what=""):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = np.abs(got - ref)
    tol = atol + rtol * np.abs(ref)
    if not (err <= tol).all() or not np.isfinite(got).all():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        nbad = int((err > tol).sum())
        raise AssertionError("%s: %d/%d outside tol; worst at %s got %.6g ref %.6g (|ref|max %.4g)" %
                             (what, nbad, err.size, i, got[i], ref[i], np.abs(ref).max()))


def tols(dt, scale=1.0):
    # fp32 path: exact-fp32 MFMA, accumulation order differs from the reference -> 2e-5 of the output scale
    # bf16 path: inputs identical (pre-rounded); only fp32 accumulation order + bf16 output rounding (2^-9 rel)
    return (1e-5, 2e-5 * scale) if dt == "f32" else (4e-3, 4e-3 * scale)
