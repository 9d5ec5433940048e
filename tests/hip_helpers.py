"""Helpers for the GPU parity tests (test infrastructure)."""
import numpy as np
import torch

from mi355 import lib as milib

# "x3": split storage (precision "bf16x3", dtype code MI_BF16X3): one 4-byte element = bf16(x) << 16 | bf16(x - bf16(x)).  torch has no such
# dtype: device tensors of it are carried as torch.uint32 (allocation / views / copies only -- every conversion goes through split_encode /
# split_decode below, the host-side restatement of csrc/common.hpp split_from_f32 / split_to_f32).
X3 = torch.uint32
DT = {"f32": (milib.MI_F32, torch.float32), "bf16": (milib.MI_BF16, torch.bfloat16), "x3": (milib.MI_BF16X3, X3)}
DTS = ["f32", "bf16", "x3"]


def split_encode(a):
    """float32 array -> uint32 words hi << 16 | lo, hi = bf16(x) (round to nearest even), lo = bf16(x - hi)."""
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    hi = t.to(torch.bfloat16)
    lo = (t - hi.to(torch.float32)).to(torch.bfloat16)
    hb = hi.view(torch.int16).numpy().astype(np.uint16).astype(np.uint32)
    lb = lo.view(torch.int16).numpy().astype(np.uint16).astype(np.uint32)
    return (hb << np.uint32(16)) | lb


def split_decode(u):
    """uint32 words -> float64 values hi + lo (exact: both halves are bf16 values)."""
    u = np.ascontiguousarray(u).view(np.uint32)
    hi = (u & np.uint32(0xffff0000)).view(np.float32).astype(np.float64)
    lo = (u << np.uint32(16)).view(np.float32).astype(np.float64)
    return hi + lo


def alloc(tdtype, *shape, fill=None):
    """Device tensor of the storage type (an output buffer); fill: a float every element starts at (to detect unwritten elements)."""
    if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
        shape = tuple(shape[0])
    if tdtype is X3:
        if fill is None:
            return torch.empty(*shape, device="cuda", dtype=torch.int32).view(X3)
        word = int(split_encode(np.array([fill], np.float32)).view(np.int32)[0])
        return torch.full(tuple(shape), word, device="cuda", dtype=torch.int32).view(X3)
    if fill is None:
        return torch.empty(*shape, device="cuda", dtype=tdtype)
    return torch.full(tuple(shape), float(fill), device="cuda", dtype=tdtype)


def stream():
    return torch.cuda.current_stream().cuda_stream


def dev(a, tdtype=torch.float32):
    if tdtype is X3:
        a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
        return torch.from_numpy(split_encode(a).view(np.int32)).to(device="cuda").contiguous().view(X3)
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
    if tdtype is X3:
        return torch.from_numpy(split_decode(split_encode(a)))
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(tdtype)
    return t.to(torch.float64)


def host(t):
    torch.cuda.synchronize()
    if t.dtype == X3:
        return split_decode(t.detach().view(torch.int32).cpu().numpy())
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
    # split path ("x3"): inputs identical (pre-split); fp32 accumulation of exact bf16 x bf16 products + split output rounding (2^-18 rel): the fp32 limits
    return (1e-5, 2e-5 * scale) if dt in ("f32", "x3") else (4e-3, 4e-3 * scale)
