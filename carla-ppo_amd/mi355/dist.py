"""Data parallelism: one process per GPU; gradients are summed across ranks by RCCL over xGMI.  The reference has no distributed code at
all — this is the new component BASELINE configs 4-5 ask for.

Two transports behind the same four functions (all_reduce_sum, broadcast, barrier, world_size):
  * the library's own communicator (csrc/comm.hip: mi_comm_init / mi_allreduce_sum_f32[_async] / mi_comm_wait / mi_broadcast), which calls RCCL
    directly on the engine's stream — used for fp32 device tensors whenever the process group's backend is "nccl" (MI355_COMM=torch turns it
    off).  torch.distributed is then only the rendezvous: it carries the 128-byte RCCL id from rank 0 to the others and reduces host-side
    scalars.  Before it is trusted the communicator sums a known vector and every rank checks the result; any failure (library missing,
    init error, wrong sum) falls back, on all ranks together, to
  * torch.distributed collectives (backend "nccl" = the same RCCL; "gloo" in the CPU tests, where there is no device buffer to reduce).

Scheme: every rank draws the SAME host permutation (same legacy-numpy seed), takes rows
[r*B/W, (r+1)*B/W) of each global minibatch, computes gradients of  sum_local(loss_i) / B_global,  and the flat fp32
gradient buffer is summed across ranks.  Parameters stay replicated: broadcast once from rank 0, then identical
fused-Adam steps everywhere.  No other exchange is needed: VAE encode, GAE and per-row advantage normalisation are
row-local (PPO rows shard across ranks with no collective).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(backend=None):
    """Initialise the default process group from torchrun-style env vars; no-op for a single process."""
    world, rank, local = env_world()
    if world <= 1:
        return 1, 0, 0
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


class _MiComm:
    """The C-ABI communicator of this process (include/mi355_carla.h, collectives section)."""

    def __init__(self, handle, L):
        self.handle, self.L = handle, L

    def usable(self, t):
        return t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()

    def all_reduce(self, t, async_op=False):
        st = torch.cuda.current_stream(t.device).cuda_stream
        if async_op:
            self.L.mi_allreduce_sum_f32_async(self.handle, st, t.data_ptr(), t.numel())
            return self
        self.L.mi_allreduce_sum_f32(self.handle, st, t.data_ptr(), t.numel())
        return None

    def wait(self):                                    # joins every bucket queued so far (device-side; later handles find nothing pending)
        self.L.mi_comm_wait(self.handle, torch.cuda.current_stream().cuda_stream)

    def broadcast(self, t, src):
        self.L.mi_broadcast(self.handle, torch.cuda.current_stream(t.device).cuda_stream, t.data_ptr(), t.numel() * t.element_size(), src)


_mi_comm = None
_mi_comm_tried = False
comm_note = "single process"                          # which transport carries the gradient all-reduce (bench.py reports it)


def _all_agree(ok, device):
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def mi_comm():
    """The library communicator, created on first use from the default process group (backend nccl only); None when torch carries the traffic."""
    global _mi_comm, _mi_comm_tried, comm_note
    if _mi_comm_tried or world_size() == 1:
        return _mi_comm
    _mi_comm_tried = True
    backend = dist.get_backend()
    comm_note = "torch.distributed (%s)" % backend
    if backend != "nccl" or os.environ.get("MI355_COMM", "rccl") == "torch" or not torch.cuda.is_available():
        return None
    import ctypes
    from mi355 import lib as milib
    dev = torch.device("cuda", torch.cuda.current_device())
    W, r = dist.get_world_size(), dist.get_rank()
    # 1. everything that can fail locally, agreed on before the collective init (a rank that cannot load RCCL must not leave the others waiting)
    L, nbytes, ok = None, 128, True
    try:
        L = milib.get()
        nbytes = L.mi_comm_id_bytes()
        L.mi_comm_probe()                               # EVERY rank binds librccl here (mi_comm_id_bytes is a sizeof and loads nothing)
    except Exception:
        ok = False
    idt = torch.zeros(nbytes, dtype=torch.uint8)
    if ok and r == 0:
        try:
            L.mi_comm_unique_id(idt.data_ptr())
        except Exception:
            ok = False
    if not _all_agree(ok, dev):
        return None
    idd = idt.to(dev)
    dist.broadcast(idd, src=0)                          # the rendezvous: 128 bytes over the channel the process group already has
    idt = idd.cpu()
    # 2. collective init, then a known sum checked on every rank
    handle = ctypes.c_void_p()
    try:
        L.mi_comm_init(ctypes.addressof(handle), r, W, idt.data_ptr())
        c = _MiComm(handle, L)
        # the bucket schedule is a property of the communicator and must be the same everywhere: every rank states what its environment asks for, the
        # minimum wins (one rank without MI355_COMM_ALGO=rsag keeps all of them on ncclAllReduce), and only then is it set
        # ... and what its bound library can do (a rank whose RCCL lacks ReduceScatter / AllGather would fail set_algo ALONE and skip the probe collectives
        # below while the others enter them): the agreed value is MIN over ranks of (environment asks for rsag AND the library has both entry points)
        want_rsag = os.environ.get("MI355_COMM_ALGO") == "rsag" and bool(L.mi_comm_has_rsag())
        want_algo = torch.tensor([1 if want_rsag else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(want_algo, op=dist.ReduceOp.MIN)
        algo = int(want_algo.item())
        if algo:
            L.mi_comm_set_algo(handle, algo)                # (cannot fail on one rank only any more)
        # known sums through BOTH entry points, sized so that the selected schedule is the one exercised (rank slices of >= 1024 floats at any world size)
        n_probe = 4096 * W
        probe = torch.full((n_probe,), float(r + 1), dtype=torch.float32, device=dev)
        c.all_reduce(probe)
        c.all_reduce(probe[:n_probe // 2], async_op=True).wait()
        torch.cuda.synchronize(dev)
        want = W * (W + 1) / 2.0
        ok = bool((probe[n_probe // 2:] == want).all().item()) and bool((probe[:n_probe // 2] == want * W).all().item())
        b = torch.full((256,), float(r), dtype=torch.float32, device=dev)
        c.broadcast(b, 0)
        torch.cuda.synchronize(dev)
        ok = ok and bool((b == 0).all().item())
    except Exception:
        ok, c, algo = False, None, 0
    if not _all_agree(ok, dev):
        return None
    _mi_comm = c
    comm_note = "mi_comm (RCCL through the C ABI, on the engine's stream%s)" % ("; reduce-scatter + all-gather schedule" if algo else "")
    if not _atexit_registered[0]:
        import atexit
        atexit.register(shutdown)
        _atexit_registered[0] = True
    return _mi_comm


_atexit_registered = [False]
_recording = [None]


def recording_comm():
    """A communicator of the C ABI that records instead of communicating (mi_comm_init_recording): bench.py times the data-parallel step WITHOUT its collectives
    through the very same C call (mi_vae_train_step_dp) by handing it this one.  The log holds the first 64 entries; the library's counter saturates (nobody reads either)."""
    if _recording[0] is None:
        import ctypes
        import numpy as np
        from mi355 import lib as milib
        L = milib.get()
        log = np.zeros((64, 4), np.int64)
        h = ctypes.c_void_p()
        L.mi_comm_init_recording(ctypes.addressof(h), rank(), world_size(), log.ctypes.data, 64)
        c = _MiComm(h, L)
        c._log = log                                    # keeps the host buffer alive as long as the handle
        _recording[0] = c
    return _recording[0]


def shutdown():
    """Tears the library communicator down (ncclCommDestroy, its side stream and events) and forgets it, so that a later process group gets a
    fresh one.  Call it before dist.destroy_process_group(); registered with atexit once a communicator exists."""
    global _mi_comm, _mi_comm_tried, comm_note
    c, _mi_comm, _mi_comm_tried = _mi_comm, None, False
    comm_note = "single process"
    rec, _recording[0] = _recording[0], None            # the recording communicator was created for THIS process group's rank / world: a later group gets a fresh one
    if rec is not None:
        try:
            rec.L.mi_comm_destroy(rec.handle)
        except Exception:
            pass
    if c is not None:
        try:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            c.L.mi_comm_destroy(c.handle)
        except Exception:
            pass                                        # at interpreter exit the device context may already be gone


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_bounds(n, r=None, w=None):
    """Contiguous [lo, hi) slice of n minibatch rows owned by rank r of w (n need not divide evenly)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    base, rem = divmod(n, w)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


def shard(mb_idx, r=None, w=None):
    lo, hi = shard_bounds(len(mb_idx), r, w)
    return mb_idx[lo:hi]


def all_reduce_sum(t, async_op=False):
    """In-place sum over ranks of a flat tensor (gradient bucket / metric accumulators)."""
    if world_size() == 1:
        return None
    c = mi_comm()
    if c is not None and c.usable(t):
        return c.all_reduce(t, async_op)
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)


def broadcast(t, src=0):
    if world_size() > 1:
        c = mi_comm()
        if c is not None and c.usable(t):
            c.broadcast(t, src)
        else:
            dist.broadcast(t, src=src)


def barrier():
    if world_size() > 1:
        dist.barrier()
