"""BASELINE configs[3]-shaped data parallelism on what one GPU allows: the C-ABI communicator against the real RCCL at world size 1, the whole product path with two
ranks sharing this GPU (gloo) or one device per rank (nccl, needs two devices), and bench.py's --gpus 2 launch exactly as the driver issues it.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vae_oracle as vo  # noqa: E402
from vae.models import ConvVAE, MlpVAE, bce_loss, bce_loss_v2, mse_loss  # noqa: E402,F401
from vae_gpu_common import synth_frames, make, rel_err, trained_like_params, _dev_table, _mlp_params  # noqa: E402,F401


def test_comm_c_abi_one_rank_world():
    """The collective half of the C ABI against the real RCCL on this box, world size 1 (what one GPU allows): rendezvous id, communicator,
    in-stream and side-stream all-reduce (identity at one rank), the join, broadcast, destroy.  librccl.so.1 is bound at run time."""
    import ctypes
    from mi355 import lib as milib
    L = milib.get()
    assert L.mi_comm_id_bytes() == 128
    idb = np.zeros(128, np.uint8)
    L.mi_comm_unique_id(idb.ctypes.data)
    assert idb.any()
    h = ctypes.c_void_p()
    L.mi_comm_init(ctypes.addressof(h), 0, 1, idb.ctypes.data)
    try:
        st = torch.cuda.current_stream().cuda_stream
        x = torch.randn(100003, device="cuda")
        ref = x.clone()
        L.mi_allreduce_sum_f32(h, st, x.data_ptr(), x.numel())
        y = x * 2                                           # work queued behind the bucket on the caller's stream
        L.mi_allreduce_sum_f32_async(h, st, x.data_ptr(), 4096)
        L.mi_allreduce_sum_f32_async(h, st, x.data_ptr() + 4 * 4096, x.numel() - 4096)
        L.mi_comm_wait(h, st)
        L.mi_comm_wait(h, st)                               # nothing pending: no-op
        L.mi_broadcast(h, st, x.data_ptr(), x.numel() * 4, 0)
        torch.cuda.synchronize()
        assert torch.equal(x, ref) and torch.equal(y, ref * 2)
        with pytest.raises(milib.MiError):
            L.mi_broadcast(h, st, x.data_ptr(), 16, 3)      # root outside the communicator
    finally:
        L.mi_comm_destroy(h)


def _run_two_ranks(tmp_path, backend, port):
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "dp")
    os.makedirs(out)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", port, os.path.join(ROOT, "tests", "dp_gpu_worker.py"), out, backend], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(os.path.join(out, "rank0.npz")), np.load(os.path.join(out, "rank1.npz"))


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_data_parallel_two_ranks_on_the_gpu(tmp_path, backend):
    """The product's data-parallel path end to end with world_size 2 (tests/dp_gpu_worker.py): the all-reduced gradients of a global minibatch
    equal the single-process gradients of the same minibatch, the epoch metrics agree, both ranks end with identical parameters, and those
    equal the single-process result up to Adam's sensitivity where |g| ~ 1e-8.
    gloo: two ranks share this GPU (RCCL refuses two ranks on one device; gloo reduces device tensors through the host).
    nccl: one device per rank, gradients summed by the library's own communicator (mi_comm, RCCL through the C ABI); needs two devices."""
    import dp_gpu_worker as W
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("the RCCL run needs two devices; this box has %d" % torch.cuda.device_count())
    r0, r1 = _run_two_ranks(tmp_path, backend, "29541" if backend == "gloo" else "29543")
    assert int(r0["world"]) == 2
    if backend == "nccl":
        assert str(r0["comm"]).startswith("mi_comm"), str(r0["comm"])
    frames, eps = W.dataset()
    m = W.build(str(tmp_path / "single"), trained_like_params(2))
    grads, losses, params = W.run(m, frames, eps)
    for k, g in grads.items():
        key = "g|" + k.replace("/", "|")
        assert np.array_equal(r0[key], r1[key]), k                                   # the same reduced buffer on both ranks
        assert rel_err(r0[key], g) < 2e-5, (k, rel_err(r0[key], g))
    assert np.allclose(r0["losses"], losses, rtol=1e-5) and np.array_equal(r0["losses"], r1["losses"])
    for k, v in params.items():
        key = "p|" + k.replace("/", "|")
        assert np.array_equal(r0[key], r1[key]), k                                   # replicas stay bit-identical
        assert rel_err(r0[key], v) < 2e-2, (k, rel_err(r0[key], v))
    # PPO.train under data parallelism (each rank passes its 16 of 32 rows): the global loss scalars and the replicas against one process on all 32
    pl, pp = W.run_ppo(str(tmp_path / "ppo_single"))
    assert np.array_equal(r0["ppo_params"], r1["ppo_params"]) and np.array_equal(r0["ppo_losses"], r1["ppo_losses"])
    assert np.allclose(r0["ppo_losses"], pl, rtol=2e-4, atol=1e-6), (r0["ppo_losses"], pl)
    assert rel_err(r0["ppo_params"], pp) < 1e-4, rel_err(r0["ppo_params"], pp)


def test_bench_data_parallel_path_with_two_ranks_on_one_gpu(tmp_path):
    """bench.py --gpus 2 exactly as the driver launches it (torch.distributed.run, two ranks), with the two ranks sharing this GPU over gloo
    (RCCL refuses two ranks on one device): the weak-scaling bookkeeping, the per-rank gather, the exposed all-reduce measurement and the
    parameter re-broadcast run end to end and the one JSON line has the contract's fields."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", MI355_BENCH_BACKEND="gloo", MI355_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29547",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "3", "--batch", "64", "--pool", "256"],
                       env=env, capture_output=True, text=True, timeout=300)      # (seconds: a rank-count mismatch in any loop of the bench is a hang, and must fail here, fast)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2"
    assert d["value"] == pytest.approx(128 * 4 / (d["ms_per_step"] * 4e-3), rel=1e-6)
    dp = d["data_parallel"]
    assert len(dp["ms_per_step_by_rank"]) == 2 and dp["gradient_bytes_per_step"] > 0 and "transport" in dp
    assert d["roofline"] is not None and d["cpu_baseline"] is None


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_dp_step_as_one_c_call_equals_the_host_loop(tmp_path, precision):
    """Round 5: mi_vae_train_step_dp -- forward, the three backward parts with each finished bucket queued on the communicator, the join, Adam -- as
    ONE C call, driven here with a RECORDING communicator (what one GPU allows: it logs what it would issue and leaves the data alone = the sum over one rank).
    Two SGD steps: the parameters are bitwise those of the host-sequenced loop (forward, backward(part) x 3, Adam: the path vae/models.py used to run per step), and the
    log is the bucket schedule of mi_vae_dp_buckets -- three async all-reduces over [lo, hi) of the gradient buffer, one join -- per step."""
    import ctypes
    from mi355 import lib as milib
    from vae.models import adam_alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON
    L = milib.get()
    B = 48
    frames = synth_frames(B, seed=77)
    eps = np.random.RandomState(5).standard_normal((2, B, 64)).astype(np.float32)
    params = trained_like_params(4)
    res = []
    for mode in ("c_call", "host_loop"):
        m = make(tmp_path / (precision + mode), precision, params=params, seed=0)
        dev = m.dev
        src = m._frames(frames, 38400, "src")
        log = np.zeros((16, 4), np.int64)
        h = ctypes.c_void_p()
        L.mi_comm_init_recording(ctypes.addressof(h), 0, 2, log.ctypes.data, 16)
        b1p, b2p = np.float32(ADAM_BETA1), np.float32(ADAM_BETA2)
        for s in range(2):
            e = m._eps(B, eps[s])
            alpha = adam_alpha(1e-4, b1p, b2p)
            if mode == "c_call":
                dev.train_step_dp(h, src, src, None, B, 0.5 / B, e, alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON)
            else:
                dev.forward(src, src, None, B, 0.5 / B, e, 1, 1)
                for part, lo, hi in dev.grad_buckets:
                    dev.backward(src, None, e, 0.5 / B, part=part)
                dev.apply_adam(alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON)
            b1p, b2p = np.float32(b1p * np.float32(ADAM_BETA1)), np.float32(b2p * np.float32(ADAM_BETA2))
        torch.cuda.synchronize()
        n = L.mi_comm_recorded(h)
        L.mi_comm_destroy(h)
        res.append((dev.export_params(), dev.losses.cpu().numpy().copy(), log[:max(n, 0)].copy(), dev.grads.data_ptr(), list(dev.grad_buckets)))
        m.dev.close()
    (p_c, l_c, log_c, gptr, buckets), (p_h, l_h, log_h, _, _) = res
    assert len(log_h) == 0 and len(log_c) == 8
    want = [(1, hi - lo, 1, gptr + 4 * lo) for (_, lo, hi) in buckets] + [(5, 3, 0, 0)]
    assert [tuple(int(x) for x in e) for e in log_c] == want + want
    assert np.array_equal(l_c, l_h)
    for k in p_h:
        assert np.array_equal(p_c[k], p_h[k]), k


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_dp_step_c_call_through_the_real_rccl_at_one_rank(tmp_path, precision):
    """mi_vae_train_step_dp with a REAL communicator (RCCL bound at run time, world size 1 = what one GPU allows): the three gradient buckets travel through
    ncclAllReduce on the communicator's side stream (the identity at one rank) and the caller's stream joins them before Adam.  Three SGD steps: parameters and losses are
    BITWISE those of the same call on the recording communicator (which leaves the buffers alone) -- the collective plumbing (side stream, events, join) neither loses,
    repeats nor reorders an update -- and equal the single-process step (mi_vae_train_step: one backward pass instead of three parts) to rounding."""
    import ctypes
    from mi355 import lib as milib
    from vae.models import adam_alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON
    L = milib.get()
    B = 64
    frames = synth_frames(B, seed=91)
    eps = np.random.RandomState(6).standard_normal((3, B, 64)).astype(np.float32)
    params = trained_like_params(5)
    res = {}
    for mode in ("rccl", "recording", "single"):
        m = make(tmp_path / (precision + mode), precision, params=params, seed=0)
        dev = m.dev
        src = m._frames(frames, 38400, "src")
        h = ctypes.c_void_p()
        log = np.zeros((32, 4), np.int64)
        if mode == "rccl":
            idb = np.zeros(128, np.uint8)
            L.mi_comm_unique_id(idb.ctypes.data)
            L.mi_comm_init(ctypes.addressof(h), 0, 1, idb.ctypes.data)
        elif mode == "recording":
            L.mi_comm_init_recording(ctypes.addressof(h), 0, 1, log.ctypes.data, 32)
        try:
            b1p, b2p = np.float32(ADAM_BETA1), np.float32(ADAM_BETA2)
            for s in range(3):
                e = m._eps(B, eps[s])
                alpha = adam_alpha(1e-4, b1p, b2p)
                if mode == "single":
                    dev.train_step(src, src, None, B, 1.0 / B, e, alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON)
                else:
                    dev.train_step_dp(h, src, src, None, B, 1.0 / B, e, alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON)
                b1p, b2p = np.float32(b1p * np.float32(ADAM_BETA1)), np.float32(b2p * np.float32(ADAM_BETA2))
            torch.cuda.synchronize()
            res[mode] = (dev.export_params(), dev.losses.cpu().numpy().copy())
        finally:
            if mode != "single":
                L.mi_comm_destroy(h)
            m.dev.close()
    (p_r, l_r), (p_c, l_c), (p_s, l_s) = res["rccl"], res["recording"], res["single"]
    assert np.isfinite(l_r).all() and np.array_equal(l_r, l_c), (l_r, l_c)
    for k in p_c:
        assert np.array_equal(p_r[k], p_c[k]), k
        assert rel_err(p_r[k], p_s[k]) < 1e-5, (k, rel_err(p_r[k], p_s[k]))
    assert np.allclose(l_r, l_s, rtol=1e-5, atol=1e-6), (l_r, l_s)


def _ppo_pair(tmp_path, tag):
    from oracle import ppo_oracle as po
    from ppo import PPO
    space = po.ActionSpace()
    hp = dict(learning_rate=1e-4, lr_decay=1.0, epsilon=0.2, value_scale=1.0, entropy_scale=0.01, initial_std=1.0)
    o = po.OraclePPO([67], space, seed=2, **hp)
    rng = np.random.RandomState(3)
    for k in o.params:                                   # theta != theta_old: ratio != 1, some samples clipped
        o.params[k] = o.params[k] + (0.02 * rng.standard_normal(o.params[k].shape)).astype(np.float32)
    m = PPO(np.array([67]), space, model_dir=str(tmp_path / tag), seed=2, **hp)
    m.set_weights(o.params)
    m.init_session(init_logging=False)
    return m


@pytest.mark.parametrize("M,gather", [(32, True), (32, False), (77, True), (300, True)])
def test_ppo_dp_step_as_one_c_call_equals_the_host_sequence(tmp_path, M, gather):
    """Round 6: mi_ppo_train_step_dp -- the fused launch chain with the gradients of this rank's rows left in the flat buffer, ONE all-reduce of that buffer
    through the communicator, tf.train.AdamOptimizer -- as ONE C call, with the minibatch gather of train.py:199-204 kept inside the kernels (row_idx) as in the single-rank step.
    Driven with a RECORDING communicator as rank 1 of 2 (what one GPU allows: the data stays the sum over one rank): two SGD steps give BITWISE the parameters, Adam slots and
    loss scalars of the sequence ppo.py issued from Python before (forward_backward on host-gathered rows -> all-reduce -> apply_adam), and the log is one in-stream all-reduce
    of the whole gradient buffer per step."""
    import ctypes
    from mi355 import lib as milib
    L = milib.get()
    T = 512
    rng = np.random.RandomState(M)
    s = (0.5 * rng.standard_normal((T, 67))).astype(np.float32)
    a = rng.uniform(-1, 1, (T, 2)).astype(np.float32)
    R, A = rng.randn(T).astype(np.float32), rng.randn(T).astype(np.float32)
    rows = [rng.permutation(T)[:M].astype(np.int32) for _ in range(2)]
    res = []
    for mode in ("c_call", "host"):
        m = _ppo_pair(tmp_path, mode)
        d = m.dev
        d.ensure_batch(T)
        sd, ad, Rd, Ad = (m._to_dev(x, x.shape) for x in (s, a, R, A))
        log = np.zeros((8, 4), np.int64)
        h = ctypes.c_void_p()
        L.mi_comm_init_recording(ctypes.addressof(h), 1, 2, log.ctypes.data, 8)
        for step in range(2):
            rd = torch.from_numpy(rows[step]).to(d.device)
            mb = rd.long()
            alpha = 1e-4 * (1.0 + step)
            if mode == "c_call" and gather:
                d.train_step_dp(h, sd, ad, Rd, Ad, None, rd, M, 0.5 / M, 0.5, alpha)
            elif mode == "c_call":
                d.train_step_dp(h, sd[mb].contiguous(), ad[mb].contiguous(), Rd[mb].contiguous(), Ad[mb].contiguous(), None, None, M, 0.5 / M, 0.5, alpha)
            else:
                d.forward_backward(sd[mb].contiguous(), ad[mb].contiguous(), Rd[mb].contiguous(), Ad[mb].contiguous(), M, 0.5 / M, 0.5)
                d.apply_adam(alpha)
        torch.cuda.synchronize()
        n = L.mi_comm_recorded(h)
        L.mi_comm_destroy(h)
        res.append((d.params.cpu().numpy().copy(), d.adam_m.cpu().numpy().copy(), d.adam_v.cpu().numpy().copy(), d.losses.cpu().numpy()[:5].copy(), log[:max(n, 0)].copy(),
                    d.grads.data_ptr(), d.grads.numel()))
    (p_c, m_c, v_c, l_c, log_c, gptr, gn), (p_h, m_h, v_h, l_h, log_h, _, _) = res
    assert len(log_h) == 0
    assert [tuple(int(x) for x in e) for e in log_c] == [(1, gn, 0, gptr)] * 2
    assert np.array_equal(p_c, p_h) and np.array_equal(m_c, m_h) and np.array_equal(v_c, v_h), float(np.abs(p_c - p_h).max())
    assert np.array_equal(l_c, l_h), (l_c, l_h)
    assert not np.array_equal(p_c, _ppo_pair(tmp_path, "fresh").dev.params.cpu().numpy())      # (the steps did something)


def test_ppo_dp_step_through_the_real_rccl_at_one_rank(tmp_path):
    """mi_ppo_train_step_dp with a REAL communicator (RCCL bound at run time, world size 1): the all-reduce is the identity, so three steps through it give bitwise the
    parameters of the recording communicator, and -- inv_m = 1 / M, grad_scale = 1 -- those of the single-rank fused step (mi_ppo_train_step_idx: Adam inside the
    filter-gradient launch instead of a flat launch behind the all-reduce) to rounding."""
    import ctypes
    from mi355 import lib as milib
    L = milib.get()
    T, M = 256, 64
    rng = np.random.RandomState(11)
    s = (0.5 * rng.standard_normal((T, 67))).astype(np.float32)
    a = rng.uniform(-1, 1, (T, 2)).astype(np.float32)
    R, A = rng.randn(T).astype(np.float32), rng.randn(T).astype(np.float32)
    rows = [rng.permutation(T)[:M].astype(np.int32) for _ in range(3)]
    res = {}
    for mode in ("rccl", "recording", "single"):
        m = _ppo_pair(tmp_path, mode)
        d = m.dev
        d.ensure_batch(T)
        sd, ad, Rd, Ad = (m._to_dev(x, x.shape) for x in (s, a, R, A))
        h = ctypes.c_void_p()
        log = np.zeros((8, 4), np.int64)
        if mode == "rccl":
            idb = np.zeros(128, np.uint8)
            L.mi_comm_unique_id(idb.ctypes.data)
            L.mi_comm_init(ctypes.addressof(h), 0, 1, idb.ctypes.data)
        elif mode == "recording":
            L.mi_comm_init_recording(ctypes.addressof(h), 0, 1, log.ctypes.data, 8)
        try:
            for step in range(3):
                rd = torch.from_numpy(rows[step]).to(d.device)
                if mode == "single":
                    d.train_step_idx(sd, ad, Rd, Ad, None, rd, M, 1.0 / M, 1.0, 1e-4)
                else:
                    d.train_step_dp(h, sd, ad, Rd, Ad, None, rd, M, 1.0 / M, 1.0, 1e-4)
            torch.cuda.synchronize()
            res[mode] = (d.params.cpu().numpy().copy(), d.losses.cpu().numpy()[:5].copy())
        finally:
            if mode != "single":
                L.mi_comm_destroy(h)
    assert np.array_equal(res["rccl"][0], res["recording"][0]) and np.array_equal(res["rccl"][1], res["recording"][1])
    assert rel_err(res["rccl"][0], res["single"][0]) < 1e-6 and np.allclose(res["rccl"][1], res["single"][1], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_mlp_vae_dp_step_as_one_c_call_equals_the_host_loop(tmp_path, precision):
    """Round 6: mi_mlpvae_train_step_dp -- forward, the decoder half of the backward pass, its all-reduce queued on the communicator under the encoder
    half, that half's all-reduce, the join, Adam -- as ONE C call.  RECORDING communicator (rank 0 of 2), two SGD steps through an index vector: parameters, both Adam slots
    and losses bitwise those of the host-sequenced loop vae/models.py ran for the MlpVAE (forward, backward(part 1), backward(part 2), Adam); the log is the two buckets of
    mi_mlpvae_dp_buckets as async all-reduces + one join per step; and the same call through the real RCCL at one rank gives the same bits."""
    import ctypes
    from mi355 import lib as milib
    from vae.models import adam_alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON
    L = milib.get()
    src_shape, enc, dec, B = (80, 160, 3), (512, 256), (256, 512), 37
    rng = np.random.RandomState(4)
    table = (rng.randint(0, 256, (50,) + src_shape).astype(np.float32) / 255.0)
    eps = rng.standard_normal((2, B, 64)).astype(np.float32)
    idx = rng.permutation(50)[:B].astype(np.int32)
    params = _mlp_params(9, src_shape, src_shape, enc, dec)
    res = {}
    for mode in ("c_call", "host_loop", "rccl"):
        m = MlpVAE(np.array(src_shape), z_dim=64, model_dir=str(tmp_path / (precision + mode)), precision=precision, learning_rate=1e-4)
        m.set_weights(params)
        m.init_session(init_logging=False)
        d = m.dev
        s_dev, i_dev = m._frames(table, 38400, "src"), torch.from_numpy(idx).cuda()
        log = np.zeros((16, 4), np.int64)
        h = ctypes.c_void_p()
        if mode == "rccl":
            idb = np.zeros(128, np.uint8)
            L.mi_comm_unique_id(idb.ctypes.data)
            L.mi_comm_init(ctypes.addressof(h), 0, 1, idb.ctypes.data)
        else:
            L.mi_comm_init_recording(ctypes.addressof(h), 0, 2, log.ctypes.data, 16)
        try:
            b1p, b2p = np.float32(ADAM_BETA1), np.float32(ADAM_BETA2)
            for s in range(2):
                e_dev = m._eps(B, eps[s])
                alpha = adam_alpha(1e-4, b1p, b2p)
                if mode == "host_loop":
                    d.forward(s_dev, s_dev, i_dev, B, 0.5 / B, e_dev, 1, 1)
                    for part, lo, hi in d.grad_buckets:
                        d.backward(s_dev, i_dev, e_dev, 0.5 / B, part=part)
                    d.apply_adam(alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON)
                else:
                    d.train_step_dp(h, s_dev, s_dev, i_dev, B, 0.5 / B, e_dev, alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON)
                b1p, b2p = np.float32(b1p * np.float32(ADAM_BETA1)), np.float32(b2p * np.float32(ADAM_BETA2))
            torch.cuda.synchronize()
            n = L.mi_comm_recorded(h)
            bk = np.zeros(6, np.int64)
            L.mi_mlpvae_dp_buckets(d.handle, bk.ctypes.data)                  # the library's own table = the host mirror's
            assert [tuple(int(x) for x in bk[3 * i:3 * i + 3]) for i in range(2)] == [tuple(b) for b in d.grad_buckets]
            res[mode] = (d.params.clone(), d.adam_m.clone(), d.adam_v.clone(), d.losses.cpu().numpy().copy(), log[:max(n, 0)].copy(), d.grads.data_ptr(), list(d.grad_buckets))
        finally:
            L.mi_comm_destroy(h)
    p_c, m_c, v_c, l_c, log_c, gptr, buckets = res["c_call"]
    p_h, m_h, v_h, l_h, log_h, _, _ = res["host_loop"]
    want = [(1, hi - lo, 1, gptr + 4 * lo) for (_, lo, hi) in buckets] + [(5, 2, 0, 0)]
    assert len(log_h) == 0 and [tuple(int(x) for x in e) for e in log_c] == want + want
    assert [int(p_) for p_, _, _ in buckets] == [1, 2]
    for x, y in ((p_c, p_h), (m_c, m_h), (v_c, v_h)):
        assert torch.equal(x, y)
    assert np.array_equal(l_c, l_h)
    assert torch.equal(res["rccl"][0], p_c) and np.array_equal(res["rccl"][3], l_c)
