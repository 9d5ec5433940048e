"""Synthetic replay (BASELINE configs[4], SURVEY 8c "C5"): many independent trajectories through the whole path at once --
VAE encode of every frame, value estimates, GAE + per-trajectory advantage normalisation, PPO minibatch SGD -- device resident.

It is the reference's per-horizon update (train.py:171-207: `encode_state` per step vae_common.py:45-59, `compute_gae`, returns /
normalised advantages, `update_old_policy`, `num_epochs` x shuffled minibatches of `model.train`) applied to R recorded trajectories of
T steps instead of one live rollout:

    states[r, t]   = concat(vae.encode(frame[r, t]), measurements[r, t])         (vae_common.py:48,59; mean of the posterior)
    values[r, t]   = V(states[r, t]),  t = 0..T  (the last one is the bootstrap value, train.py:172)
    A[r, :]        = compute_gae(rewards[r], values[r, :T], values[r, T], dones[r], gamma, lam)        per row, fp64, bit-exact
    returns[r, :]  = A + values ;  A <- (A - mean_r) / (std_r + 1e-8)                                   per row (train.py:176-177)
    theta_old <- theta ; for each epoch: legacy-numpy shuffle of the R*T samples, minibatches of batch_size (last one partial)

Data parallel: trajectories are independent, so each rank owns the rows [lo, hi) of shard_bounds(R); encode, values, GAE and
the normalisation need no exchange; every minibatch step sums the flat gradient buffer over the ranks (PPO._step_resident), each rank
contributing batch_size / world of ITS OWN samples (so the sample order differs from a single-process run of the same data: fp32
parity across GPU counts is 1e-4 on the losses, not bit-exact).  No CPU fallback: needs the HIP library and a GPU.
"""
import numpy as np

from mi355 import dist as midist


def encode_resident(vae, frames, chunk=512):
    """frames: uint8 or float [N, H, W, C] / [N, H*W*C] host array (uint8 is normalised on the device, bit-exact with /255), or the same table as a
    device-resident torch tensor (uint8 / float32, contiguous: read where it lies) -> device tensor [N, z_dim] of posterior means, `chunk` frames per launch."""
    import torch
    dev = vae._need_dev()
    n = len(frames)
    feat = vae._src_feat()
    out = torch.empty(n, int(vae.z_dim), device=dev.device)
    resident = torch.is_tensor(frames) and frames.is_cuda          # a frame table that already lives in HBM (recorded there, or uploaded once by the caller): no PCIe in this call
    if resident:
        if frames.dtype == torch.uint8 and not getattr(dev, "accepts_u8", False):
            raise ValueError("encode_resident: a device-resident uint8 frame table needs the bf16 engine (the fp32 / bf16x3 engines read float frames)")
        if frames.dtype not in (torch.uint8, torch.float32):
            raise ValueError("encode_resident: device-resident frames must be uint8 camera bytes or float32 in [0, 1]")
        table = frames.reshape(n, -1)
        if table.shape[1] != feat or not table.is_contiguous():
            raise ValueError("encode_resident: expected a contiguous table of %d values per frame, got shape %s" % (feat, tuple(frames.shape)))
        if table.dtype == torch.float32 and not dev.range_ok(table):
            raise ValueError("verify_range: device-resident frames outside [0, 1]")
    for lo in range(0, n, chunk):
        hi = min(lo + chunk, n)
        src = table[lo:hi] if resident else vae._frames(frames[lo:hi], feat, "frames", keep_u8_ok=True)      # uint8 stays uint8 in HBM on the bf16 engine (conv1 normalises in registers)
        dev.encode(src, None, hi - lo, out[lo:hi])
    return out


def replay_update(vae, ppo, frames, measurements, actions, rewards, dones, gamma=0.99, lam=0.95, num_epochs=3, batch_size=32, encode_chunk=512,
                  local_rows=False, stage_times=None, return_z=False):
    """One PPO update over R recorded trajectories (see the module docstring).

    frames [R, T+1, H, W, C] (uint8 or float in [0,1], host array or device-resident torch tensor; the last frame of a row is the state after its last step), measurements [R, T+1, k],
    actions [R, T, A] (the actions that were taken), rewards [R, T], dones [R, T].  batch_size is the GLOBAL minibatch size.
    local_rows=True: the arrays already hold only this rank's trajectories (each rank loaded / generated its own shard; every rank must
    hold the same number of them, so that all ranks run the same number of SGD steps).
    stage_times: optional dict that receives the wall time (seconds, device synchronised at the stage boundaries) of "encode" (upload + VAE
    encode), "values", "gae" and "sgd".
    Returns a dict: per-minibatch loss records (this rank's device scalars, read back once at the end), and this rank's returns /
    advantages / values (fp64 / fp64 / fp32 numpy, read back once after the SGD loop) for inspection; return_z=True adds the latents
    (R*(T+1) x z_dim floats: 34 MB at 1024 x 128 -- only on request)."""
    import torch
    import utils

    import torch as _torch
    if not (_torch.is_tensor(frames) and frames.is_cuda):      # (device-resident frame tables are read where they lie: encode_resident)
        frames = np.asarray(frames)
    measurements = np.asarray(measurements, np.float32)
    actions, rewards, dones = np.asarray(actions, np.float32), np.asarray(rewards, np.float64), np.asarray(dones, np.float64)
    R, T = rewards.shape
    if tuple(frames.shape[:2]) != (R, T + 1) or measurements.shape[:2] != (R, T + 1) or actions.shape[:2] != (R, T) or dones.shape != (R, T):
        raise ValueError("replay_update: frames / measurements [R, T+1, ...], actions [R, T, A], rewards / dones [R, T]")
    world, rank = midist.world_size(), midist.rank()
    if world > 1 and not local_rows and R % world != 0:
        raise ValueError("replay_update: %d trajectories do not split evenly over %d ranks (every rank must run the same number of SGD steps)" % (R, world))
    if world > 1 and (batch_size < world or batch_size % world != 0):
        # unequal shares would give the ranks different numbers of SGD steps (= different numbers of gradient all-reduces: a hang)
        raise ValueError("replay_update: the global minibatch size %d must be a positive multiple of the %d ranks" % (batch_size, world))
    lo, hi = (0, R) if local_rows else midist.shard_bounds(R, rank, world)
    r_loc = hi - lo
    pdev = ppo._need_dev()
    device = pdev.device

    import time

    def mark(name, t0):
        if stage_times is not None:
            torch.cuda.synchronize(device)
            stage_times[name] = stage_times.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()
    t_stage = time.perf_counter()
    # 1. states of this rank's rows: encode every frame, append the measurements
    z = encode_resident(vae, frames[lo:hi].reshape((r_loc * (T + 1),) + frames.shape[2:]), encode_chunk)
    meas = torch.from_numpy(np.ascontiguousarray(measurements[lo:hi].reshape(r_loc * (T + 1), -1))).to(device)
    states_all = torch.cat([z, meas], dim=1).contiguous()                       # [r_loc * (T+1), input_dim]
    if states_all.shape[1] != ppo.input_dim:
        raise ValueError("replay_update: z_dim + measurements = %d but the policy takes %d inputs" % (states_all.shape[1], ppo.input_dim))

    t_stage = mark("encode", t_stage)
    # 2. value estimates of every state (greedy predict: no noise drawn; the action output is not used)
    n_all = states_all.shape[0]
    values_all = torch.empty(n_all, device=device)
    scratch_act = torch.empty(min(n_all, 4096), ppo.num_actions, device=device)
    for a in range(0, n_all, 4096):
        b = min(a + 4096, n_all)
        pdev.predict(states_all[a:b], b - a, None, True, scratch_act[:b - a], values_all[a:b])
    values = values_all.view(r_loc, T + 1)

    t_stage = mark("values", t_stage)
    # 3. GAE + returns + per-row normalisation (fp64 on the device, bit-exact with numpy / scipy)
    # (the value estimates never leave the device: widened to fp64 there, exactly)
    _, returns_d, adv_d = utils.gae_resident(rewards[lo:hi], values, dones[lo:hi], gamma, lam)

    t_stage = mark("gae", t_stage)
    # 4. minibatch SGD on the flattened samples of this rank
    s = states_all.view(r_loc, T + 1, -1)[:, :T].reshape(r_loc * T, -1).contiguous()
    a = torch.from_numpy(np.ascontiguousarray(actions[lo:hi].reshape(r_loc * T, -1))).to(device)
    ret = returns_d.reshape(-1).to(torch.float32)                                # f64 -> f32 at the feed (ppo.py:108-109), round-to-nearest-even as numpy's astype
    adv_t = adv_d.reshape(-1).to(torch.float32)
    n_loc = r_loc * T
    mb_lo, mb_hi = midist.shard_bounds(batch_size, rank, world)
    mb_loc = mb_hi - mb_lo                                                       # this rank's share of a full global minibatch
    ppo.update_old_policy()
    # theta_old is now fixed for the whole update: log pi_old(a | s) of every sample is computed ONCE (the reference's graph recomputes the old
    # policy's forward pass in every minibatch step, ppo.py:112-121 -- same numbers)
    logp_old = None
    if hasattr(pdev, "logp_old") and pdev.fused_ok():      # (data parallel too: theta_old is replicated, every rank caches the values of ITS samples)
        logp_old = torch.empty(n_loc, device=device)
        for lo_ in range(0, n_loc, 4096):
            hi_ = min(lo_ + 4096, n_loc)
            pdev.logp_old(s[lo_:hi_], a[lo_:hi_], hi_ - lo_, logp_old[lo_:hi_])
    records = []
    # the minibatch gather inside the step's kernels: single rank, and (round 6) data parallel -- PPO._step_rows keeps it in the one-call step mi_ppo_train_step_dp when the
    # library's communicator carries the all-reduce, and gathers the rows itself otherwise (torch.distributed as the transport)
    fused_rows = hasattr(ppo, "_step_rows") and hasattr(pdev, "train_step_idx") and pdev.fused_ok()
    for _ in range(num_epochs):
        indices = np.arange(n_loc)
        np.random.shuffle(indices)                                               # legacy numpy RNG, as train.py:194-195
        perm = torch.from_numpy(indices.astype(np.int32) if fused_rows else indices).to(device)
        n_steps = int(np.ceil(n_loc / mb_loc)) if mb_loc > 0 else 0
        for i in range(n_steps):
            mb = perm[i * mb_loc:(i + 1) * mb_loc]                               # the last one may be partial (train.py:199-201)
            m_local = int(mb.numel())
            m_global = m_local * world if world > 1 else m_local                 # ranks hold equal shares (R divisible by world is the C5 layout)
            if fused_rows:                                                       # the minibatch gather happens inside the step's kernels (a slice of the permutation is the row index)
                ppo._step_rows(s, a, ret, adv_t, logp_old, mb, m_local, m_global)
            elif logp_old is not None:
                ppo._step_resident(s[mb].contiguous(), a[mb].contiguous(), ret[mb].contiguous(), adv_t[mb].contiguous(), m_local, m_global, logp_old=logp_old[mb].contiguous())
            else:
                ppo._step_resident(s[mb].contiguous(), a[mb].contiguous(), ret[mb].contiguous(), adv_t[mb].contiguous(), m_local, m_global)
            ppo.train_step_counter += 1
            records.append(pdev.losses.clone())
    losses = torch.stack(records).cpu().numpy() if records else np.zeros((0, 5), np.float32)
    mark("sgd", t_stage)
    keys = ("policy_loss", "value_loss", "entropy_loss", "loss", "prob_ratio")
    out = {"losses": [dict(zip(keys, (float(x) for x in row))) for row in losses], "returns": returns_d.cpu().numpy(), "advantages": adv_d.cpu().numpy(),
           "values": values.cpu().numpy(), "rows": (lo, hi), "samples_per_rank": n_loc}
    if return_z:
        out["z"] = z.cpu().numpy()
    return out
