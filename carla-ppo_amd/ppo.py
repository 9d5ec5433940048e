"""ppo — drop-in for the reference's ppo.py (PPO class surface) on MI355X through libmi355_carla.so.

Reference symbols mirrored (file:line in the reference checkout):
    PolicyGraph            ppo.py:11-66     (fused into the native engine: two 67-500-300 MLPs, Gaussian head)
    PPO.__init__           ppo.py:73-190    init_session :192-200   save :202-205   load_latest_checkpoint :207-216
    train                  ppo.py:218-229   (alias: learn; explicit single step: train_step)
    predict                ppo.py:231-251   get_*_idx :253-260      write_*_summary :262-273   update_old_policy :275-276

Host code is plumbing (numpy in / numpy out, torch tensors as HBM handles).  All arithmetic runs in HIP kernels;
there is no CPU fallback — init_session() raises without a GPU or without the built library.
"""
import os

import numpy as np

from mi355 import checkpoint as ckpt
from mi355 import dist as midist
from mi355 import lib as milib
from mi355.init import init_ppo, ppo_variables

ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON = 0.9, 0.999, 1e-8


def _adam_alpha(lr, b1p, b2p):
    one = np.float32(1.0)
    return np.float32(np.float32(lr) * np.sqrt(one - np.float32(b2p), dtype=np.float32) / (one - np.float32(b1p)))


class PPO():
    """Proximal policy optimisation model (reference ppo.py:68-276)."""

    def __init__(self, input_shape, action_space,
                 learning_rate=3e-4, lr_decay=0.998, epsilon=0.2,
                 value_scale=0.5, entropy_scale=0.01, initial_std=0.4,
                 model_dir="./", seed=None):
        self.input_dim = int(np.asarray(input_shape).reshape(-1)[0])
        self.num_actions = int(action_space.shape[0])
        self.action_low = np.asarray(action_space.low, np.float32).reshape(-1)
        self.action_high = np.asarray(action_space.high, np.float32).reshape(-1)
        self.learning_rate_value, self.lr_decay = learning_rate, lr_decay
        self.epsilon, self.value_scale, self.entropy_scale = epsilon, value_scale, entropy_scale
        self.initial_std, self.seed = initial_std, seed
        self._variables = ppo_variables(self.input_dim, self.num_actions)
        self._init_values = None

        # counters (ppo.py:95-98)
        self.train_step_counter = 0
        self.predict_step_counter = 0
        self.episode_counter = 0
        self.beta1_power, self.beta2_power = np.float32(ADAM_BETA1), np.float32(ADAM_BETA2)
        self._metric_sums, self._metric_n = {}, 0
        self.dev = None
        self.sess = None
        self.train_writer = None
        self._noise_offset = 0                                 # N(0,1) values drawn so far from this agent's Philox stream (predict)

        # Setup model saver and dirs (ppo.py:183-190)
        self.model_dir = model_dir
        self.checkpoint_dir = "{}/checkpoints/".format(self.model_dir)
        self.log_dir = "{}/logs/".format(self.model_dir)
        self.video_dir = "{}/videos/".format(self.model_dir)
        self.dirs = [self.checkpoint_dir, self.log_dir, self.video_dir]
        for d in self.dirs:
            os.makedirs(d, exist_ok=True)

    # ------------------------------------------------------------------ session / engine
    def init_session(self, sess=None, init_logging=True):
        from mi355.ppo_device import PpoDevice
        self.sess = sess if sess is not None else self
        if self.seed is None:                                  # train.py's np.random.seed(--seed) drives initialisation and sampling (mi355.init.seed_from_numpy_state)
            from mi355.init import seed_from_numpy_state
            self.seed = seed_from_numpy_state()
        self.dev = PpoDevice(self.input_dim, self.num_actions, self.action_low, self.action_high,
                             self.epsilon, self.value_scale, self.entropy_scale)
        values = self._init_values or init_ppo(self.seed, self.input_dim, self.num_actions, self.initial_std)
        # policy_old is a separately initialised copy in the reference; the trainer overwrites it with
        # update_old_policy() before the first train() (train.py:192), so it starts as a copy of policy.
        self.dev.load_params(values, {k.replace("policy/", "policy_old/", 1): v for k, v in values.items()})
        if midist.world_size() > 1:
            midist.broadcast(self.dev.params, 0)
            midist.broadcast(self.dev.params_old, 0)
        if init_logging:
            from mi355.summary import SummaryWriter
            self.train_writer = SummaryWriter(self.log_dir)

    def _need_dev(self):
        if self.dev is None:
            raise RuntimeError("call init_session() first")
        return self.dev

    def set_weights(self, named):
        if self.dev is None:
            self._init_values = {k: np.asarray(v, np.float32) for k, v in named.items()}
        else:
            self.dev.load_params(named)

    # ------------------------------------------------------------------ checkpoints
    def state_dict(self):
        dev = self._need_dev()
        out = dict(dev.export_params())
        out.update(dev.export_old())
        m, v = dev.export_slots()
        for k in m:
            out[k + "/Adam"] = m[k]
            out[k + "/Adam_1"] = v[k]
        out["beta1_power"], out["beta2_power"] = np.float32(self.beta1_power), np.float32(self.beta2_power)
        out["train_step_counter"] = np.int32(self.train_step_counter)
        out["predict_step_counter"] = np.int32(self.predict_step_counter)
        out["episode_counter"] = np.int32(self.episode_counter)
        return out

    def load_state_dict(self, sd):
        dev = self._need_dev()
        old = {k.replace("policy/", "policy_old/", 1): sd.get(k.replace("policy/", "policy_old/", 1), sd[k]) for k in self._variables}
        dev.load_params({k: sd[k] for k in self._variables}, old)
        if all((k + "/Adam") in sd for k in self._variables):
            dev.load_slots({k: sd[k + "/Adam"] for k in self._variables}, {k: sd[k + "/Adam_1"] for k in self._variables})
            self.beta1_power = np.float32(sd.get("beta1_power", ADAM_BETA1))
            self.beta2_power = np.float32(sd.get("beta2_power", ADAM_BETA2))
        self.train_step_counter = int(sd.get("train_step_counter", 0))
        self.predict_step_counter = int(sd.get("predict_step_counter", 0))
        self.episode_counter = int(sd.get("episode_counter", 0))

    def save(self):
        if midist.rank() == 0:
            model_checkpoint = ckpt.save(self.checkpoint_dir, self.episode_counter, self.state_dict())
            print("Model checkpoint saved to {}".format(model_checkpoint))

    def load_latest_checkpoint(self):
        model_checkpoint = ckpt.latest(self.checkpoint_dir)
        if model_checkpoint:
            try:
                self.load_state_dict(ckpt.load(model_checkpoint))
                print("Model checkpoint restored from {}".format(model_checkpoint))
                return True
            except Exception as e:
                print(e)
                return False

    # ------------------------------------------------------------------ training
    def current_learning_rate(self):
        """tf.train.exponential_decay(lr, episode_counter, 1, lr_decay, staircase=True) in fp32 (ppo.py:142)."""
        return np.float32(np.float32(self.learning_rate_value) * np.power(np.float32(self.lr_decay), np.float32(self.episode_counter)))

    def _to_dev(self, a, shape):
        import torch
        a = np.ascontiguousarray(np.asarray(a, np.float32).reshape(shape))      # f64 -> f32 at the feed (ppo.py:108-109)
        return torch.from_numpy(a).to(self.dev.device)

    def _step_resident(self, s, a, r, adv, m_local, m_global, logp_old=None):
        """One SGD step on device-resident minibatch tensors.  Single rank: one C call (fused forward / losses / backward / Adam, five launches);
        data parallel: gradients, all-reduce, Adam.  logp_old: cached log pi_old(a|s) of these samples (PpoDevice.logp_old), optional."""
        dev = self.dev
        alpha = _adam_alpha(self.current_learning_rate(), self.beta1_power, self.beta2_power)
        comm = self._dp_comm()
        if midist.world_size() == 1 and os.environ.get("MI355_PPO_FUSED", "1") != "0":
            dev.train_step(s, a, r, adv, m_local, 1.0 / m_global, m_local / float(m_global), alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON, logp_old=logp_old)
        elif comm is not None:
            # data parallel with the library's communicator live: the step is ONE C call -- the fused chain, the all-reduce of the flat gradient buffer, Adam (round 6)
            dev.train_step_dp(comm.handle, s, a, r, adv, logp_old, None, m_local, 1.0 / m_global, m_local / float(m_global), alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON)
        else:
            dev.forward_backward(s, a, r, adv, m_local, 1.0 / m_global, m_local / float(m_global))
            if midist.world_size() > 1:
                midist.all_reduce_sum(dev.grads)
            dev.apply_adam(alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON)
        self.beta1_power = np.float32(self.beta1_power * np.float32(ADAM_BETA1))
        self.beta2_power = np.float32(self.beta2_power * np.float32(ADAM_BETA2))

    def _dp_comm(self):
        """The C-ABI communicator that carries this process's gradient all-reduce inside the one-call data-parallel step (mi_ppo_train_step_dp), or None: single
        rank, torch.distributed as the transport (gloo in the CPU tests, a rank without a usable RCCL), or MI355_DP_HOST_LOOP=1 (the host-sequenced form, A/B runs).
        MI355_DP_SKIP_ALLREDUCE=1 (bench.py only): the recording communicator -- the same C call with its collective recorded instead of issued."""
        if midist.world_size() == 1 or os.environ.get("MI355_DP_HOST_LOOP") == "1":
            return None
        comm = midist.mi_comm()
        if comm is not None and os.environ.get("MI355_DP_SKIP_ALLREDUCE") == "1":
            comm = midist.recording_comm()
        return comm

    def _step_rows(self, s_all, a_all, r_all, adv_all, logp_old_all, rows, m_local, m_global):
        """One SGD step on rows `rows` (int32 device tensor) of device-resident horizon-batch tables: the minibatch gather of train.py:199-204 runs inside
        the step's kernels (single rank, fused kernels); otherwise the rows are gathered here and _step_resident takes over."""
        dev = self.dev
        # (shapes outside the fused kernels' range -- e.g. more than 8 actions -- have no in-kernel gather: mi_ppo_fused_shape_ok says so up front and the rows
        #  are gathered here instead; row values themselves are clamped into the tables by the kernels)
        if midist.world_size() == 1 and os.environ.get("MI355_PPO_FUSED", "1") != "0" and os.environ.get("MI355_PPO_IDX", "1") != "0" and dev.fused_ok():
            alpha = _adam_alpha(self.current_learning_rate(), self.beta1_power, self.beta2_power)
            dev.train_step_idx(s_all, a_all, r_all, adv_all, logp_old_all, rows, m_local, 1.0 / m_global, m_local / float(m_global), alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON)
            self.beta1_power = np.float32(self.beta1_power * np.float32(ADAM_BETA1))
            self.beta2_power = np.float32(self.beta2_power * np.float32(ADAM_BETA2))
            return
        comm = self._dp_comm()
        if comm is not None and os.environ.get("MI355_PPO_FUSED", "1") != "0" and os.environ.get("MI355_PPO_IDX", "1") != "0" and dev.fused_ok():
            # data parallel: the in-kernel gather is kept -- `rows` index THIS rank's tables -- and the step stays one C call (round 6; before, world_size > 1 gathered here
            # and sequenced forward_backward -> all-reduce -> Adam from Python)
            alpha = _adam_alpha(self.current_learning_rate(), self.beta1_power, self.beta2_power)
            dev.train_step_dp(comm.handle, s_all, a_all, r_all, adv_all, logp_old_all, rows, m_local, 1.0 / m_global, m_local / float(m_global), alpha, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON)
            self.beta1_power = np.float32(self.beta1_power * np.float32(ADAM_BETA1))
            self.beta2_power = np.float32(self.beta2_power * np.float32(ADAM_BETA2))
            return
        mb = rows.to("cuda" if rows.is_cuda else rows.device).long()
        self._step_resident(s_all[mb].contiguous(), a_all[mb].contiguous(), r_all[mb].contiguous(), adv_all[mb].contiguous(), m_local, m_global,
                            logp_old=None if logp_old_all is None else logp_old_all[mb].contiguous())

    def train(self, input_states, taken_actions, returns, advantage):
        """One SGD step on a minibatch + metric update + train_step_counter += 1; returns None (ppo.py:218-229)."""
        dev = self._need_dev()
        m = len(input_states)
        s = self._to_dev(input_states, (m, self.input_dim))
        a = self._to_dev(taken_actions, (m, self.num_actions))
        r = self._to_dev(returns, (m,))
        adv = self._to_dev(advantage, (m,))
        # data parallel (torch.distributed initialised): every rank passes ITS rows of the global minibatch, equal counts on all ranks;
        # gradients are those of sum_local / M_global and the state-independent entropy term is shared through grad_scale = 1 / world,
        # so that the all-reduced buffer is the gradient of the global mean (mi355/dist.py).  Single process: m_global = m.
        self._step_resident(s, a, r, adv, m, m * midist.world_size())
        if self.train_writer is not None:                      # episodic means (ppo.py:150-163); never on a timed path
            # NO collective here: whether a rank logs is that rank's own business (rank 0 only, usually), and an all-reduce issued by the
            # logging ranks alone would pair with the other ranks' next gradient all-reduce.  A data-parallel rank logs ITS rows' means.
            L = self._local_losses()
            for k, v in zip(("train_loss/policy", "train_loss/value", "train_loss/entropy", "train_loss/loss", "train/prob_ratio"), L):
                self._metric_sums[k] = self._metric_sums.get(k, 0.0) + float(v)
            ta = np.asarray(taken_actions, np.float64).reshape(m, self.num_actions)
            for i in range(self.num_actions):                  # ppo.py:155-158: minibatch means of the taken action, the policy mean and its std
                for tag, val in (("taken_actions", ta[:, i].mean()), ("mean", L[5 + i]), ("std", L[5 + self.num_actions + i])):
                    k = "train_actor/action_{}/{}".format(i, tag)
                    self._metric_sums[k] = self._metric_sums.get(k, 0.0) + float(val)
            self._metric_sums["train/returns"] = self._metric_sums.get("train/returns", 0.0) + float(np.mean(returns))
            self._metric_sums["train/advantage"] = self._metric_sums.get("train/advantage", 0.0) + float(np.mean(advantage))
            self._metric_sums["train/learning_rate"] = self._metric_sums.get("train/learning_rate", 0.0) + float(self.current_learning_rate())
            self._metric_n += 1
        self.train_step_counter += 1

    def _local_losses(self):
        """Loss scalars + per-action mean / std of the last step from THIS rank's rows, no communication.  The device holds sums over the local
        rows / M_global (so that the gradient all-reduce yields the global mean): times the world size that is the mean over the local rows.
        The entropy term and std = exp(logstd) are state independent -- identical on every rank, left as they are."""
        L = self.dev.losses.clone().cpu().numpy()
        w = midist.world_size()
        if w > 1:
            A = self.num_actions
            L[[0, 1, 4]] *= w
            L[5:5 + A] *= w
            L[3] = -L[0] + L[1] - L[2]
        return L

    def _global_losses(self):
        """The five loss scalars of the last step as numpy.  Data parallel: the device holds this rank's share of the batch means (sums over
        its rows / M_global) for the policy / value terms and the probability ratio; they are summed over the ranks here, the
        state-independent entropy term is already the global value."""
        L = self.dev.losses.clone()
        if midist.world_size() > 1:                            # COLLECTIVE: every rank must call this (train_step() does; train()'s logging does not)
            A = self.num_actions
            keep = L[[2] + list(range(5 + A, 5 + 2 * A))].clone()   # entropy term and std = exp(logstd): identical on every rank, not sums
            midist.all_reduce_sum(L)
            L[[2] + list(range(5 + A, 5 + 2 * A))] = keep
            L[3] = -L[0] + L[1] - L[2]
        return L.cpu().numpy()

    learn = train                                              # north-star alias

    def train_step(self, input_states, taken_actions, returns, advantage):
        """train() that also returns the five loss scalars {policy, value, entropy, loss, prob_ratio} (for parity tests/benchmarks)."""
        self.train(input_states, taken_actions, returns, advantage)
        L = self._global_losses()
        return dict(policy_loss=float(L[0]), value_loss=float(L[1]), entropy_loss=float(L[2]), loss=float(L[3]), prob_ratio=float(L[4]))

    def update_old_policy(self):
        self._need_dev().update_old()

    # ------------------------------------------------------------------ inference
    def predict(self, input_states, greedy=False, write_to_summary=False, noise=None):
        """Returns (action, value); squeezes the batch dim when it is 1 (ppo.py:231-251).  `noise` injects the N(0,1) draw."""
        import torch
        dev = self._need_dev()
        input_states = np.asarray(input_states)
        if len(input_states.shape) != 2:
            input_states = input_states[None] if input_states.ndim == 1 else [input_states]
        s = self._to_dev(input_states, (len(input_states), self.input_dim))
        m = s.shape[0]
        nz = None
        if not greedy:
            if noise is not None:
                nz = self._to_dev(noise, (m, self.num_actions))
            else:
                # exploration noise from the library's own Philox4x32-10 + Box-Muller kernel (mi_normal_philox; the same generator the VAE engine
                # samples with): no torch kernel on the rollout path.  Stream = (seed, rank), offset = values drawn so far.
                nz = torch.empty(m, self.num_actions, device=dev.device)
                n = m * self.num_actions
                milib.get().mi_normal_philox(torch.cuda.current_stream(dev.device).cuda_stream,
                                             0xAC7 + 1000003 * int(self.seed or 0) + midist.rank(), self._noise_offset, milib.ptr(nz), n)
                self._noise_offset += n
        action = torch.empty(m, self.num_actions, device=dev.device)
        value = torch.empty(m, device=dev.device)
        dev.predict(s, m, nz, greedy, action, value)
        sampled_action, value = action.cpu().numpy(), value.cpu().numpy()
        if write_to_summary:
            if self.train_writer is not None:                  # ppo.py:175-180: sampled action, policy mean and std of the FIRST state of the batch
                mean0 = dev.action_mean[0].cpu().numpy()
                o, n = dev.layout["policy/action_logstd"]
                std = np.exp(dev.params[o:o + n].cpu().numpy())
                for i in range(self.num_actions):
                    self.train_writer.add_scalar("predict_actor/action_{}/sampled_action".format(i), sampled_action[0, i], self.predict_step_counter)
                    self.train_writer.add_scalar("predict_actor/action_{}/mean".format(i), float(mean0[i]), self.predict_step_counter)
                    self.train_writer.add_scalar("predict_actor/action_{}/std".format(i), float(std[i]), self.predict_step_counter)
            self.predict_step_counter += 1
        if len(input_states) == 1:
            return sampled_action[0], value[0]
        return sampled_action, value

    def get_episode_idx(self):
        return int(self.episode_counter)

    def get_train_step_idx(self):
        return int(self.train_step_counter)

    def get_predict_step_idx(self):
        return int(self.predict_step_counter)

    # ------------------------------------------------------------------ summaries
    def write_value_to_summary(self, summary_name, value, step):
        if self.train_writer is not None:
            self.train_writer.add_scalar(summary_name, value, step)

    def write_dict_to_summary(self, summary_name, params, step):
        if self.train_writer is not None:                      # the reference adds this summary WITHOUT a global step (ppo.py:269): event step 0
            self.train_writer.add_text(summary_name, {k: str(v) for k, v in params.items()}, 0)

    def write_episodic_summaries(self):
        """Writes the episodic means, then increments episode_counter and resets the accumulators (ppo.py:271-273)."""
        if self.train_writer is not None and self._metric_n:
            for k, v in self._metric_sums.items():
                self.train_writer.add_scalar(k, v / self._metric_n, self.get_episode_idx())
            self.train_writer.flush()
        self.episode_counter += 1
        self._metric_sums, self._metric_n = {}, 0
