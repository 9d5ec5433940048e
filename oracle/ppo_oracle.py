"""CPU oracle: PPO policy/value graph, clipped-surrogate loss, gradients, TF-Adam, GAE, minibatch schedule.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parity: the reference has no tests and TF 1.13 cannot run here, but the policy /
value forward pass, sampling + clipping, the three loss terms, all 13 gradients, update_old_policy and the Adam trajectory are pinned to
the shipped agent's serialized graph (models/pretrained_agent `.meta` -> tests/golden/ref_graph_ppo.json.gz, executed by
oracle/tf_graph.py; tests/test_ref_graph.py); GAE / advantage normalisation / minibatch schedule (Python in the reference) by
oracle/gae_ref.c and the KATs in tests/.

Restates (citations relative to /root/reference):
  utils.py:25-28        build_mlp                       -> _mlp()
  utils.py:45-50        compute_gae (scipy lfilter f64) -> compute_gae()
  ppo.py:38-66          PolicyGraph                     -> policy_forward()
  ppo.py:119-132        ratio / clipped surrogate / value / entropy losses -> ppo_losses()
  ppo.py:142-147        Adam(lr*decay^episode), theta_old <- theta         -> OraclePPO
  ppo.py:218-251        train / predict                 -> OraclePPO.train / predict
  train.py:175-177      returns, advantage normalisation (f64, population std) -> returns_and_normalized_advantages()
  train.py:192-207      update_old_policy + epochs x shuffled minibatches (last partial kept) -> ppo_update()

tfp.distributions.Normal formulas (tensorflow_probability, un-vendored third-party dependency, version
unstated by the reference; formulas as recorded in the shipped GraphDef):
  log_prob(x) = -0.5*((x-mu)/sigma)^2 - (0.5*log(2*pi) + log(sigma)),  sigma = exp(logstd)
  entropy     = 0.5 + 0.5*log(2*pi) + log(sigma)
"""
from collections import OrderedDict

import numpy as np
import scipy.signal
import torch

from .vae_oracle import AdamTF, glorot_uniform, _t

HALF_LOG_2PI = 0.5 * np.log(2.0 * np.pi)          # 0.9189385...


def ppo_variable_specs(input_dim=67, num_actions=2, pi_hidden=(500, 300), vf_hidden=(500, 300), scope="policy"):
    """Trainable variables in TF creation order (ppo.py:42-55; names pinned by ref_variables.json)."""
    s = OrderedDict()
    s[scope + "/dense/kernel"] = (input_dim, pi_hidden[0]); s[scope + "/dense/bias"] = (pi_hidden[0],)
    s[scope + "/dense_1/kernel"] = (pi_hidden[0], pi_hidden[1]); s[scope + "/dense_1/bias"] = (pi_hidden[1],)
    s[scope + "/action_mean/kernel"] = (pi_hidden[1], num_actions); s[scope + "/action_mean/bias"] = (num_actions,)
    s[scope + "/action_logstd"] = (num_actions,)
    s[scope + "/dense_2/kernel"] = (input_dim, vf_hidden[0]); s[scope + "/dense_2/bias"] = (vf_hidden[0],)
    s[scope + "/dense_3/kernel"] = (vf_hidden[0], vf_hidden[1]); s[scope + "/dense_3/bias"] = (vf_hidden[1],)
    s[scope + "/value/kernel"] = (vf_hidden[1], 1); s[scope + "/value/bias"] = (1,)
    return s


def variance_scaling_truncnormal(rng, shape, scale):
    """tf.initializers.variance_scaling(scale) defaults: fan_in, truncated normal (ppo.py:45)."""
    stddev = np.sqrt(scale / shape[0]) / 0.87962566103423978
    x = rng.standard_normal(size=shape)
    bad = np.abs(x) > 2.0
    while bad.any():                                      # resample outside +-2 sigma (TF truncated_normal)
        x[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * stddev).astype(np.float32)


def init_ppo_params(seed=0, input_dim=67, num_actions=2, initial_std=0.4, initial_mean_factor=0.1):
    rng = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in ppo_variable_specs(input_dim, num_actions).items():
        if name.endswith("action_mean/kernel"):
            out[name] = variance_scaling_truncnormal(rng, shape, initial_mean_factor)
        elif name.endswith("kernel"):
            out[name] = glorot_uniform(rng, shape)
        elif name.endswith("action_logstd"):
            out[name] = np.full(shape, np.log(initial_std), dtype=np.float32)      # ppo.py:48
        else:
            out[name] = np.zeros(shape, np.float32)
    return out


def policy_forward(p, states, action_low, action_high, scope="policy"):
    """ppo.py:42-55. Returns action_mean [M,A], logstd [A], value [M]."""
    x = states
    h = torch.relu(x @ p[scope + "/dense/kernel"] + p[scope + "/dense/bias"])
    h = torch.relu(h @ p[scope + "/dense_1/kernel"] + p[scope + "/dense_1/bias"])          # output_activation=relu too
    t = torch.tanh(h @ p[scope + "/action_mean/kernel"] + p[scope + "/action_mean/bias"])
    lo, hi = _t(action_low, x.dtype), _t(action_high, x.dtype)
    mean = lo + ((t + 1) / 2) * (hi - lo)                                                   # ppo.py:47
    g = torch.relu(x @ p[scope + "/dense_2/kernel"] + p[scope + "/dense_2/bias"])
    g = torch.relu(g @ p[scope + "/dense_3/kernel"] + p[scope + "/dense_3/bias"])
    value = (g @ p[scope + "/value/kernel"] + p[scope + "/value/bias"]).squeeze(-1)
    return mean, p[scope + "/action_logstd"], value


def normal_log_prob(x, mean, logstd):
    sigma = torch.exp(logstd)
    z = (x - mean) / sigma
    return -0.5 * z * z - (HALF_LOG_2PI + torch.log(sigma))


def ppo_losses(p, p_old, states, actions, returns, advantage, action_low, action_high, epsilon=0.2,
               value_scale=0.5, entropy_scale=0.01):
    """ppo.py:112-132. Returns dict of scalars (torch) incl. total loss."""
    mean, logstd, value = policy_forward(p, states, action_low, action_high, "policy")
    with torch.no_grad():
        mean_o, logstd_o, _ = policy_forward(p_old, states, action_low, action_high, "policy_old")
    logp = normal_log_prob(actions, mean, logstd).sum(dim=-1, keepdim=True)               # [M,1]
    logp_old = normal_log_prob(actions, mean_o, logstd_o).sum(dim=-1, keepdim=True)
    ratio = torch.exp(logp - logp_old)
    adv = advantage.unsqueeze(-1)
    policy_loss = torch.minimum(ratio * adv, torch.clamp(ratio, 1.0 - epsilon, 1.0 + epsilon) * adv).mean()
    value_loss = ((value - returns) ** 2).mean() * value_scale
    entropy = (0.5 + HALF_LOG_2PI + torch.log(torch.exp(logstd))).sum(dim=-1)               # state independent
    entropy_loss = entropy.mean() * entropy_scale
    loss = -policy_loss + value_loss - entropy_loss
    return dict(loss=loss, policy_loss=policy_loss, value_loss=value_loss, entropy_loss=entropy_loss,
                ratio=ratio, value=value, mean=mean)


def compute_gae(rewards, values, bootstrap_values, terminals, gamma, lam):
    """utils.py:45-50 verbatim semantics (f64; no done-mask inside the recursion)."""
    rewards = np.array(rewards)
    values = np.array(list(values) + [bootstrap_values])
    terminals = np.array(terminals)
    deltas = rewards + (1.0 - terminals) * gamma * values[1:] - values[:-1]
    return scipy.signal.lfilter([1], [1, -gamma * lam], deltas[::-1], axis=0)[::-1]


def returns_and_normalized_advantages(advantages, values):
    """train.py:176-177 (f64, population std, whole-horizon batch)."""
    returns = advantages + values
    advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
    return returns, advantages


def minibatch_schedule(num_samples, batch_size, num_epochs):
    """train.py:193-204: per epoch arange+np.random.shuffle (legacy RNG), ceil(T/bs) minibatches, last partial kept."""
    out = []
    for _ in range(num_epochs):
        indices = np.arange(num_samples)
        np.random.shuffle(indices)
        for i in range(int(np.ceil(num_samples / batch_size))):
            begin = i * batch_size
            end = begin + batch_size
            if end > num_samples:
                end = None
            out.append(indices[begin:end])
    return out


class ActionSpace:
    """Duck-typed gym.spaces.Box (gym is not installed): CarlaEnv/carla_lap_env.py:136."""

    def __init__(self, low=(-1.0, 0.0), high=(1.0, 1.0)):
        self.low = np.asarray(low, np.float32)
        self.high = np.asarray(high, np.float32)
        self.shape = self.low.shape


class OraclePPO:
    """Mirror of ppo.PPO (ppo.py:68-276) on CPU."""

    def __init__(self, input_shape, action_space, learning_rate=3e-4, lr_decay=0.998, epsilon=0.2, value_scale=0.5,
                 entropy_scale=0.01, initial_std=0.4, params=None, seed=0, dtype=torch.float32):
        self.input_dim = int(np.asarray(input_shape).reshape(-1)[0])
        self.low, self.high = np.asarray(action_space.low, np.float32), np.asarray(action_space.high, np.float32)
        self.num_actions = int(action_space.shape[0])
        self.learning_rate, self.lr_decay, self.epsilon = learning_rate, lr_decay, epsilon
        self.value_scale, self.entropy_scale, self.dtype = value_scale, entropy_scale, dtype
        src = init_ppo_params(seed, self.input_dim, self.num_actions, initial_std) if params is None else params
        self.params = OrderedDict((k, np.array(v, np.float32)) for k, v in src.items())
        # policy_old has its OWN initial values in the reference (separately initialised graph copy); the
        # trainer always calls update_old_policy() before the first train() (train.py:192), so start equal.
        self.params_old = OrderedDict((k.replace("policy/", "policy_old/", 1), v.copy()) for k, v in self.params.items())
        self.adam = AdamTF(OrderedDict((k, v.shape) for k, v in self.params.items()))
        self.train_step_counter = self.predict_step_counter = self.episode_counter = 0

    def current_lr(self):
        # tf.train.exponential_decay(lr, episode_counter, 1, lr_decay, staircase=True)  (ppo.py:142), fp32 pow
        return np.float32(np.float32(self.learning_rate) * np.power(np.float32(self.lr_decay), np.float32(self.episode_counter)))

    def update_old_policy(self):
        for k, v in self.params.items():
            self.params_old[k.replace("policy/", "policy_old/", 1)] = v.copy()

    def loss_and_grads(self, states, actions, returns, advantage):
        dt = self.dtype
        p = OrderedDict((k, _t(v, dt).clone().requires_grad_(True)) for k, v in self.params.items())
        po = {k: _t(v, dt) for k, v in self.params_old.items()}
        L = ppo_losses(p, po, _t(np.asarray(states, np.float32), dt), _t(np.asarray(actions, np.float32), dt),
                       _t(np.asarray(returns, np.float32), dt), _t(np.asarray(advantage, np.float32), dt),
                       self.low, self.high, self.epsilon, self.value_scale, self.entropy_scale)
        L["loss"].backward()
        grads = OrderedDict((k, (v.grad if v.grad is not None else torch.zeros_like(v)).detach().numpy()) for k, v in p.items())
        scal = {k: float(v.detach()) for k, v in L.items() if k.endswith("loss")}
        scal["ratio_mean"] = float(L["ratio"].detach().mean())
        return scal, grads

    def train(self, states, actions, returns, advantage):
        """ppo.py:218-229 — inputs are rounded to f32 at the feed (ppo.py:108-109)."""
        scal, grads = self.loss_and_grads(states, actions, returns, advantage)
        self.adam.step(self.params, {k: np.asarray(g, np.float32) for k, g in grads.items()}, self.current_lr())
        self.train_step_counter += 1
        return scal

    def predict(self, state, greedy=False, noise=None):
        """ppo.py:231-251; `noise` [M,A] is the injected N(0,1) sample (TF RNG cannot be reproduced)."""
        s = np.asarray(state, np.float32)
        single = s.ndim != 2
        if single:
            s = s[None]
        with torch.no_grad():
            p = {k: _t(v, self.dtype) for k, v in self.params.items()}
            mean, logstd, value = policy_forward(p, _t(s, self.dtype), self.low, self.high)
            if greedy:
                act = mean
            else:
                act = mean + torch.exp(logstd) * _t(np.asarray(noise, np.float32).reshape(mean.shape), self.dtype)
                act = torch.clamp(act, _t(self.low, self.dtype), _t(self.high, self.dtype))
        act, value = act.numpy().astype(np.float32), value.numpy().astype(np.float32)
        if len(s) == 1:
            return act[0], value[0]
        return act, value

    def write_episodic_summaries(self):
        self.episode_counter += 1                                   # ppo.py:271-273 side effect


def ppo_update(model, states, taken_actions, values, rewards, dones, last_value, gamma, lam, num_epochs, batch_size):
    """train.py:171-207 for one horizon: GAE -> returns/normalise -> update_old -> epochs of minibatch SGD."""
    advantages = compute_gae(rewards, values, last_value, dones, gamma, lam)
    returns, advantages = returns_and_normalized_advantages(advantages, values)
    states, taken_actions = np.array(states), np.array(taken_actions)
    returns, advantages = np.array(returns), np.array(advantages)
    model.update_old_policy()
    logs = []
    for mb_idx in minibatch_schedule(len(states), batch_size, num_epochs):
        logs.append(model.train(states[mb_idx], taken_actions[mb_idx], returns[mb_idx], advantages[mb_idx]))
    return logs, returns, advantages
