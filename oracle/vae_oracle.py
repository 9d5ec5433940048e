"""CPU oracle: ConvVAE forward / ELBO / gradients / TF-Adam / epoch loops.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parity: the reference has no tests and TF 1.13 cannot run here, but forward,
losses, all 22 gradients and the Adam trajectory are pinned to the reference's serialized graphs (rgb and seg `.meta`
MetaGraphDefs -> tests/golden/ref_graph_vae_*.json.gz, executed by oracle/tf_graph.py; tests/test_ref_graph.py, 1e-9 in float64);
the epoch loops around them by the golden fixtures + analytic KATs + finite differences in tests/.

Restates (all citations relative to /root/reference):
  vae/models.py:7-9     kl_divergence          -> kl_divergence()
  vae/models.py:11-15   bce_loss               -> bce_with_logits()
  vae/models.py:17-19   bce_loss_v2            -> loss_fn="bce_v2"
  vae/models.py:21-22   mse_loss               -> loss_fn="mse"
  vae/models.py:24-30   verify_range           -> verify_range()
  vae/models.py:85-137  graph: encoder, heads, sample, decoder, ELBO -> vae_forward(), vae_losses()
  vae/models.py:140-142 AdamOptimizer(lr const).minimize -> AdamTF
  vae/models.py:188-231 generate_from_latent / reconstruct / encode / train_one_epoch / evaluate -> OracleVAE
  vae/models.py:249-266 ConvVAE.build_encoder / build_decoder -> vae_forward()
  vae/models.py:271-299 MlpVAE build_mlp / build_encoder / build_decoder -> mlp_vae_variable_specs(), mlp_vae_forward()

Layouts are TensorFlow's: conv kernels HWIO [kh,kw,in,out]; transposed-conv kernels [kh,kw,out,in];
dense kernels [in,out]; activations NHWC; flatten order (H,W,C).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

ENC_FILTERS = (32, 64, 128, 256)          # vae/models.py:250-253
DEC_FILTERS = (128, 64, 32)               # vae/models.py:261-263 (+ target depth for deconv4 :264)
DEC_KERNELS = (4, 4, 5, 4)                # vae/models.py:261-264
ENCODED_SHAPE = (3, 8, 256)               # conv4 output for an 80x160 input (vae/models.py:254)


def vae_variable_specs(z_dim=64, source_shape=(80, 160, 3), target_shape=None):
    """Trainable variables in TF creation order with TF shapes (pinned by tests/golden/ref_variables.json)."""
    target_shape = source_shape if target_shape is None else target_shape
    specs = OrderedDict()
    cin = int(source_shape[-1])
    for i, f in enumerate(ENC_FILTERS):
        specs["vae/encoder/conv%d/kernel" % (i + 1)] = (4, 4, cin, f)
        specs["vae/encoder/conv%d/bias" % (i + 1)] = (f,)
        cin = f
    flat = int(np.prod(ENCODED_SHAPE))
    specs["vae/mean/kernel"] = (flat, z_dim)
    specs["vae/mean/bias"] = (z_dim,)
    specs["vae/logstd_sqare/kernel"] = (flat, z_dim)        # (sic) vae/models.py:98
    specs["vae/logstd_sqare/bias"] = (z_dim,)
    specs["vae/decoder/dense1/kernel"] = (z_dim, flat)
    specs["vae/decoder/dense1/bias"] = (flat,)
    cin = ENCODED_SHAPE[-1]
    for i, (f, k) in enumerate(zip(DEC_FILTERS + (int(target_shape[-1]),), DEC_KERNELS)):
        specs["vae/decoder/deconv%d/kernel" % (i + 1)] = (k, k, f, cin)   # [kh,kw,out,in]
        specs["vae/decoder/deconv%d/bias" % (i + 1)] = (f,)
        cin = f
    return specs


def glorot_uniform(rng, shape):
    """tf.glorot_uniform_initializer as used by tf.layers defaults: fans from the LAST two dims x receptive field."""
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def init_vae_params(seed=0, z_dim=64, source_shape=(80, 160, 3), target_shape=None):
    """Glorot-uniform kernels, zero biases (tf.layers defaults) from numpy RandomState(seed)."""
    rng = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in vae_variable_specs(z_dim, source_shape, target_shape).items():
        out[name] = glorot_uniform(rng, shape) if name.endswith("kernel") else np.zeros(shape, np.float32)
    return out


def verify_range(x, vmin=0.0, vmax=1.0):
    """vae/models.py:24-30 — TF raises InvalidArgumentError; we raise ValueError."""
    x = np.asarray(x)
    if x.size and not (x.min() >= vmin and x.max() <= vmax):
        raise ValueError("verify_range: min=%r max=%r outside [%r, %r]" % (x.min(), x.max(), vmin, vmax))


def _t(a, dtype):
    return a.to(dtype) if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a)).to(dtype)


def bce_with_logits(labels, logits):
    """tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log1p(exp(-|x|))  (vae/models.py:11-15)."""
    return torch.clamp(logits, min=0) - logits * labels + torch.log1p(torch.exp(-torch.abs(logits)))


def kl_divergence(mean, logvar):
    """vae/models.py:7-9."""
    return -0.5 * torch.sum(1.0 + logvar - mean * mean - torch.exp(logvar), dim=1)


class _RoundBF16(torch.autograd.Function):
    """bf16 storage emulation: round the value forward AND the gradient backward (fp32 accumulate in between)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _q(x, storage):
    return _RoundBF16.apply(x) if storage == "bf16" else x


def _qw(w, storage):
    # weights: the bf16 path multiplies a bf16 shadow copy of the fp32 master weights; grads stay fp32
    return (w.to(torch.bfloat16).to(w.dtype) - w).detach() + w if storage == "bf16" else w


def vae_forward(params, src, eps=None, sample=True, dtype=torch.float32, storage="fp32", z_override=None,
                keep=False):
    """Encoder -> heads -> z -> decoder (vae/models.py:97-113,249-266).

    params: dict name -> tensor in TF layout.  src: [B,H,W,C] in [0,1].  eps: [B,z] injected N(0,1) noise
    (the reference's TF RNG is unseeded).  sample=False reproduces `training=False` (z = mean).
    z_override feeds z in place of `self.sample` (generate_from_latent, vae/models.py:188-191).
    storage="bf16" rounds weights and every stored activation (and its gradient) to bf16, emulating the
    HIP path's bf16 HBM layout with fp32 accumulation.
    """
    out = {}
    if z_override is None:
        x = _q(_t(src, dtype), storage).permute(0, 3, 1, 2)                     # NHWC -> NCHW for torch
        for i in range(4):
            w = _qw(params["vae/encoder/conv%d/kernel" % (i + 1)], storage).permute(3, 2, 0, 1)   # HWIO -> OIHW
            x = _q(F.relu(F.conv2d(x, w, params["vae/encoder/conv%d/bias" % (i + 1)], stride=2)), storage)
            if keep:
                out["conv%d" % (i + 1)] = x.permute(0, 2, 3, 1)
        flat = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)                     # flatten in (H,W,C) order
        mean = flat @ _qw(params["vae/mean/kernel"], storage) + params["vae/mean/bias"]
        logvar = flat @ _qw(params["vae/logstd_sqare/kernel"], storage) + params["vae/logstd_sqare/bias"]
        out.update(mean=mean, logvar=logvar)
        if sample:
            assert eps is not None, "training-mode forward needs injected noise"
            z = mean + torch.exp(0.5 * logvar) * _t(eps, dtype)                  # Normal(mean, exp(.5 lv)).sample
        else:
            z = mean
    else:
        z = _t(z_override, dtype)
    out["z"] = z
    h = _q(z, storage) @ _qw(params["vae/decoder/dense1/kernel"], storage) + params["vae/decoder/dense1/bias"]
    x = _q(h, storage).reshape(-1, *ENCODED_SHAPE).permute(0, 3, 1, 2)
    for i in range(4):
        w = _qw(params["vae/decoder/deconv%d/kernel" % (i + 1)], storage).permute(3, 2, 0, 1)     # [kh,kw,out,in] -> [in,out,kh,kw]
        x = F.conv_transpose2d(x, w, params["vae/decoder/deconv%d/bias" % (i + 1)], stride=2)
        if i < 3:
            x = _q(F.relu(x), storage)
        if keep:
            out["deconv%d" % (i + 1)] = x.permute(0, 2, 3, 1)
    logits = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)                        # [B, H*W*C_t]
    out["logits"] = logits
    return out


def mlp_vae_variable_specs(z_dim=64, source_shape=(80, 160, 3), target_shape=None, encoder_sizes=(512, 256), decoder_sizes=(256, 512)):
    """Trainable variables of MlpVAE in TF creation order (tf.layers.dense default names inside variable_scope("encoder") /
    ("decoder"): dense, dense_1, ...; vae/models.py:287-297)."""
    target_shape = source_shape if target_shape is None else target_shape
    specs = OrderedDict()
    cin = int(np.prod(source_shape))
    for i, h in enumerate(encoder_sizes):
        name = "vae/encoder/dense" + ("_%d" % i if i else "")
        specs[name + "/kernel"], specs[name + "/bias"] = (cin, int(h)), (int(h),)
        cin = int(h)
    for head in ("mean", "logstd_sqare"):
        specs["vae/%s/kernel" % head], specs["vae/%s/bias" % head] = (cin, z_dim), (z_dim,)
    cin = z_dim
    for i, h in enumerate(list(decoder_sizes) + [int(np.prod(target_shape))]):
        name = "vae/decoder/dense" + ("_%d" % i if i else "")
        specs[name + "/kernel"], specs[name + "/bias"] = (cin, int(h)), (int(h),)
        cin = int(h)
    return specs


def init_mlp_vae_params(seed=0, **kw):
    rng = np.random.RandomState(seed)
    return OrderedDict((n, glorot_uniform(rng, s) if n.endswith("kernel") else np.zeros(s, np.float32)) for n, s in mlp_vae_variable_specs(**kw).items())


def mlp_vae_forward(params, src, eps=None, sample=True, dtype=torch.float32, storage="fp32", z_override=None):
    """MlpVAE (vae/models.py:287-297): flatten -> dense+ReLU per encoder size (the LAST one too: output_activation=relu) -> heads -> z ->
    dense+ReLU per decoder size -> dense(prod(target_shape)) without activation = logits."""
    enc = sorted((k for k in params if k.startswith("vae/encoder/") and k.endswith("kernel")), key=lambda k: (len(k), k))
    dec = sorted((k for k in params if k.startswith("vae/decoder/") and k.endswith("kernel")), key=lambda k: (len(k), k))
    out = {}
    if z_override is None:
        x = _q(_t(src, dtype).reshape(len(src), -1), storage)
        for k in enc:
            x = _q(F.relu(x @ _qw(params[k], storage) + params[k[:-6] + "bias"]), storage)
        mean = x @ _qw(params["vae/mean/kernel"], storage) + params["vae/mean/bias"]
        logvar = x @ _qw(params["vae/logstd_sqare/kernel"], storage) + params["vae/logstd_sqare/bias"]
        out.update(mean=mean, logvar=logvar)
        if sample:
            assert eps is not None, "training-mode forward needs injected noise"
            z = mean + torch.exp(0.5 * logvar) * _t(eps, dtype)
        else:
            z = mean
    else:
        z = _t(z_override, dtype)
    out["z"] = z
    x = _q(z, storage)
    for i, k in enumerate(dec):
        x = x @ _qw(params[k], storage) + params[k[:-6] + "bias"]
        if i + 1 < len(dec):
            x = _q(F.relu(x), storage)
    out["logits"] = x
    return out


def mlp_vae_loss_and_grads(params_np, src, tgt, eps, beta=1.0, kl_tolerance=0.0, loss_fn="bce", dtype=torch.float32, storage="fp32"):
    params = OrderedDict((k, _t(v, dtype).clone().requires_grad_(True)) for k, v in params_np.items())
    fw = mlp_vae_forward(params, src, eps, sample=True, dtype=dtype, storage=storage)
    recon, kl, loss = vae_losses(fw, tgt, beta, kl_tolerance, loss_fn, dtype=dtype)
    loss.backward()
    grads = OrderedDict((k, (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy()) for k, p in params.items())
    return (float(recon.detach()), float(kl.detach()), float(loss.detach())), grads, {k: v.detach() for k, v in fw.items()}


def vae_losses(fw, tgt, beta=1.0, kl_tolerance=0.0, loss_fn="bce", z_dim=None, dtype=torch.float32):
    """ELBO pieces (vae/models.py:122-137): recon = mean_B(sum_pix loss), kl = mean_B(max(kl_b, tol*z))."""
    logits = fw["logits"]
    labels = _t(tgt, dtype).reshape(logits.shape[0], -1)
    if loss_fn == "bce":
        per = bce_with_logits(labels, logits)
    elif loss_fn == "bce_v2":
        t = torch.sigmoid(logits)
        per = -(labels * torch.log(1e-10 + t) + (1 - labels) * torch.log(1e-10 + 1 - t))
    elif loss_fn == "mse":
        per = (labels - torch.sigmoid(logits)) ** 2
    else:
        raise ValueError(loss_fn)
    recon = per.sum(dim=1).mean()
    kl_b = kl_divergence(fw["mean"], fw["logvar"])
    if kl_tolerance > 0:
        zd = fw["mean"].shape[1] if z_dim is None else z_dim
        kl_b = torch.maximum(kl_b, torch.full_like(kl_b, kl_tolerance * zd))
    kl = kl_b.mean()
    return recon, kl, recon + beta * kl


def vae_loss_and_grads(params_np, src, tgt, eps, beta=1.0, kl_tolerance=0.0, loss_fn="bce",
                       dtype=torch.float32, storage="fp32"):
    """One forward+backward. Returns (recon, kl, loss) floats and grads dict (TF layouts, numpy)."""
    params = OrderedDict((k, _t(v, dtype).clone().requires_grad_(True)) for k, v in params_np.items())
    fw = vae_forward(params, src, eps, sample=True, dtype=dtype, storage=storage)
    recon, kl, loss = vae_losses(fw, tgt, beta, kl_tolerance, loss_fn, dtype=dtype)
    loss.backward()
    grads = OrderedDict((k, (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy()) for k, p in params.items())
    return (float(recon.detach()), float(kl.detach()), float(loss.detach())), grads, {k: v.detach() for k, v in fw.items()}


class AdamTF:
    """tf.train.AdamOptimizer exactly as the TF 1.13 ApplyAdam kernel computes it:

        alpha = lr * sqrt(1 - beta2_power) / (1 - beta1_power)      (fp32)
        m    += (g - m) * (1 - beta1)
        v    += (g*g - v) * (1 - beta2)
        var  -= (m * alpha) / (sqrt(v) + epsilon)
        beta1_power *= beta1 ; beta2_power *= beta2                 (after all variables)

    beta1=0.9, beta2=0.999, epsilon=1e-8 — "epsilon hat" form, NOT torch.optim.Adam's.
    """

    def __init__(self, names_shapes, beta1=0.9, beta2=0.999, epsilon=1e-8, dtype=np.float32):
        # dtype=np.float64: the same recurrence without float32 rounding -- the "exact trajectory" the parity tests measure BOTH float32
        # implementations (this oracle's and the device's) against
        self.ft = ft = dtype
        self.b1, self.b2, self.eps = ft(beta1), ft(beta2), ft(epsilon)
        self.m = OrderedDict((k, np.zeros(s, ft)) for k, s in names_shapes.items())
        self.v = OrderedDict((k, np.zeros(s, ft)) for k, s in names_shapes.items())
        self.beta1_power, self.beta2_power = ft(beta1), ft(beta2)

    def alpha(self, lr):
        ft = self.ft
        one = ft(1.0)
        return ft(ft(lr) * np.sqrt(one - self.beta2_power, dtype=ft) / (one - self.beta1_power))

    def step(self, params, grads, lr):
        ft = self.ft
        a = self.alpha(lr)
        one = ft(1.0)
        for k in self.m:
            g = np.asarray(grads[k], ft)
            self.m[k] += (g - self.m[k]) * (one - self.b1)
            self.v[k] += (g * g - self.v[k]) * (one - self.b2)
            params[k] -= (self.m[k] * a) / (np.sqrt(self.v[k]) + self.eps)
        self.beta1_power = ft(self.beta1_power * self.b1)
        self.beta2_power = ft(self.beta2_power * self.b2)


class OracleVAE:
    """Mirror of the reference VAE/ConvVAE class surface on CPU (vae/models.py:33-268), numpy in / numpy out."""

    def __init__(self, source_shape=(80, 160, 3), target_shape=None, z_dim=64, beta=1.0, learning_rate=1e-4,
                 lr_decay=0.98, kl_tolerance=0.0, loss_fn="bce", training=True, params=None, seed=0,
                 dtype=torch.float32, storage="fp32"):
        self.source_shape = tuple(int(s) for s in source_shape)
        self.target_shape = self.source_shape if target_shape is None else tuple(int(s) for s in target_shape)
        self.z_dim, self.beta, self.kl_tolerance = int(z_dim), float(beta), float(kl_tolerance)
        self.learning_rate, self.lr_decay, self.loss_fn, self.training = learning_rate, lr_decay, loss_fn, training
        self.dtype, self.storage = dtype, storage
        src = init_vae_params(seed, z_dim, self.source_shape, self.target_shape) if params is None else params
        self.params = OrderedDict((k, np.array(v, np.float32)) for k, v in src.items())
        self.adam = AdamTF(OrderedDict((k, v.shape) for k, v in self.params.items()))
        self.step_idx = 0
        self.last_metrics = None

    # -- single SGD step (one sess.run([train_step, ...]), vae/models.py:213-216) --
    def train_step(self, src, tgt, eps):
        verify_range(src), verify_range(tgt)
        (recon, kl, loss), grads, _ = vae_loss_and_grads(self.params, src, tgt, eps, self.beta, self.kl_tolerance,
                                                         self.loss_fn, self.dtype, self.storage)
        self.adam.step(self.params, grads, self.learning_rate)      # constant lr (vae/models.py:141)
        return recon, kl

    def eval_step(self, src, tgt, eps):
        with torch.no_grad():
            p = {k: _t(v, self.dtype) for k, v in self.params.items()}
            fw = vae_forward(p, src, eps, sample=self.training, dtype=self.dtype, storage=self.storage)
            recon, kl, _ = vae_losses(fw, tgt, self.beta, self.kl_tolerance, self.loss_fn, dtype=self.dtype)
        return float(recon), float(kl)

    # -- epoch loops: legacy numpy RNG permutation, N//bs full minibatches (vae/models.py:207-231) --
    def train_one_epoch(self, train_source, train_target, batch_size, eps_fn):
        indices = np.arange(len(train_source))
        np.random.shuffle(indices)
        recs, kls = [], []
        for i in range(train_source.shape[0] // batch_size):
            mb_idx = indices[i * batch_size:(i + 1) * batch_size]
            r, k = self.train_step(train_source[mb_idx], train_target[mb_idx], eps_fn(len(mb_idx)))
            recs.append(r), kls.append(k)
        self.step_idx += 1
        self.last_metrics = [float(np.mean(recs)), float(np.mean(kls))] if recs else [0.0, 0.0]
        return self.last_metrics

    def evaluate(self, val_source, val_target, batch_size, eps_fn):
        indices = np.arange(len(val_source))
        np.random.shuffle(indices)
        recs, kls = [], []
        for i in range(val_source.shape[0] // batch_size):
            mb_idx = indices[i * batch_size:(i + 1) * batch_size]
            r, k = self.eval_step(val_source[mb_idx], val_target[mb_idx], eps_fn(len(mb_idx)))
            recs.append(r), kls.append(k)
        return [float(np.mean(recs)), float(np.mean(kls))] if recs else [0.0, 0.0]

    # -- inference surface (vae/models.py:188-202) --
    def _fw(self, **kw):
        with torch.no_grad():
            p = {k: _t(v, self.dtype) for k, v in self.params.items()}
            return vae_forward(p, dtype=self.dtype, storage=self.storage, **kw)

    def encode(self, source_states):
        src = np.asarray(source_states, np.float32)
        verify_range(src)
        return self._fw(src=src, sample=False)["mean"].numpy().astype(np.float32)

    def reconstruct(self, source_states, eps=None):
        src = np.asarray(source_states, np.float32)
        verify_range(src)
        fw = self._fw(src=src, eps=eps, sample=self.training)
        rec = torch.sigmoid(fw["logits"]).numpy().astype(np.float32)
        return [s.reshape(self.source_shape) for s in rec]          # (sic) source_shape, vae/models.py:197

    def generate_from_latent(self, z):
        fw = self._fw(src=None, z_override=np.asarray(z, np.float32))
        return torch.sigmoid(fw["logits"]).numpy().astype(np.float32)
