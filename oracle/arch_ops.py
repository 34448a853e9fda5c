"""ORACLE -- test infrastructure only (never imported by the product path).

CPU restatement (PyTorch-CPU, fp64 by default) of the layer arithmetic in the reference's
compare_gan/architectures/arch_ops.py and resnet_ops.py.  The reference itself cannot be imported
here (TensorFlow 1.x / gin / tensorflow_gan are not installable offline -- SURVEY.md section 8c),
so every function follows the cited reference lines and is pinned against the reference's own
golden vectors in tests/test_oracle_pins.py:
  * batch-norm golden array            architectures/arch_ops_test.py:32-61
  * accumulator semantics              architectures/arch_ops_test.py:63-132
  * zero-insertion unpool              architectures/resnet_ops.py:35-56 (traced by hand)
Conv / deconv / spectral-norm / attention numerics are NOT pinned by any reference test
("parity unpinned" for those rows); TF op semantics are restated from the TF1 API contract
(SURVEY.md App. A).

Layouts are the reference's: activations NHWC, conv kernels HWIO, deconv kernels
[kh, kw, Cout, Cin], linear kernels [in, out].
"""
import math

import torch
import torch.nn.functional as F


def _bf16(t):
  return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


class _RoundBoth(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x):
    return _bf16(x)

  @staticmethod
  def backward(ctx, g):
    return _RoundBoth.apply(g)


class _RoundFwd(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x):
    return _bf16(x)

  @staticmethod
  def backward(ctx, g):
    return g


# ------------------------------------------------------------------------------------------------
# Variable store (tf.get_variable / tf.variable_scope(reuse=AUTO_REUSE) stand-in)
# ------------------------------------------------------------------------------------------------
class VarStore(object):
  """name -> tensor, created on first use in call order (abstract_arch.py:71-74 AUTO_REUSE)."""

  def __init__(self, dtype=torch.float64, seed=0, weights_initializer="normal",
               weights_stddev=0.02, emulate_bf16=False, device="cpu"):
    # emulate_bf16: snap every tensor the HIP path STORES in bf16 (activations, their gradients,
    # MFMA weight operands) to the bf16 grid, keeping all arithmetic in fp64.  ReLU networks have
    # discontinuous gradients, so the exact-fp64 oracle and a bf16 pipeline disagree by O(sqrt(eps))
    # through flipped ReLU masks; this mode removes that effect and isolates real defects.
    self.emulate_bf16 = emulate_bf16
    # device: where the variables live.  "cpu" is the oracle proper; a CUDA device runs the SAME
    # restatement in fp64 on plain torch ops (no libcgamd kernel) for the parity tests at the
    # benchmark's batch sizes, where the CPU needs minutes per case (tests/ only)
    self.device = torch.device(device)
    self.vars = {}
    self.trainable = []
    self.dtype = dtype
    self.gen = torch.Generator().manual_seed(seed)
    self.weights_initializer = weights_initializer  # gin "weights.initializer"
    self.weights_stddev = weights_stddev            # gin "weights.stddev"

  def q(self, t):
    """Activation / gradient storage rounding (identity unless emulate_bf16)."""
    return _RoundBoth.apply(t) if self.emulate_bf16 else t

  def q_in(self, t):
    """Rounding of an op INPUT that is already stored in bf16: idempotent in the forward pass; in
    the backward pass it is where the consumer's input-gradient gets stored (bf16 unless
    round_input_grads is switched off to model fp32 gradient tensors)."""
    if not self.emulate_bf16:
      return t
    return _RoundBoth.apply(t) if getattr(self, "round_input_grads", True) else _RoundFwd.apply(t)

  def qw(self, w):
    """MFMA weight-operand rounding: forward only (weight gradients stay fp32)."""
    return _RoundFwd.apply(w) if self.emulate_bf16 else w

  def get(self, name, shape, init, trainable=True):
    if name not in self.vars:
      v = init(tuple(shape)).to(self.dtype).to(self.device)
      if trainable:
        v.requires_grad_(True)
        self.trainable.append(name)
      self.vars[name] = v
    v = self.vars[name]
    if tuple(v.shape) != tuple(shape):
      raise ValueError("variable %s has shape %s, requested %s" % (name, tuple(v.shape), shape))
    return v

  # initialisers -------------------------------------------------------------------------------
  def weight_init(self, stddev=0.02):
    """arch_ops.py:46-63 weight_initializer (the gin-bound stddev overrides the call site's)."""
    kind, sd = self.weights_initializer, self.weights_stddev
    if kind == "normal":
      return lambda s: torch.randn(s, generator=self.gen, dtype=torch.float64) * sd
    if kind == "truncated":
      def trunc(s):
        t = torch.randn(s, generator=self.gen, dtype=torch.float64)
        bad = t.abs() > 2
        while bad.any():
          t[bad] = torch.randn(int(bad.sum()), generator=self.gen, dtype=torch.float64)
          bad = t.abs() > 2
        return t * sd
      return trunc
    if kind == "orthogonal":
      def orth(s):
        rows = 1
        for d in s[:-1]:
          rows *= d
        cols = s[-1]
        a = torch.randn((max(rows, cols), min(rows, cols)), generator=self.gen,
                        dtype=torch.float64)
        q, r = torch.linalg.qr(a)
        q = q * torch.sign(torch.diagonal(r))
        if rows < cols:
          q = q.t()
        return q.reshape(s)
      return orth
    raise ValueError("Unknown weight initializer {}.".format(kind))

  def normal_init(self, stddev=1.0):
    return lambda s: torch.randn(s, generator=self.gen, dtype=torch.float64) * stddev

  def glorot_normal_init(self):
    def init(s):
      fan_in, fan_out = s[0], s[1]
      sd = math.sqrt(2.0 / (fan_in + fan_out)) / 0.87962566103423978  # TF truncated-normal glorot
      t = torch.randn(s, generator=self.gen, dtype=torch.float64).clamp_(-2, 2)
      return t * sd
    return init

  @staticmethod
  def const_init(value):
    return lambda s: torch.full(s, float(value), dtype=torch.float64)


# ------------------------------------------------------------------------------------------------
# Convolution geometry (TF 'SAME')
# ------------------------------------------------------------------------------------------------
def same_pads(size, k, stride):
  out = -(-size // stride)
  total = max((out - 1) * stride + k - size, 0)
  return out, total // 2, total - total // 2


def conv2d_same(x, w, stride):
  """tf.nn.conv2d(x, w, strides=[1,s,s,1], padding='SAME')  (arch_ops.py:568).  x NHWC, w HWIO."""
  if x.is_cuda:
    return conv2d_same_gemm(x, w, stride)
  kh, kw = w.shape[0], w.shape[1]
  _, pt, pb = same_pads(x.shape[1], kh, stride)
  _, pl, pr = same_pads(x.shape[2], kw, stride)
  xp = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
  y = F.conv2d(xp, w.permute(3, 2, 0, 1), stride=stride)
  return y.permute(0, 2, 3, 1)


def conv2d_same_gemm(x, w, stride):
  """The same convolution as one matrix product per filter tap (the shifted, strided view of the
  padded input times w[r, s]): every step is a plain tensor op, so it runs in fp64 on any device
  and differentiates to any order.  tests/test_oracle_direct.py pins it to conv2d_same and to the
  direct-loop restatement (oracle/direct.py)."""
  kh, kw, ci, co = w.shape
  n, h, wd, _ = x.shape
  ho, pt, pb = same_pads(h, kh, stride)
  wo, pl, pr = same_pads(wd, kw, stride)
  xp = F.pad(x, (0, 0, pl, pr, pt, pb))
  out = None
  for r in range(kh):
    for s in range(kw):
      v = xp[:, r:r + (ho - 1) * stride + 1:stride, s:s + (wo - 1) * stride + 1:stride, :]
      t = v.reshape(-1, ci) @ w[r, s]
      out = t if out is None else out + t
  return out.reshape(n, ho, wo, co)


def conv2d_transpose_same(x, w, out_hw, stride):
  """tf.nn.conv2d_transpose(x, w, output_shape, strides, 'SAME') (arch_ops.py:588-589): the exact
  adjoint of conv2d_same(y, w, stride) for y of spatial size out_hw; w is [kh, kw, Cout, Cin]."""
  if x.is_cuda:
    return conv2d_transpose_same_gemm(x, w, out_hw, stride)
  kh, kw = w.shape[0], w.shape[1]
  Hy, Wy = out_hw
  _, pt, _ = same_pads(Hy, kh, stride)
  _, pl, _ = same_pads(Wy, kw, stride)
  full = F.conv_transpose2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), stride=stride)
  fh, fw = full.shape[2], full.shape[3]
  need_h, need_w = pt + Hy, pl + Wy
  if need_h > fh or need_w > fw:
    full = F.pad(full, (0, max(need_w - fw, 0), 0, max(need_h - fh, 0)))
  return full[:, :, pt:pt + Hy, pl:pl + Wy].permute(0, 2, 3, 1)


def conv2d_transpose_same_gemm(x, w, out_hw, stride):
  """The same transposed convolution as one matrix product per filter tap, scattered into the
  padded output (the adjoint of conv2d_same_gemm tap by tap): plain tensor ops, fp64 on any
  device.  tests/test_oracle_direct.py pins it to conv2d_transpose_same."""
  kh, kw, cout, cin = w.shape
  n, h, wd, _ = x.shape
  Hy, Wy = out_hw
  ho, pt, pb = same_pads(Hy, kh, stride)
  wo, pl, pr = same_pads(Wy, kw, stride)
  assert (ho, wo) == (h, wd), "input is not the strided size of the requested output"
  full = x.new_zeros((n, Hy + pt + pb, Wy + pl + pr, cout))
  x2 = x.reshape(-1, cin)
  for r in range(kh):
    for s in range(kw):
      t = (x2 @ w[r, s].t()).reshape(n, h, wd, cout)
      v = full[:, r:r + (h - 1) * stride + 1:stride, s:s + (wd - 1) * stride + 1:stride, :]
      v += t
  return full[:, pt:pt + Hy, pl:pl + Wy, :]


def unpool(x):
  """resnet_ops.py:35-56: zero insertion, out[b,2h,2w,c] = x[b,h,w,c], zeros elsewhere."""
  n, h, w, c = x.shape
  out = x.new_zeros((n, 2 * h, 2 * w, c))
  out[:, ::2, ::2, :] = x
  return out


def avg_pool2(x):
  """tf.nn.pool(x, [2,2], 'AVG', 'SAME', strides=[2,2]) (resnet_ops.py:132-133): output size
  ceil(size / 2), a window that overhangs the border averages its valid elements only."""
  return F.avg_pool2d(x.permute(0, 3, 1, 2), 2, 2, ceil_mode=True,
                      count_include_pad=False).permute(0, 2, 3, 1)


def max_pool2(x):
  """tf.layers.max_pooling2d(pool_size=[2,2], strides=2) VALID (arch_ops.py:741,750)."""
  return F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)


def lrelu(x, leak=0.2):
  """arch_ops.py:595-597 tf.maximum(x, leak*x)."""
  return torch.maximum(x, leak * x)


# ------------------------------------------------------------------------------------------------
# Spectral norm (arch_ops.py:453-535)
# ------------------------------------------------------------------------------------------------
def l2_normalize(x, eps):
  """tf.math.l2_normalize(x, axis=None, epsilon): x * rsqrt(max(sum(x^2), eps))."""
  return x * torch.rsqrt(torch.clamp(torch.sum(x * x), min=eps))


def sn_mode(shape2d, singular_value):
  if singular_value == "auto":
    singular_value = "left" if shape2d[0] <= shape2d[1] else "right"  # arch_ops.py:489-490
  return singular_value


def spectral_norm(vs, w, var_name, epsilon=1e-12, singular_value="left", update=True):
  """Returns w / sigma after ONE power-iteration round; persists u (arch_ops.py:479-535).

  var_name is the full variable name of `w` ("<scope>/kernel"); the vector is stored as
  "<scope>/kernel/u_var" (arch_ops.py:487-498), shape [K,1] (left) or [1,Cout] (right)."""
  w2 = w.reshape(-1, w.shape[-1])
  mode = sn_mode(w2.shape, singular_value)
  u_shape = (w2.shape[0], 1) if mode == "left" else (1, w2.shape[1])
  u_var = vs.get(var_name + "/u_var", u_shape, vs.normal_init(1.0), trainable=False)
  u = u_var.detach()
  wd = w2.detach()
  if mode == "left":
    v = l2_normalize(wd.t() @ u, epsilon)       # :507-508
    u = l2_normalize(wd @ v, epsilon)           # :509
  else:
    v = l2_normalize(u @ wd.t(), epsilon)       # :511-512
    u = l2_normalize(v @ wd, epsilon)           # :513
  if update:
    with torch.no_grad():
      u_var.copy_(u)                            # :516 tf.assign(u_var, u)
  if mode == "left":
    norm_value = (u.t() @ w2) @ v               # :525 (gradient flows through w only, :521-522)
  else:
    norm_value = (v @ w2) @ u.t()               # :527
  return (w2 / norm_value).reshape(w.shape)     # :531-535


# ------------------------------------------------------------------------------------------------
# linear / conv2d / deconv2d (arch_ops.py:538-592)
# ------------------------------------------------------------------------------------------------
class SNConfig(object):
  def __init__(self, epsilon=1e-12, singular_value="left"):
    self.epsilon = epsilon
    self.singular_value = singular_value


def linear(vs, x, output_size, scope, sn_cfg, bias_start=0.0, use_sn=False, use_bias=True,
           kernel_init=None, out_f32=False):
  """out_f32 marks outputs the HIP path keeps in fp32 (logits, CBN gamma/beta): no storage
  rounding in emulate_bf16 mode."""
  kernel = vs.get(scope + "/kernel", (x.shape[1], output_size),
                  kernel_init or vs.weight_init())
  if use_sn:
    kernel = spectral_norm(vs, kernel, scope + "/kernel", sn_cfg.epsilon, sn_cfg.singular_value)
  out = vs.q_in(x) @ vs.qw(kernel)
  if use_bias:
    out = out + vs.get(scope + "/bias", (output_size,), vs.const_init(bias_start))
  return out if out_f32 else vs.q(out)


def conv2d(vs, x, output_dim, k_h, k_w, d_h, d_w, name, sn_cfg, use_sn=False, use_bias=True,
           residual=None, out_f32=False):
  """`residual` is added before the (emulated) storage rounding, as the fused HIP epilogue does;
  in exact arithmetic conv + bias + residual is the reference's `output += shortcut`."""
  w = vs.get(name + "/kernel", (k_h, k_w, x.shape[-1], output_dim), vs.weight_init())
  if use_sn:
    w = spectral_norm(vs, w, name + "/kernel", sn_cfg.epsilon, sn_cfg.singular_value)
  assert d_h == d_w
  out = conv2d_same(vs.q_in(x), vs.qw(w), d_h)
  if use_bias:
    out = out + vs.get(name + "/bias", (output_dim,), vs.const_init(0.0))
  if residual is not None:
    out = out + residual
  return out if out_f32 else vs.q(out)


def deconv2d(vs, x, output_shape, k_h, k_w, d_h, d_w, name, sn_cfg, use_sn=False, out_f32=False):
  w = vs.get(name + "/kernel", (k_h, k_w, output_shape[-1], x.shape[-1]), vs.weight_init())
  if use_sn:
    w = spectral_norm(vs, w, name + "/kernel", sn_cfg.epsilon, sn_cfg.singular_value)
  assert d_h == d_w
  out = conv2d_transpose_same(vs.q_in(x), vs.qw(w), (output_shape[1], output_shape[2]), d_h)
  out = out + vs.get(name + "/bias", (output_shape[-1],), vs.const_init(0.0))
  return out if out_f32 else vs.q(out)


# ------------------------------------------------------------------------------------------------
# Batch norm family (arch_ops.py:66-445)
# ------------------------------------------------------------------------------------------------
class BNConfig(object):
  """gin bindings standardize_batch.{decay,epsilon,use_moving_averages} (arch_ops.py:194-202)."""

  def __init__(self, decay=0.999, epsilon=1e-3, use_moving_averages=True,
               cross_replica=None):
    self.decay = decay
    self.epsilon = epsilon
    self.use_moving_averages = use_moving_averages
    self.cross_replica = cross_replica  # callable(mean, mean_sq) -> (mean, mean_sq) or None


def accumulated_moments_for_inference(vs, scope, mean, variance, is_training):
  """arch_ops.py:122-191."""
  c = mean.shape
  accu_mean = vs.get(scope + "accu/accu_mean", c, vs.const_init(0.0), trainable=False)
  accu_var = vs.get(scope + "accu/accu_variance", c, vs.const_init(0.0), trainable=False)
  accu_counter = vs.get(scope + "accu/accu_counter", (), vs.const_init(1e-12), trainable=False)
  update_accus = vs.get(scope + "accu/update_accus", (), vs.const_init(0), trainable=False)
  if is_training:
    return mean, variance
  if int(update_accus.item()) == 1:
    with torch.no_grad():
      accu_mean.add_(mean.detach())
      accu_var.add_(variance.detach())
      accu_counter.add_(1)
  return accu_mean / accu_counter, accu_var / accu_counter


def moving_moments_for_inference(vs, scope, mean, variance, is_training, decay):
  """arch_ops.py:66-119 (assign_moving_average, zero_debias=False)."""
  c = mean.shape
  mm = vs.get(scope + "moving_mean", c, vs.const_init(0.0), trainable=False)
  mv = vs.get(scope + "moving_variance", c, vs.const_init(1.0), trainable=False)
  if is_training:
    with torch.no_grad():
      mm.sub_((1 - decay) * (mm - mean.detach()))
      mv.sub_((1 - decay) * (mv - variance.detach()))
    return mean, variance
  return mm, mv


def standardize_batch(vs, x, is_training, scope, bn_cfg):
  """arch_ops.py:194-319; scope is the variable-scope prefix ending in '/' (or '')."""
  rank2 = x.dim() == 2
  if rank2:
    x = x.reshape(-1, 1, 1, x.shape[-1])          # :281-285
  if x.dim() != 4:
    raise ValueError("Inputs has unsupported rank. Expected 2 or 4 but got %d" % x.dim())
  mean = x.mean(dim=(0, 1, 2))                     # sufficient_statistics / normalize_moments
  mean_sq = (x * x).mean(dim=(0, 1, 2))
  if bn_cfg.cross_replica is not None:             # tpu_ops.cross_replica_moments (:291-292)
    mean, mean_sq = bn_cfg.cross_replica(mean, mean_sq)
  variance = mean_sq - mean * mean                 # :294-297 (biased)
  if bn_cfg.use_moving_averages:
    mean, variance = moving_moments_for_inference(vs, scope, mean, variance, is_training,
                                                  bn_cfg.decay)
  else:
    mean, variance = accumulated_moments_for_inference(vs, scope, mean, variance, is_training)
  out = (x - mean) * torch.rsqrt(variance + bn_cfg.epsilon)  # tf.nn.batch_normalization :306-312
  if rank2:
    out = out.reshape(-1, out.shape[-1])
  return out


def _finish_bn(vs, out, relu):
  """tf.nn.relu follows every BN of the example generators (resnet_ops.py:165,175); the HIP kernel
  stores relu(bn(x)) once, so the storage rounding sits after the ReLU."""
  if relu:
    out = torch.relu(out)
  return vs.q(out) if getattr(vs, "round_bn_output_grads", True) else vs.qw(out)


def batch_norm(vs, x, is_training, name, bn_cfg, center=True, scale=True, relu=False):
  """arch_ops.py:327-367 (name = full scope, e.g. 'generator/B1/bn1')."""
  out = standardize_batch(vs, vs.q_in(x), is_training, name + "/", bn_cfg)
  c = x.shape[-1]
  if scale:
    out = out * vs.get(name + "/gamma", (c,), vs.const_init(1.0))
  if center:
    out = out + vs.get(name + "/beta", (c,), vs.const_init(0.0))
  return _finish_bn(vs, out, relu)


def conditional_batch_norm(vs, x, y, is_training, use_sn, name, bn_cfg, sn_cfg, use_bias=False,
                           relu=False):
  """arch_ops.py:423-445: gamma = linear(y), beta = linear(y), NO +1 offset."""
  if y is None:
    raise ValueError("You must provide y for conditional batch normalization.")
  if y.dim() != 2:
    raise ValueError("Conditioning must have rank 2.")
  out = standardize_batch(vs, vs.q_in(x), is_training, name + "/", bn_cfg)
  c = x.shape[-1]
  gamma = linear(vs, y, c, name + "/condition/gamma", sn_cfg, use_sn=use_sn, use_bias=use_bias,
                 out_f32=True)
  beta = linear(vs, y, c, name + "/condition/beta", sn_cfg, use_sn=use_sn, use_bias=use_bias,
                out_f32=True)
  return _finish_bn(vs, out * gamma.reshape(-1, 1, 1, c) + beta.reshape(-1, 1, 1, c), relu)


def self_modulated_batch_norm(vs, x, z, is_training, use_sn, name, bn_cfg, sn_cfg, num_hidden=32,
                              relu=False):
  """arch_ops.py:370-420."""
  if z is None:
    raise ValueError("You must provide z for self modulation.")
  out = standardize_batch(vs, vs.q_in(x), is_training, name + "/", bn_cfg)
  c = x.shape[-1]
  h = z
  if num_hidden > 0:
    h = torch.relu(linear(vs, h, num_hidden, name + "/sbn/hidden", sn_cfg, use_sn=use_sn))
  gamma = linear(vs, h, c, name + "/sbn/gamma", sn_cfg, bias_start=1.0, use_sn=use_sn,
                 out_f32=True)
  beta = linear(vs, h, c, name + "/sbn/beta", sn_cfg, use_sn=use_sn, out_f32=True)
  return _finish_bn(vs, out * gamma.reshape(-1, 1, 1, c) + beta.reshape(-1, 1, 1, c), relu)


# ------------------------------------------------------------------------------------------------
# Self-attention (arch_ops.py:709-758)
# ------------------------------------------------------------------------------------------------
def layer_norm(vs, x, is_training, scope):
  """tf.contrib.layers.layer_norm(x, trainable=is_training, scope=scope) (arch_ops.py:448-450) with
  the library defaults: moments over axes [1, rank) per sample, beta (zeros) / gamma (ones) over
  the last axis, tf.nn.batch_normalization with variance_epsilon = 1e-12."""
  c = x.shape[-1]
  beta = vs.get(scope + "/beta", (c,), vs.const_init(0.0), trainable=bool(is_training))
  gamma = vs.get(scope + "/gamma", (c,), vs.const_init(1.0), trainable=bool(is_training))
  xin = vs.q_in(x)
  axes = tuple(range(1, x.dim()))
  mean = xin.mean(dim=axes, keepdim=True)
  var = ((xin - mean) ** 2).mean(dim=axes, keepdim=True)
  return vs.q((xin - mean) * torch.rsqrt(var + 1e-12) * gamma + beta)


def non_local_block(vs, x, name, use_sn, sn_cfg):
  n, h, w, c = x.shape
  ca, cg = c // 8, c // 2
  theta = conv2d(vs, x, ca, 1, 1, 1, 1, name + "/conv2d_theta", sn_cfg, use_sn, use_bias=False)
  theta = theta.reshape(n, h * w, ca)
  phi = conv2d(vs, x, ca, 1, 1, 1, 1, name + "/conv2d_phi", sn_cfg, use_sn, use_bias=False)
  phi = max_pool2(phi).reshape(n, h * w // 4, ca)
  attn = torch.softmax(theta @ phi.transpose(1, 2), dim=-1)
  g = conv2d(vs, x, cg, 1, 1, 1, 1, name + "/conv2d_g", sn_cfg, use_sn, use_bias=False)
  g = max_pool2(g).reshape(n, h * w // 4, cg)
  attn_g = vs.q((attn @ g).reshape(n, h, w, cg))
  sigma = vs.get(name + "/sigma", (), vs.const_init(0.0))
  attn_g = conv2d(vs, attn_g, c, 1, 1, 1, 1, name + "/conv2d_attn_g", sn_cfg, use_sn,
                  use_bias=False)
  return vs.q(x + sigma * attn_g)
