"""Layer ops of the G/D graphs on the HIP kernels -- the product-side mirror of the reference's
compare_gan/architectures/arch_ops.py (same function names, argument meaning, gin configurables
and error behaviour; cited per function).

Variables live in a `VariableStore` that plays the role of TF's variable scopes: created on first
use, in call order, under the reference's names, then reused.  Every op runs a
hand-written HIP kernel through compare_gan_amd.hip; a tensor on the "meta" device only builds the
graph (variables, shapes) -- there is no CPU arithmetic path.

Fusion device: `Act(x, slope)` is a *pending* leaky-ReLU.  Convolutions, linear layers and the
spatial reductions consume it as an input gate of their kernel instead of materialising act(x).
"""
import contextlib
import math

import torch

from compare_gan_amd import gin
from compare_gan_amd.gans import consts
from compare_gan_amd.hip import functional as Fn
from compare_gan_amd.hip import kernels as K
from compare_gan_amd.tpu import tpu_ops

BF16 = torch.bfloat16
F32 = torch.float32


# ------------------------------------------------------------------------------------------------
# variables
# ------------------------------------------------------------------------------------------------
class VariableStore(object):
  """name -> fp32 device tensor; the checkpoint naming contract of SURVEY.md App. D."""

  def __init__(self, device, seed=0):
    self.device = torch.device(device)
    self.seed = seed
    self.vars = {}            # insertion-ordered
    self.trainable = set()
    self.initializers = {}    # name -> kind (for the initializer tests)
    self._gen = torch.Generator().manual_seed(seed)
    self._scope = []
    # per top-level module ("generator" / "discriminator"), in call order:
    self.sn_registry = {}     # module -> {weight name: (u_var name, mode, eps)}
    self.conv_registry = {}   # module -> {weight name: None}
    self.sn_ready = {}        # weight name -> w / sigma of the current module call
    self.bt_ready = {}        # weight name -> (bt_fwd, bt_bwd) of the current module call

  # -- scopes --
  @contextlib.contextmanager
  def scope(self, name):
    self._scope.append(name)
    try:
      yield
    finally:
      self._scope.pop()

  def full_name(self, name):
    return "/".join(self._scope + [name]) if name else "/".join(self._scope)

  # -- access --
  def get(self, name, shape, initializer, trainable=True, dtype=F32):
    full = self.full_name(name)
    if full not in self.vars:
      kind, fn = initializer
      if self.device.type == "meta":
        t = torch.empty(tuple(shape), dtype=dtype, device="meta")
      else:
        t = fn(tuple(shape), self._gen).to(dtype).contiguous().to(self.device)
      if trainable:
        t.requires_grad_(True)
        self.trainable.add(full)
      self.vars[full] = t
      self.initializers[full] = kind
    v = self.vars[full]
    if tuple(v.shape) != tuple(shape):
      raise ValueError("Variable %s has shape %s but %s was requested." % (
          full, tuple(v.shape), tuple(shape)))
    return v

  def trainable_variables(self, scope_substr=None):
    """abstract_arch.py:43-45: trainable variables whose name contains the module name."""
    return [(n, v) for n, v in self.vars.items()
            if n in self.trainable and (scope_substr is None or scope_substr in n)]

  def global_variables(self):
    return list(self.vars.items())

  def state_dict(self):
    return {n: v.detach() for n, v in self.vars.items()}

  def load_state_dict(self, sd, strict=True):
    with torch.no_grad():
      for n, v in sd.items():
        if n in self.vars:
          self.vars[n].copy_(v.to(self.vars[n].dtype))
        elif strict:
          raise KeyError("unexpected variable %s" % n)
    # the accumulator switch of accumulator-mode batch norm (arch_ops.py:136-147 `update_accus`) is
    # persisted as a variable but read from its host mirror on the hot path: a checkpoint saved
    # with the switch on must turn the mirror on too (one device read per load, none per call)
    switches = [v for n, v in self.vars.items() if "accu/update_accus" in n and n in sd]
    if switches and not switches[0].is_meta:
      self.accu_fill = bool(max(float(v.max()) for v in switches) > 0)

  def set_accu_fill(self, on):
    """Turns the filling of the batch-norm accumulators on / off: the persisted `update_accus`
    variables (checkpoint contract) and their host mirror move together."""
    with torch.no_grad():
      for n, v in self.vars.items():
        if "accu/update_accus" in n:
          v.fill_(1 if on else 0)
    self.accu_fill = bool(on)


_STORE = [None]


def _store_cell():
  """The one-element list holding the active store; a thread attached to an in-process replica
  set (tpu_ops.InProcessReplicas) has its own."""
  ts = tpu_ops.thread_state()
  return _STORE if ts is None else ts.setdefault("store", [None])


@contextlib.contextmanager
def use_store(store):
  cell = _store_cell()
  old = cell[0]
  cell[0] = store
  try:
    yield store
  finally:
    cell[0] = old


def current_store():
  store = _store_cell()[0]
  if store is None:
    raise RuntimeError("No VariableStore is active (wrap the call in arch_ops.use_store(...)).")
  return store


def variable_scope(name):
  return current_store().scope(name)


def get_variable(name, shape, initializer, trainable=True, dtype=F32):
  return current_store().get(name, shape, initializer, trainable, dtype)


# ------------------------------------------------------------------------------------------------
# initialisers (arch_ops.py:46-63)
# ------------------------------------------------------------------------------------------------
def _normal(stddev):
  return ("random_normal", lambda s, g: torch.randn(s, generator=g, dtype=torch.float32) * stddev)


def _truncated(stddev):
  def fn(s, g):
    t = torch.randn(s, generator=g, dtype=torch.float32)
    bad = t.abs() > 2
    while bool(bad.any()):
      t[bad] = torch.randn(int(bad.sum()), generator=g, dtype=torch.float32)
      bad = t.abs() > 2
    return t * stddev
  return ("truncated_normal", fn)


def _orthogonal():
  def fn(s, g):
    rows = 1
    for d in s[:-1]:
      rows *= d
    cols = s[-1]
    a = torch.randn((max(rows, cols), min(rows, cols)), generator=g, dtype=torch.float64)
    q, r = torch.linalg.qr(a)
    q = q * torch.sign(torch.diagonal(r))
    if rows < cols:
      q = q.t()
    return q.reshape(s).to(torch.float32)
  return ("orthogonal", fn)


def glorot_normal():
  def fn(s, g):
    sd = math.sqrt(2.0 / (s[0] + s[1])) / 0.87962566103423978
    return torch.randn(s, generator=g, dtype=torch.float32).clamp_(-2, 2) * sd
  return ("glorot_normal", fn)


def constant(value):
  kind = "zeros" if value == 0 else ("ones" if value == 1 else "constant")
  return (kind, lambda s, g: torch.full(s, float(value), dtype=torch.float32))


@gin.configurable("weights")
def weight_initializer(initializer=consts.NORMAL_INIT, stddev=0.02):
  """Returns the (kind, fn) initializer for the given name (arch_ops.py:46-63)."""
  if initializer == consts.NORMAL_INIT:
    return _normal(stddev)
  if initializer == consts.TRUNCATED_INIT:
    return _truncated(stddev)
  if initializer == consts.ORTHOGONAL_INIT:
    return _orthogonal()
  raise ValueError("Unknown weight initializer {}.".format(initializer))


# ------------------------------------------------------------------------------------------------
# pending activations
# ------------------------------------------------------------------------------------------------
class Act(object):
  """x with a not-yet-applied leaky-ReLU of `slope` (0 = ReLU)."""

  def __init__(self, x, slope):
    self.x, self.slope = x, float(slope)

  @property
  def shape(self):
    return self.x.shape

  def materialize(self):
    if self.x.is_meta:
      return self.x
    return Fn.LreluFn.apply(self.x, self.slope)


class PendingBN(object):
  """relu(batch_norm(x)) whose normalisation has not been applied yet: the statistics exist, the
  consumer convolution applies (x - mean) * rstd * gamma + beta -> ReLU to its staged input tile in
  LDS (cg_gconv_fused), so the normalised activation never goes through HBM.  Only built for
  forward passes without an autograd graph (the generator forward of the discriminator sub-steps,
  modular_gan.py:465-467); any consumer that cannot fuse it calls materialize()."""

  def __init__(self, x, mean, var, gamma, beta, eps, per_sample):
    self.x, self.mean, self.var = x, mean, var
    self.gamma, self.beta, self.eps, self.per_sample = gamma, beta, eps, per_sample

  @property
  def shape(self):
    return self.x.shape

  def bn_tuple(self):
    g = None if self.gamma is None else self.gamma.contiguous()
    b = None if self.beta is None else self.beta.contiguous()
    return (self.mean, self.var, g, b, self.eps, self.per_sample)

  def materialize(self):
    n, c = self.x.shape[0], self.x.shape[-1]
    y3 = K.bn_apply(self.x.contiguous().reshape(n, -1, c), self.mean, self.var, self.eps,
                    self.gamma, self.beta, self.per_sample, True)
    return y3.reshape(self.x.shape)


# set by AbstractGenerator.__call__: convolutions of a no-gradient training-mode forward pass emit
# the partial sums of their output for the batch norm that follows (cg_gconv_fused)
_EMIT_BN_STATS = [False]
# > 1 inside statistics_groups(): the batch of the current no-gradient forward pass is `groups`
# consecutive sub-batches that stand for separate network calls, each with its own batch statistics
_BN_GROUPS = [1]


class statistics_groups(object):
  """Context for ONE batched generator call that replaces `groups` calls on the same weights (the
  generator forwards of the discriminator sub-steps, modular_gan.py:464-467): every batch norm
  computes its training statistics per group of consecutive samples -- the arithmetic of the
  separate calls -- and updates the moving averages once per group, in order."""

  def __init__(self, groups):
    self.groups = int(groups)

  def __enter__(self):
    self._old = _BN_GROUPS[0]
    _BN_GROUPS[0] = self.groups

  def __exit__(self, *exc):
    _BN_GROUPS[0] = self._old


import os as _os
_FUSED_BN = _os.environ.get("CGAMD_FUSED_BN", "1") != "0"   # A/B switch (read once)
_FUSED_POOL = _os.environ.get("CGAMD_FUSED_POOL", "1") != "0"
_PAD_ATTENTION = _os.environ.get("CGAMD_PAD_ATTENTION", "1") != "0"


def _split_act(inputs):
  if isinstance(inputs, PendingBN):
    return inputs.materialize(), None
  if isinstance(inputs, Act):
    return inputs.x, inputs.slope
  return inputs, None


def fork(x):
  """Two aliases of a block input that feeds both the shortcut and the main branch: their gradient
  contributions are summed by a HIP launch (Fn.ForkFn).  Pending activations are materialised."""
  return Fn.fork(as_tensor(x))


def relu(x):
  """tf.nn.relu as a pending activation (resnet_ops.py:165,175)."""
  return Act(x, 0.0)


def lrelu(inputs, leak=0.2, name="lrelu"):
  """Leaky-ReLU as a pending activation (arch_ops.py:595-597)."""
  del name
  return Act(inputs, leak)


def as_tensor(x):
  return x.materialize() if isinstance(x, (Act, PendingBN)) else x


def _to_bf16(x):
  """fp32 host-side inputs (z, one-hot y products) enter the bf16 domain through a cast kernel."""
  if x.dtype == BF16:
    return x
  if x.is_meta:
    return torch.empty(x.shape, dtype=BF16, device="meta")
  return Fn.CastFn.apply(x)


# ------------------------------------------------------------------------------------------------
# spectral norm (arch_ops.py:453-535)
# ------------------------------------------------------------------------------------------------
@gin.configurable(blacklist=["inputs", "var_name", "build_only"])
def spectral_norm(inputs, epsilon=1e-12, singular_value="left", var_name=None, build_only=False):
  """Performs Spectral Normalization on a weight tensor; one power-iteration round per call, the
  persisted vector "<weight>/u_var" is updated in place.  `var_name` is the weight's variable name
  relative to the current scope (TF derives it from inputs.name, arch_ops.py:487-488)."""
  if inputs.dim() < 2:
    raise ValueError("Spectral norm can only be applied to multi-dimensional tensors")
  k, co = inputs.numel() // inputs.shape[-1], inputs.shape[-1]
  if singular_value == "auto":
    singular_value = "left" if k <= co else "right"
  if singular_value not in ("left", "right"):
    raise ValueError("Unknown singular_value {}.".format(singular_value))
  mode = 0 if singular_value == "left" else 1
  u_shape = (k, 1) if mode == 0 else (1, co)
  u_var = get_variable((var_name or "kernel") + "/u_var", u_shape, _normal(1.0), trainable=False)
  store = current_store()
  wname = store.full_name(var_name or "kernel")
  module = wname.split("/", 1)[0]
  store.sn_registry.setdefault(module, {})[wname] = (wname + "/u_var", mode, epsilon)
  if inputs.is_meta or build_only:
    return inputs   # graph construction only: variables exist, no power iteration is run
  ready = store.sn_ready.pop(wname, None)
  if ready is not None:
    return ready     # computed by prepare_module() together with the module's other weights
  return Fn.spectral_norm(inputs, u_var, mode, epsilon)


def prepare_module(module):
  """Batched per-call weight bookkeeping of one network ("generator" / "discriminator"), run at the
  start of its __call__ once its variables are known (any earlier call, including the shape-only
  build pass, registers them): one multi-tensor power iteration for every spectrally-normalised
  weight (the reference runs one per weight per call, arch_ops.py:479-535 -- same arithmetic, u is
  still updated on every call) and one multi-tensor fp32 -> bf16 operand preparation."""
  store = current_store()
  store.sn_ready, store.bt_ready = {}, {}
  if store.device.type == "meta":
    return
  # whoever updates this network's variables on another stream (the data-parallel optimiser on its
  # communication stream, modular_gan._OptimizerState) registers a join here: EVERY reader of the
  # variables goes through this function, so no call site can forget the dependency edge
  hook = getattr(store, "before_call", {}).get(module)
  if hook is not None:
    hook()
  sn = store.sn_registry.get(module, {})
  names = [n for n in sn if n in store.vars]
  if names:
    weights = [store.vars[n] for n in names]
    u_vars = [store.vars[sn[n][0]] for n in names]
    modes = [sn[n][1] for n in names]
    wbars = Fn.spectral_norm_batch(weights, u_vars, modes, sn[names[0]][2])
    for n, wb in zip(names, wbars):
      store.sn_ready[n] = wb
  conv = [n for n in store.conv_registry.get(module, {}) if n in store.vars]
  if conv:
    eff = [store.sn_ready.get(n, store.vars[n]).detach() for n in conv]
    w4 = [w if w.dim() == 4 else w.reshape(1, 1, w.shape[0], w.shape[1]) for w in eff]
    bt_f, bt_b = K.weight_prep_multi(w4, want_fwd=True, want_bwd=torch.is_grad_enabled())
    for n, f, b in zip(conv, bt_f, bt_b):
      store.bt_ready[n] = (f, b)


# ------------------------------------------------------------------------------------------------
# linear / conv2d / deconv2d (arch_ops.py:538-592)
# ------------------------------------------------------------------------------------------------
def _conv_call(x, slope, w, bias, spec_geom, transpose, residual, out_f32, dx_f32, pending_bn=None,
               pool=False, padded=False):
  """Runs inside the weight's variable scope: the kernel variable is <scope>/kernel.
  pending_bn: the PendingBN whose tensor `x` is (conv2d only).  pool: 2x2 average pooling fused
  behind the convolution (residual at the pooled size; callers check conv_pool_supported)."""
  spec = Fn.ConvSpec(spec_geom, transpose=transpose, slope_in=slope, out_f32=out_f32)
  store = current_store()
  wname = store.full_name("kernel")
  store.conv_registry.setdefault(wname.split("/", 1)[0], {})[wname] = None
  if x.is_meta:
    shape = spec.out_shape
    if pool:
      shape = (shape[0], shape[1] // 2, shape[2] // 2, shape[3])
    return torch.empty(shape, dtype=F32 if out_f32 else BF16, device="meta")
  bt_pair = store.bt_ready.pop(wname, None)
  if padded:
    bt_pair = None     # the operand images prepared per module have the variable's own shape
  if pool:
    if pending_bn is not None:
      x = pending_bn.materialize()
    gate = x.detach() if slope is not None else None
    return Fn.conv_pool(x, w, bias, residual, gate, spec, dx_f32, bt_pair)
  if (_FUSED_BN and not torch.is_grad_enabled() and not transpose and slope is None and
      bt_pair is not None and (pending_bn is not None or _EMIT_BN_STATS[0]) and
      (K.gconv_fused_rows(spec_geom) > 0 or
       (pending_bn is not None and residual is None and   # (RGB outputs: the prologue only)
        K.gconv_fused_prologue_supported(spec_geom)))):
    # no autograd graph: batch norm fused around the convolution (cg_gconv_fused)
    out, partials = K.gconv_fused(
        spec_geom, x.contiguous(), bt_pair[0], bias=bias, residual=residual, out_f32=out_f32,
        bn=None if pending_bn is None else pending_bn.bn_tuple(),
        want_stats=_EMIT_BN_STATS[0] and not out_f32 and K.gconv_fused_rows(spec_geom) > 0)
    if partials is not None:
      out._cg_bn_partials = (partials, spec_geom.N * spec_geom.Ho * spec_geom.Wo,   # pylint: disable=protected-access
                             K.gconv_fused_phases(spec_geom))
    return out
  if pending_bn is not None:
    x = pending_bn.materialize()
  gate = x.detach() if slope is not None else None
  return Fn.gconv(x, w, bias, residual, gate, None, spec, dx_f32, bt_pair)


def linear(inputs, output_size, scope=None, stddev=0.02, bias_start=0.0, use_sn=False,
           use_bias=True, out_f32=False, kernel_initializer=None):
  """Linear layer without the non-linear activation applied (arch_ops.py:538-556).
  inputs [B, K] (bf16, fp32 or a pending Act) -> [B, output_size] bf16 (fp32 if out_f32)."""
  x, slope = _split_act(inputs)
  if x.dim() != 2:
    raise ValueError("linear expects rank-2 inputs, got rank %d" % x.dim())
  x = _to_bf16(x)
  b, k = x.shape
  with variable_scope(scope or "linear"):
    kernel = get_variable("kernel", [k, output_size],
                          kernel_initializer or weight_initializer(stddev=stddev))
    if use_sn:
      kernel = spectral_norm(kernel, build_only=x.is_meta)
    bias = get_variable("bias", [output_size], constant(bias_start)) if use_bias else None
    geom = K.make_geom(b, 1, 1, k, 1, 1, output_size, 1, 1)
    w4 = kernel.reshape(1, 1, k, output_size)
    out = _conv_call(x.reshape(b, 1, 1, k), slope, w4, bias, geom, False, None, out_f32, False)
    return out.reshape(b, output_size)


def conv_pool_supported(inputs, output_dim, k_h, k_w):
  """True when conv2d(inputs, ..., pool=True) exists for this call site: unit-stride convolution
  + 2x2 average pooling in one kernel (cg_gconv_fused pool_out)."""
  if isinstance(inputs, PendingBN):
    x, slope = inputs.x, None
  elif isinstance(inputs, Act):
    x, slope = inputs.x, inputs.slope
  else:
    x, slope = inputs, None
  if x.is_meta or x.dim() != 4 or slope not in (None, 0.0) or not _FUSED_POOL:
    return False
  n, h, w_, ci = x.shape
  return K.gconv_pool_supported(K.geom_conv_same(n, h, w_, ci, output_dim, k_h, k_w, 1, 1))


def conv2d(inputs, output_dim, k_h, k_w, d_h, d_w, stddev=0.02, name="conv2d", use_sn=False,
           use_bias=True, upsample=False, residual=None, out_f32=False, dx_f32=False, pool=False,
           pad_out_to=None, logical_in=None, kernel_scale=None):
  """2-D convolution, TF 'SAME' padding (arch_ops.py:559-573).

  Extensions that keep the reference semantics but fuse its neighbours into the kernel:
  `inputs` may be a pending Act (input gate), `upsample=True` convolves the zero-inserted input of
  resnet_ops.unpool (resnet_ops.py:35-56,122-123) without materialising it, `residual` is added in
  the epilogue (resnet_ops.py:181)."""
  pending_bn = inputs if isinstance(inputs, PendingBN) else None
  x, slope = (inputs.x, None) if pending_bn is not None else _split_act(inputs)
  if x.dim() != 4:
    raise ValueError("conv2d expects NHWC inputs of rank 4, got rank %d" % x.dim())
  if d_h != d_w:
    raise ValueError("conv2d: only equal strides are supported (got %d, %d)" % (d_h, d_w))
  n, h, w_, ci = x.shape
  # Channel padding (the self-attention block's 12 / 24 / 48-channel projections): the VARIABLE
  # keeps the reference's shape [k_h, k_w, logical_in or ci, output_dim]; the kernel that runs sees
  # it zero-padded to `pad_out_to` output channels / to the `ci` (already padded) input channels,
  # so that the MFMA-tiled kernels apply (they slice K in 32-channel granules).  Zero rows /
  # columns change nothing in the arithmetic of the real channels.
  ci_var = ci if logical_in is None else int(logical_in)
  co_run = output_dim if pad_out_to is None else int(pad_out_to)
  resized = ci_var != ci or co_run != output_dim
  padded = resized and not x.is_meta
  with variable_scope(name):
    w = get_variable("kernel", [k_h, k_w, ci_var, output_dim], weight_initializer(stddev=stddev))
    if use_sn:
      w = spectral_norm(w, build_only=x.is_meta)
    bias = get_variable("bias", [output_dim], constant(0.0)) if use_bias else None
    if kernel_scale is not None and not x.is_meta:
      # a trainable scalar in front of the convolution (`x + sigma * conv(...)`) folded into the
      # kernel: Fn.ScaleWeightFn; the operand images prepare_module() made are of the unscaled kernel
      w = Fn.ScaleWeightFn.apply(w, kernel_scale)
      resized = True
    if padded:
      if bias is not None and co_run != output_dim:
        bias = torch.nn.functional.pad(bias, (0, co_run - output_dim))
      w = torch.nn.functional.pad(w, (0, co_run - output_dim, 0, ci - ci_var))   # data movement
    output_dim = co_run
    geom = K.geom_conv_same(n, h, w_, ci, output_dim, k_h, k_w, d_h, 2 if upsample else 1)
    # gradients w.r.t. image-like inputs (the network input) are kept in fp32: they feed the
    # gradient penalty's norm (penalty_lib.py:77-78) and the generator's output head
    if pool and (upsample or d_h != 1):
      raise ValueError("conv2d: pool=True needs a unit-stride convolution without upsampling")
    return _conv_call(x, slope, w, bias, geom, False, residual, out_f32, dx_f32 or ci <= 4,
                      pending_bn, pool, padded=resized)


def conv1x1(inputs, output_dim, **kwargs):
  return conv2d(inputs, output_dim, k_h=1, k_w=1, d_h=1, d_w=1, **kwargs)


def deconv2d(inputs, output_shape, k_h, k_w, d_h, d_w, stddev=0.02, name="deconv2d",
             use_sn=False, out_f32=False):
  """Transposed 2-D convolution, TF conv2d_transpose 'SAME' (arch_ops.py:579-592); kernel layout
  [k_h, k_w, C_out, C_in]."""
  x, slope = _split_act(inputs)
  if d_h != d_w:
    raise ValueError("deconv2d: only equal strides are supported")
  n, h, w_, ci = x.shape
  _, ho, wo, co = output_shape
  with variable_scope(name):
    w = get_variable("kernel", [k_h, k_w, co, ci], weight_initializer(stddev=stddev))
    if use_sn:
      w = spectral_norm(w, build_only=x.is_meta)
    bias = get_variable("bias", [co], constant(0.0))
    # forward conv F: [n,ho,wo,co] -> [n,h,w,ci]; the deconvolution is its adjoint
    geom = K.geom_conv_same(n, ho, wo, co, ci, k_h, k_w, d_h, 1)
    if (geom.Ho, geom.Wo) != (h, w_):
      raise ValueError("deconv2d: output_shape %s is inconsistent with input %s at stride %d" % (
          list(output_shape), list(x.shape), d_h))
    return _conv_call(x, slope, w, bias, geom, True, None, out_f32, False)


# ------------------------------------------------------------------------------------------------
# batch norm family (arch_ops.py:66-445)
# ------------------------------------------------------------------------------------------------
def _moments_for_inference(is_training, use_moving_averages, num_channels):
  """Creates the inference statistics; returns them.

  arch_ops.py:66-119 (moving averages, zero_debias=False): (moving_mean, moving_variance);
  arch_ops.py:122-191 (accumulators): (accu_mean, accu_variance, accu_counter, update_accus)."""
  del is_training
  c = [num_channels]
  if use_moving_averages:
    mm = get_variable("moving_mean", c, constant(0.0), trainable=False)
    mv = get_variable("moving_variance", c, constant(1.0), trainable=False)
    return mm, mv
  with variable_scope("accu"):
    accu_mean = get_variable("accu_mean", c, constant(0.0), trainable=False)
    accu_var = get_variable("accu_variance", c, constant(0.0), trainable=False)
    accu_counter = get_variable("accu_counter", [], constant(1e-12), trainable=False)
    update_accus = get_variable("update_accus", [], constant(0), trainable=False,
                                dtype=torch.int32)
  return accu_mean, accu_var, accu_counter, update_accus


@gin.configurable(whitelist=["decay", "epsilon", "use_cross_replica_mean", "use_moving_averages"])
def standardize_batch(inputs, is_training, decay=0.999, epsilon=1e-3, data_format="NHWC",
                      use_moving_averages=True, use_cross_replica_mean=None,
                      gamma=None, beta=None, per_sample=False, relu=False):
  """(x - mean) * rsqrt(var + eps) [* gamma + beta] [relu]  (arch_ops.py:194-319).

  The optional affine / activation arguments fuse the reference's follow-up ops
  (arch_ops.py:353-366,436-444; resnet_ops.py:165,175) into the same HIP kernel."""
  if data_format not in {"NCHW", "NHWC"}:
    raise ValueError("Invalid data_format {}. Allowed: NCHW, NHWC.".format(data_format))
  if data_format != "NHWC":
    raise ValueError("Only NHWC is supported by the HIP kernels.")
  inputs = as_tensor(inputs)
  if inputs.dim() not in (2, 4):
    raise ValueError("Inputs has unsupported rank. Expected 2 or 4 but got %d" % inputs.dim())
  if use_cross_replica_mean is None:
    use_cross_replica_mean = tpu_ops.in_replica_context()   # arch_ops.py:258-263
  num_channels = inputs.shape[-1]
  sync_fn = tpu_ops.SyncMoments() if use_cross_replica_mean else None
  stats = _moments_for_inference(is_training, use_moving_averages, num_channels)
  if inputs.is_meta:
    return inputs
  partials = getattr(inputs, "_cg_bn_partials", None)
  inputs = _to_bf16(inputs)
  if is_training:
    moving = (stats[0], stats[1], decay) if use_moving_averages else None
    if _BN_GROUPS[0] > 1:
      return _standardize_groups(inputs, partials, _BN_GROUPS[0], moving, sync_fn, gamma, beta,
                                 epsilon, per_sample, relu)
    if _FUSED_BN and not torch.is_grad_enabled() and relu and inputs.dim() == 4:
      # no autograd graph: statistics now (from the producer convolution's partial sums when it
      # emitted them), normalisation + ReLU inside the consumer convolution (PendingBN).  Under
      # data parallelism the local moments cross the replicas before they are used, and the
      # moving averages are updated from the global ones.
      mm, mv, dc = moving if moving is not None else (None, None, 0.0)
      if sync_fn is not None:
        mm_, mv_, dc_ = None, None, 0.0
      else:
        mm_, mv_, dc_ = mm, mv, dc
      if partials is not None:
        mean, var = K.bn_finalize(partials[0], partials[1], mm_, mv_, dc_)
      else:
        n, c = inputs.shape[0], inputs.shape[-1]
        mean, var = K.bn_stats(inputs.contiguous().reshape(n, -1, c), mm_, mv_, dc_)
      if sync_fn is not None:
        mean, var = sync_fn.forward_sync(mean, var)
        if mm is not None:
          K.bn_update_moving(mm, mv, mean, var, dc)
      return PendingBN(inputs, mean, var, gamma, beta, epsilon, per_sample)
    out, _, _ = Fn.batch_norm_act(inputs, gamma, beta, None, None, epsilon, per_sample, relu,
                                  sync_fn, moving)
    return out
  if use_moving_averages:
    mean, var = stats
  else:
    accu_mean, accu_var, accu_counter, update_accus = stats
    # the fill switch is mirrored on the host by eval_gan_lib._update_bn_accumulators (the
    # `update_accus` variable keeps the reference's name for checkpoints): no device read per call
    if getattr(current_store(), "accu_fill", False):      # eval_gan_lib.py:65-92 fills the accumulators
      n, c = inputs.shape[0], inputs.shape[-1]
      bmean, bvar = K.bn_stats(inputs.contiguous().reshape(n, -1, c))
      K.bn_accumulate(accu_mean, accu_var, accu_counter.reshape(1), bmean, bvar)
    mean, var = K.bn_accumulated_moments(accu_mean, accu_var, accu_counter.reshape(1))  # :191
  out, _, _ = Fn.batch_norm_act(inputs, gamma, beta, mean.contiguous(), var.contiguous(), epsilon,
                                per_sample, relu, None)
  return out


def _standardize_groups(inputs, partials, groups, moving, sync_fn, gamma, beta, epsilon,
                        per_sample, relu):
  """Training-mode standardize_batch of a batched no-gradient call (statistics_groups): moments
  [groups, C], one set per group of consecutive samples."""
  if torch.is_grad_enabled():
    raise RuntimeError("statistics groups exist for no-gradient forward passes only")
  n, c = inputs.shape[0], inputs.shape[-1]
  if n % groups:
    raise ValueError("batch of %d does not split into %d statistics groups" % (n, groups))
  mm, mv, dc = moving if moving is not None else (None, None, 0.0)
  local = sync_fn is None
  if partials is not None:
    mean, var = K.bn_finalize(partials[0], partials[1], mm if local else None,
                              mv if local else None, dc, groups=groups, phases=partials[2])
  else:
    mean, var = K.bn_stats(inputs.contiguous().reshape(n, -1, c), mm if local else None,
                           mv if local else None, dc, groups=groups)
  if sync_fn is not None:
    gm, gv = sync_fn.forward_sync(mean.reshape(-1), var.reshape(-1))
    mean, var = gm.reshape(groups, c), gv.reshape(groups, c)
    if mm is not None:
      for g in range(groups):      # one moving-average update per call the group stands for
        K.bn_update_moving(mm, mv, mean[g], var[g], dc)
  if _FUSED_BN and relu and inputs.dim() == 4:
    return PendingBN(inputs, mean, var, gamma, beta, epsilon, per_sample)
  y3 = K.bn_apply(inputs.contiguous().reshape(n, -1, c), mean, var, epsilon, gamma, beta,
                  per_sample, relu)
  return y3.reshape(inputs.shape)


@gin.configurable(blacklist=["inputs"])
def no_batch_norm(inputs):
  return inputs


@gin.configurable(blacklist=["inputs", "is_training", "center", "scale", "name", "relu"])
def batch_norm(inputs, is_training, center=True, scale=True, name="batch_norm", relu=False):
  """Vanilla batch normalization with trainable scale and offset (arch_ops.py:327-367)."""
  with variable_scope(name):
    c = inputs.shape[-1]
    gamma = get_variable("gamma", [c], constant(1.0)) if scale else None
    beta = get_variable("beta", [c], constant(0.0)) if center else None
    return standardize_batch(inputs, is_training=is_training, gamma=gamma, beta=beta, relu=relu)


@gin.configurable(whitelist=["num_hidden"])
def self_modulated_batch_norm(inputs, z, is_training, use_sn, center=True, scale=True,
                              name="batch_norm", num_hidden=32, relu=False):
  """Self-modulated batch normalization (arch_ops.py:370-420)."""
  if z is None:
    raise ValueError("You must provide z for self modulation.")
  with variable_scope(name):
    c = inputs.shape[-1]
    gamma = beta = None
    with variable_scope("sbn"):
      h = z
      if num_hidden > 0:
        h = relu_act(linear(h, num_hidden, scope="hidden", use_sn=use_sn))
      if scale:
        gamma = linear(h, c, scope="gamma", bias_start=1.0, use_sn=use_sn, out_f32=True)
      if center:
        beta = linear(h, c, scope="beta", use_sn=use_sn, out_f32=True)
    return standardize_batch(inputs, is_training=is_training, gamma=gamma, beta=beta,
                             per_sample=True, relu=relu)


def relu_act(x):
  return Act(x, 0.0)


@gin.configurable(whitelist=["use_bias"])
def conditional_batch_norm(inputs, y, is_training, use_sn, center=True, scale=True,
                           name="batch_norm", use_bias=False, relu=False):
  """Conditional batch normalization (arch_ops.py:423-445): gamma = linear(y), beta = linear(y)."""
  if y is None:
    raise ValueError("You must provide y for conditional batch normalization.")
  if y.dim() != 2:
    raise ValueError("Conditioning must have rank 2.")
  with variable_scope(name):
    c = inputs.shape[-1]
    gamma = beta = None
    with variable_scope("condition"):
      if scale:
        gamma = linear(y, c, scope="gamma", use_sn=use_sn, use_bias=use_bias, out_f32=True)
      if center:
        beta = linear(y, c, scope="beta", use_sn=use_sn, use_bias=use_bias, out_f32=True)
    return standardize_batch(inputs, is_training=is_training, gamma=gamma, beta=beta,
                             per_sample=True, relu=relu)


def layer_norm(input_, is_training, scope):
  """tf.contrib.layers.layer_norm(input_, trainable=is_training, scope=scope) (arch_ops.py:448-450)
  with the library's defaults: statistics per sample over (H, W, C) (begin_norm_axis = 1),
  `beta` (zeros) and `gamma` (ones) per channel (begin_params_axis = -1), variance_epsilon 1e-12;
  the variables are <scope>/beta and <scope>/gamma, trainable iff is_training."""
  input_ = as_tensor(input_)
  with variable_scope(scope):
    c = input_.shape[-1]
    beta = get_variable("beta", [c], constant(0.0), trainable=bool(is_training))
    gamma = get_variable("gamma", [c], constant(1.0), trainable=bool(is_training))
  if input_.is_meta:
    return input_
  return Fn.layer_norm(input_, gamma, beta, 1e-12)


# ------------------------------------------------------------------------------------------------
# pooling / reductions / heads
# ------------------------------------------------------------------------------------------------
def avg_pool2(x):
  """tf.nn.pool(x, [2,2], 'AVG', 'SAME', strides=[2,2]) (resnet_ops.py:131-133)."""
  x = as_tensor(x)
  n, h, w, c = x.shape
  if h == 1 and w == 1:
    # 'SAME' pooling of a 1x1 map is the map itself (output size ceil(1/2) = 1, the average runs
    # over the valid element only): resnet5's sixth down block on 32x32 inputs (ssgan_test.py:35)
    return x
  if h % 2 or w % 2:
    raise ValueError("avg_pool2: odd map sizes other than 1x1 are not supported (got %dx%d)" % (h, w))
  if x.is_meta:
    return torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device="meta")
  return Fn.avg_pool2(x)


def unpool(x, residual=None):
  """resnet_ops.unpool (resnet_ops.py:35-56) on its own: zero-insertion 2x upsampling, with the
  block's `outputs += shortcut` fused in as `residual` (resnet_biggan_deep.py:102-103,188)."""
  x = as_tensor(x)
  if x.is_meta:
    n, h, w, c = x.shape
    return torch.empty((n, 2 * h, 2 * w, c), dtype=x.dtype, device="meta")
  return Fn.unpool2(x, None if residual is None else as_tensor(residual))


def max_pool2(x):
  """tf.layers.max_pooling2d(pool_size=[2,2], strides=2) (arch_ops.py:741,750)."""
  if x.is_meta:
    n, h, w, c = x.shape
    return torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device="meta")
  return Fn.max_pool2(x)


def reduce_spatial(inputs, mean):
  """tf.reduce_mean / tf.reduce_sum over axes [1, 2] of a (possibly pending-ReLU) NHWC tensor
  (resnet_cifar.py:154-156, resnet_biggan.py:404-405) -> [N, C] bf16."""
  x, slope = _split_act(inputs)
  if slope not in (None, 0.0):
    x, slope = Act(x, slope).materialize(), None
  n, c = x.shape[0], x.shape[-1]
  if x.is_meta:
    return torch.empty((n, c), dtype=BF16, device="meta")
  hw = x.numel() // (n * c)
  gate = x.detach() if slope is not None else None
  return Fn.SpatialReduceFn.apply(x, gate, (1.0 / hw) if mean else 1.0)


_FOLD_SIGMA = _os.environ.get("CGAMD_FOLD_SIGMA", "1") != "0"    # A/B switch (read once)
# the four consumers of the self-attention block's input summed by one launch (A/B switch)
_FORK_ATTENTION = _os.environ.get("CGAMD_FORK_ATTENTION", "1") != "0"
_DOUBLE_BWD = [False]
_FUSED_HEAD = _os.environ.get("CGAMD_FUSED_HEAD", "1") != "0"    # A/B switch (read once)


class twice_differentiable(object):
  """Context for network calls whose gradient will itself be differentiated (the discriminator call
  of a gradient penalty, penalty_lib.py:59-82): fused ops without a second derivative fall back
  to their separate, twice differentiable launches."""

  def __enter__(self):
    self._old = _DOUBLE_BWD[0]
    _DOUBLE_BWD[0] = True

  def __exit__(self, *exc):
    _DOUBLE_BWD[0] = self._old


def pooled_linear_head(inputs, mean, scope, use_sn=False):
  """relu -> reduce_mean / reduce_sum over [1, 2] -> linear(C -> 1): the tail of every ResNet
  discriminator (resnet_cifar.py:154-157, resnet5.py:141-145, resnet_stl.py, resnet_biggan.py:404-407).
  inputs: ops.relu(net).  Returns (out_logit [N,1] fp32, h [N,C] bf16), both differentiable.
  One fused launch per direction where it exists (Fn.PooledHeadFn); the reference's two ops
  otherwise (gradient penalties, leaky gates, channel counts that are not multiples of 8)."""
  x, slope = _split_act(inputs)
  n, c = x.shape[0], x.shape[-1]
  hw = x.numel() // max(1, n * c)
  if (x.is_meta or not _FUSED_HEAD or _DOUBLE_BWD[0] or slope != 0.0 or x.dtype != BF16 or
      not K.pooled_head_supported(hw, c)):
    h = reduce_spatial(inputs, mean=mean)
    return linear(h, 1, scope=scope, use_sn=use_sn, out_f32=True), h
  with variable_scope(scope):
    kernel = get_variable("kernel", [c, 1], weight_initializer(stddev=0.02))
    if use_sn:
      kernel = spectral_norm(kernel)
    bias = get_variable("bias", [1], constant(0.0))
    store = current_store()
    store.bt_ready.pop(store.full_name("kernel"), None)   # (prepare_module()'s operand image is not needed)
    return Fn.PooledHeadFn.apply(x, kernel, bias, (1.0 / hw) if mean else 1.0)


def output_head(pre_activation, kind):
  """sigmoid (kind 0) or (tanh + 1) / 2 (kind 1) on the fp32 pre-activation."""
  if pre_activation.is_meta:
    return pre_activation
  return Fn.HeadFn.apply(pre_activation, kind)


# ------------------------------------------------------------------------------------------------
# self-attention (arch_ops.py:709-758)
# ------------------------------------------------------------------------------------------------
def non_local_block(x, name, use_sn):
  """SAGAN self-attention: theta/phi/g 1x1 convs, 2x2 max-pooled keys/values, fused
  softmax(theta phi^T) g, 1x1 back-projection, x + sigma * o."""
  with variable_scope(name):
    n, h, w, c = x.shape
    ca, cg = c // 8, c // 2
    # the projections run padded to multiples of 32 channels (zero kernel columns): at ch = 96 the
    # discriminator's block has 12 / 48 of them, which would leave its six 1x1 convolutions and their
    # gradients to the generic gather kernel at 3-25 TFLOP/s (7 % of the BigGAN-128 step)
    cap, cgp = (ca + 31) // 32 * 32, (cg + 31) // 32 * 32
    if not _PAD_ATTENTION or c % 32:
      cap, cgp = ca, cg
    # x has four consumers: their gradient contributions are summed by one launch (Fn.ForkNFn)
    x_t, x_p, x_g, x = Fn.fork_n(x, 4) if _FORK_ATTENTION else (x, x, x, x)
    theta = conv1x1(x_t, ca, name="conv2d_theta", use_sn=use_sn, use_bias=False, pad_out_to=cap)
    phi = max_pool2(conv1x1(x_p, ca, name="conv2d_phi", use_sn=use_sn, use_bias=False,
                            pad_out_to=cap))
    g = max_pool2(conv1x1(x_g, cg, name="conv2d_g", use_sn=use_sn, use_bias=False, pad_out_to=cgp))
    if x.is_meta:
      attn_g = torch.empty((n, h, w, cgp), dtype=BF16, device="meta")
    else:
      attn_g = Fn.AttentionFn.apply(theta.reshape(n, h * w, cap), phi.reshape(n, h * w // 4, cap),
                                    g.reshape(n, h * w // 4, cgp)).reshape(n, h, w, cgp)
    sigma = get_variable("sigma", [], constant(0.0))
    if x.is_meta or not _FOLD_SIGMA:
      attn_g = conv1x1(attn_g, c, name="conv2d_attn_g", use_sn=use_sn, use_bias=False,
                       logical_in=cg)
      if x.is_meta:
        return x
      return Fn.ScaledResidualFn.apply(x, attn_g, sigma.reshape(1))
    # x + sigma * conv(attn_g, w) as ONE convolution: sigma folded into the kernel, x the residual of
    # its epilogue (three passes over the [N, 64, 64, C] map less per call: the scaled add, its
    # gradient sigma * dy, and the dot product for d sigma)
    return conv1x1(attn_g, c, name="conv2d_attn_g", use_sn=use_sn, use_bias=False, logical_in=cg,
                   kernel_scale=sigma, residual=x.contiguous())
