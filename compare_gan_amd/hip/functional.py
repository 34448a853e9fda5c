"""torch.autograd plumbing around the HIP kernels: every arithmetic step (forward AND backward) is a
libcgamd.so launch; autograd only records the graph.  Backward passes of the convolution family
are themselves expressed through these Functions, so torch.autograd.grad(create_graph=True)
(WGAN-GP, penalty_lib.py:74-82) differentiates through the first backward with HIP kernels too.

Conventions: activations NHWC bf16; weights fp32 in the reference layouts; fp32 for logits,
losses, CBN gamma/beta and generator outputs.
"""
import os

import torch

from compare_gan_amd.hip import kernels as K

BF16 = torch.bfloat16
F32 = torch.float32


def _bf16(t):
  """Gradient tensors may arrive as fp32 (fp32-output ops) or non-contiguous views."""
  if t is None:
    return None
  if t.dtype == F32:
    if torch.is_grad_enabled() and t.requires_grad:
      # inside a create_graph=True backward (gradient penalties) the gradient is itself part of a
      # graph: the raw cast kernel has no autograd node and would cut the second-order path through
      # it silently -- CastFn is the same kernel with a straight-through gradient
      return CastFn.apply(t)
    return K.cast_f32_to_bf16(t.contiguous())
  return t.contiguous()


class ConvSpec(object):
  """Static description of one gather-convolution call site.

  geom      : cgConvGeom of the FORWARD convolution F (input [N,Hin,Win,Ci] -> [N,Ho,Wo,Co],
              kernel HWIO [kh,kw,Ci,Co]).
  transpose : False -> y = F(x);  True -> y = F^T(x) (conv2d_transpose / data gradient; x lives in
              F's output space).
  slope_in / slope_out : leaky-ReLU slopes of the input / output gates (None = no gate).
  out_f32   : fp32 output (logits, images, gradients w.r.t. network inputs).
  """

  def __init__(self, geom, transpose=False, slope_in=None, slope_out=None, out_f32=False):
    self.geom = geom
    self.transpose = transpose
    self.slope_in = slope_in
    self.slope_out = slope_out
    self.out_f32 = out_f32

  def adjoint(self, out_f32=False):
    return ConvSpec(self.geom, not self.transpose, self.slope_out, self.slope_in, out_f32)

  @property
  def out_shape(self):
    g = self.geom
    return (g.N, g.Hin, g.Win, g.Ci) if self.transpose else (g.N, g.Ho, g.Wo, g.Co)


def _run_gconv(spec, x, w, bias, gate_in, gate_out, residual, bt_pair=None):
  """bt_pair: (bt_fwd, bt_bwd) operand images prepared by the module-level batch, or None."""
  g = spec.geom
  if spec.transpose:
    bt = bt_pair[1] if bt_pair is not None else None
    if bt is None:
      _, bt = K.weight_prep(w, want_fwd=False, want_bwd=True)
    geom = K.geom_adjoint(g)
  else:
    bt = bt_pair[0] if bt_pair is not None else None
    if bt is None:
      bt, _ = K.weight_prep(w, want_fwd=True, want_bwd=False)
    geom = g
  return K.gconv(geom, x, bt, bias=bias,
                 gate_in=gate_in if spec.slope_in is not None else None,
                 slope_in=spec.slope_in or 0.0,
                 gate_out=gate_out if spec.slope_out is not None else None,
                 slope_out=spec.slope_out or 0.0, residual=residual, out_f32=spec.out_f32)


_SKIP_PARAM_GRADS = [False]

# ------------------------------------------------------------------------------------------------
# weight gradients on a second HIP stream
# ------------------------------------------------------------------------------------------------
# In the backward pass of a convolution the data gradient (needed by the next layer down) and the
# weight gradient (needed only by the optimiser) are independent.  The small layers of the GAN
# discriminators / generators do not fill 256 CUs on their own, so the weight gradients run on a
# side stream and overlap with the data-gradient chain; inside a captured step this becomes a
# parallel branch of the hipGraph.  Consumers of weight gradients (spectral-norm backward, the
# optimiser) call join_wgrad_stream() first.
_WGRAD = {"enabled": False, "stream": None, "dirty": False}


def enable_wgrad_stream(enabled=True):
  _WGRAD["enabled"] = bool(enabled)


def _wgrad_side_stream():
  if _WGRAD["stream"] is None:
    _WGRAD["stream"] = torch.cuda.Stream()
  return _WGRAD["stream"]


# ------------------------------------------------------------------------------------------------
# deferred weight gradients
# ------------------------------------------------------------------------------------------------
# tf.gradients(loss, var_list) hands the optimiser all kernel gradients of a network at once
# (modular_gan.py:480-483,494-497): nothing in the data-gradient chain reads them.  The weight
# gradients of the small feature maps (4x4 / 8x8 blocks) cannot fill 256 CUs one at a time, so
# inside a deferred_wgrads() context GConvFn.backward only RECORDS them -- it returns tensors whose
# contents are not written yet -- and flush_wgrads() runs all of them in one cg_gwgrad_multi call
# (several layers per launch, no split partials).  Readers flush first: the spectral-norm
# backward, the optimiser (through join_wgrad_stream()) and the end of the context.  The caller
# guarantees that no weight receives a second gradient contribution inside the context (autograd
# would add to the unwritten tensor): modular_gan enables it for single-call discriminator /
# generator graphs without penalties only.
# The weight gradients that do run at once inside the context (the large maps: one launch fills the
# chip) still leave a small fixed-order reduction of their per-split partials behind; those are
# recorded in a K.DeferCtx -- a caller-owned cgDeferCtx, one per device, owned HERE and handed to
# cg_gwgrad_deferred / cg_gwgrad_pooled_deferred with every call (the library keeps no deferral
# state of its own) -- and run in ONE launch at the same flush points.
_DEFER = {"on": False, "reduce": False, "jobs": [], "wptrs": set(), "ctx": {}}
_DEFER_REDUCE = os.environ.get("CGAMD_DEFER_REDUCE", "1") != "0"   # A/B switch (read once)


def _reduce_ctx(t):
  """The DeferCtx that records the split reductions of a weight gradient on t's device, or None
  when they run at once."""
  if not (_DEFER["on"] and _DEFER["reduce"]) or not t.is_cuda:
    return None
  idx = t.device.index if t.device.index is not None else torch.cuda.current_device()
  ctx = _DEFER["ctx"].get(idx)
  if ctx is None:
    ctx = _DEFER["ctx"][idx] = K.DeferCtx()
  return ctx


def _flush_reductions():
  for idx, ctx in list(_DEFER["ctx"].items()):
    if ctx.keep:
      with torch.cuda.device(idx):
        ctx.flush()


def _abort_reductions():
  for ctx in _DEFER["ctx"].values():
    ctx.abort()


class deferred_wgrads(object):
  def __init__(self, enabled=True):
    self._enabled = bool(enabled)

  def __enter__(self):
    self._old = (_DEFER["on"], _DEFER["reduce"])
    if self._old[0] and not self._enabled:
      flush_wgrads()    # an inner scope that computes at once must not meet unwritten gradients
    _DEFER["on"] = self._enabled
    # (not together with the side-stream weight gradients: the recorded reductions are launched on
    # the stream that is current at the flush, the partials would be written on the side stream)
    _DEFER["reduce"] = self._enabled and _DEFER_REDUCE and not _WGRAD["enabled"]
    return self

  def __exit__(self, etype, *exc):
    if etype is None:
      flush_wgrads()
    else:
      del _DEFER["jobs"][:]
      _DEFER["wptrs"].clear()
      _abort_reductions()
    _DEFER["on"], _DEFER["reduce"] = self._old


def flush_wgrads():
  """Runs every recorded weight gradient and every recorded reduction (their output tensors are
  valid afterwards).  Inside a context, recording continues behind the flush."""
  if _DEFER["jobs"]:
    jobs, _DEFER["jobs"] = _DEFER["jobs"], []
    K.gwgrad_multi(jobs)
  _DEFER["wptrs"].clear()
  _flush_reductions()


def join_wgrad_stream():
  """Makes the current stream wait for every weight gradient launched on the side stream, and
  runs the deferred ones."""
  flush_wgrads()
  if _WGRAD["dirty"]:
    torch.cuda.current_stream().wait_stream(_WGRAD["stream"])
    _WGRAD["dirty"] = False


class only_input_grads(object):
  """Context for torch.autograd.grad(outputs, [network input], create_graph=True): the weight /
  bias gradients of that inner backward are discarded by the engine, so do not compute them."""

  def __enter__(self):
    self._old = _SKIP_PARAM_GRADS[0]
    _SKIP_PARAM_GRADS[0] = True

  def __exit__(self, *exc):
    _SKIP_PARAM_GRADS[0] = self._old


class GConvFn(torch.autograd.Function):
  """y = D(gate_out) * (conv_spec(D(gate_in) * x, w) + bias) + residual.

  gate tensors are constants for autograd (the derivative of a piecewise-linear activation is
  piecewise constant); gate_in may be x itself (y = conv(lrelu(x)))."""

  @staticmethod
  def forward(ctx, x, w, bias, residual, gate_in, gate_out, spec, dx_f32, bt_pair=None):
    x = x.contiguous()
    y = _run_gconv(spec, x, w, bias, gate_in, gate_out, residual, bt_pair)
    ctx.spec, ctx.dx_f32 = spec, dx_f32
    ctx.has_bias, ctx.has_res = bias is not None, residual is not None
    # the operand images of THIS call travel with the node: a later call of the same module
    # (gradient penalty, next sub-step) prepares new ones
    ctx.bt_pair = bt_pair
    ctx.w_late = _weight_grad_read_late(w)
    ctx.save_for_backward(x, w, gate_in, gate_out)
    return y

  @staticmethod
  def backward(ctx, dy):
    x, w, gate_in, gate_out = ctx.saved_tensors
    spec = ctx.spec
    dy16 = _bf16(dy)
    need_x, need_w, need_b, need_r = ctx.needs_input_grad[:4]
    if _SKIP_PARAM_GRADS[0]:
      need_w = need_b = False
    dx = dw = db = dr = None
    if need_r and ctx.has_res:
      dr = dy16
    if need_x:
      aspec = spec.adjoint(out_f32=ctx.dx_f32)
      dx = GConvFn.apply(dy16, w, None, None, gate_out, gate_in, aspec, False, ctx.bt_pair)
    if need_w or (need_b and ctx.has_bias):
      want_b = bool(need_b and ctx.has_bias)
      if torch.is_grad_enabled():
        dw = GWgradFn.apply(x, dy16, gate_in, gate_out, spec) if need_w else None
        if want_b:
          db = K.colsum(_gated(dy16, gate_out, spec.slope_out).reshape(-1, dy16.shape[-1]))
      elif _WGRAD["enabled"] and x.is_cuda:
        main, side = torch.cuda.current_stream(), _wgrad_side_stream()
        side.wait_stream(main)     # x / dy16 / gates are ready on the main stream
        with torch.cuda.stream(side):
          dw, db = _run_wgrad(spec, x, dy16, gate_in, gate_out, want_b)
        for t in (x, dy16, gate_in, gate_out):
          if t is not None:
            t.record_stream(side)    # keep their memory until the side stream has read it
        for t in (dw, db):
          if t is not None:
            t.record_stream(main)    # allocated on the side stream, consumed on the main one
        _WGRAD["dirty"] = True
      elif (_DEFER["on"] and ctx.w_late and need_w and not spec.transpose and x.is_cuda and
            (gate_out is None or spec.slope_out is None) and
            (gate_in is None or spec.slope_in is None or
             (spec.slope_in == 0.0 and gate_in.data_ptr() == x.data_ptr())) and
            K.gwgrad_groupable(spec.geom)):
        g = spec.geom
        if w.data_ptr() in _DEFER["wptrs"]:
          # a second contribution to a weight whose first one is still only recorded: autograd is
          # about to ADD the two tensors, so the recorded ones are written now and this one is
          # computed on the spot (the callers' static check makes this path unreachable
          # for the example configs)
          flush_wgrads()
          dw, db = _run_wgrad(spec, x, dy16, gate_in, gate_out, want_b)
          flush_wgrads()   # (its split reduction is recorded too)
        else:
          _DEFER["wptrs"].add(w.data_ptr())
          dw = torch.empty((g.kh, g.kw, g.Ci, g.Co), dtype=F32, device=x.device)
          db = torch.empty((g.Co,), dtype=F32, device=x.device) if want_b else None
          relu_in = gate_in is not None and spec.slope_in is not None
          _DEFER["jobs"].append((g, x, dy16, relu_in, dw, db))
      else:
        dup = _note_weight_contribution(w)
        dw, db = _run_wgrad(spec, x, dy16, gate_in, gate_out, want_b)
        if dup or not ctx.w_late:
          flush_wgrads()
    return dx, dw, db, dr, None, None, None, None, None


_VIEW_NODES = ("ViewBackward0", "ReshapeAliasBackward0", "UnsafeViewBackward0", "AliasBackward0")


def _weight_grad_read_late(w):
  """True when nothing reads the gradient of the kernel tensor `w` before a flush point: `w` is a
  variable (its gradient goes to the optimiser), a view of one, or the output of the spectral-norm
  Functions (whose backward flushes first).  Anything else -- e.g. the zero-padding of the
  self-attention projections, whose backward is a torch slice of dw -- reads it inside the backward
  pass, so such a gradient must be complete when GConvFn.backward returns."""
  fn = w.grad_fn
  for _ in range(4):
    if fn is None:
      return True
    name = type(fn).__name__
    if name == "AccumulateGrad" or name.startswith("SpectralNorm"):
      return True
    if name in _VIEW_NODES and fn.next_functions:
      fn = fn.next_functions[0][0]
      continue
    return False
  return False


def _note_weight_contribution(w):
  """Inside a deferred_wgrads() context a weight gradient computed now is only valid after the flush
  (its split reduction is recorded, not run): a SECOND contribution to the same weight would be
  added by autograd to an unwritten tensor, so everything recorded is written first.
  The caller flushes again behind the second computation when this returns True."""
  if _DEFER["on"] and _DEFER["reduce"]:
    if w.data_ptr() in _DEFER["wptrs"]:
      flush_wgrads()
      return True
    _DEFER["wptrs"].add(w.data_ptr())
  return False


def _gated(t, gate, slope):
  if gate is None or slope is None:
    return t
  return K.lrelu_bwd(gate, t, slope)


def _run_wgrad(spec, x, dy16, gate_in, gate_out, want_b):
  """dw (+ dbias) for y = conv_spec(D(gate_in) x, w): roles swap for the transposed form."""
  g = spec.geom
  gi = gate_in if spec.slope_in is not None else None
  go = gate_out if spec.slope_out is not None else None
  if not spec.transpose:
    dw, db = K.gwgrad(g, x, dy16, gate_in=gi, slope_in=spec.slope_in or 0.0, gate_dy=go,
                      slope_dy=spec.slope_out or 0.0, want_dbias=want_b, defer=_reduce_ctx(x))
    return dw, db
  # y = F^T(x): dw = Wg_F(in = D(go) dy, "dy" = D(gi) x); dbias = colsum over y's pixels
  dw, _ = K.gwgrad(g, dy16, x, gate_in=go, slope_in=spec.slope_out or 0.0, gate_dy=gi,
                   slope_dy=spec.slope_in or 0.0, defer=_reduce_ctx(x))
  db = None
  if want_b:
    db = K.colsum(_gated(dy16, go, spec.slope_out).reshape(-1, dy16.shape[-1]))
  return dw, db


class GWgradFn(torch.autograd.Function):
  """dw = Wg_spec(x, dy) as a differentiable node (only built under create_graph=True)."""

  @staticmethod
  def forward(ctx, x, dy16, gate_in, gate_out, spec):
    dw, _ = _run_wgrad(spec, x, dy16, gate_in, gate_out, False)
    ctx.spec = spec
    ctx.save_for_backward(x, dy16, gate_in, gate_out)
    return dw

  @staticmethod
  def backward(ctx, ddw):
    x, dy16, gate_in, gate_out = ctx.saved_tensors
    spec = ctx.spec
    ddw = ddw.contiguous()
    dx = ddy = None
    if ctx.needs_input_grad[0]:
      # d<ddw, Wg(x,dy)>/dx = D(gi) * conv_spec^T(D(go) dy, ddw)
      dx = GConvFn.apply(dy16, ddw, None, None, gate_out, gate_in, spec.adjoint(), False)
    if ctx.needs_input_grad[1]:
      # d/d(dy) = D(go) * conv_spec(D(gi) x, ddw)
      fwd = ConvSpec(spec.geom, spec.transpose, spec.slope_in, spec.slope_out, False)
      ddy = GConvFn.apply(x, ddw, None, None, gate_in, gate_out, fwd, False)
    return dx, ddy, None, None, None


def gconv(x, w, bias=None, residual=None, gate_in=None, gate_out=None, spec=None, dx_f32=False,
          bt_pair=None):
  return GConvFn.apply(x, w, bias, residual, gate_in, gate_out, spec, dx_f32, bt_pair)


class ConvPoolFn(torch.autograd.Function):
  """y = avgpool2x2(conv_spec(relu?(x), w) + bias) + residual_p: a discriminator block's
  convolution with the down-sampling that follows it (resnet_ops.py:131-133) in ONE kernel -- the
  full-resolution output never goes through HBM.  residual_p lives at the pooled resolution.

  Backward: the pooling's gradient is the nearest-neighbour up-sampling of dy times 1/4; the
  data-gradient and weight-gradient kernels read the pooled-resolution dy directly (in_up /
  gwgrad_pooled).  Under create_graph (WGAN-GP) the backward is composed from the differentiable
  Functions instead."""

  @staticmethod
  def forward(ctx, x, w, bias, residual_p, gate_in, spec, dx_f32, bt_pair=None):
    x = x.contiguous()
    bt = bt_pair[0] if bt_pair is not None else None
    if bt is None:
      bt, _ = K.weight_prep(w, want_fwd=True, want_bwd=False)
    gi = gate_in if spec.slope_in is not None else None
    y, _ = K.gconv_fused(spec.geom, x, bt, bias=bias, residual=residual_p, gate_in=gi, pool=True,
                         out_f32=spec.out_f32)
    ctx.spec, ctx.dx_f32, ctx.bt_pair = spec, dx_f32, bt_pair
    ctx.has_bias, ctx.has_res = bias is not None, residual_p is not None
    ctx.w_late = _weight_grad_read_late(w)
    ctx.save_for_backward(x, w, gate_in)
    return y

  @staticmethod
  def backward(ctx, dy):
    x, w, gate_in = ctx.saved_tensors
    spec = ctx.spec
    dy16 = _bf16(dy)
    need_x, need_w, need_b, need_r = ctx.needs_input_grad[:4]
    if _SKIP_PARAM_GRADS[0]:
      need_w = need_b = False
    dx = dw = db = None
    dr = dy16 if (need_r and ctx.has_res) else None
    want_b = bool(need_b and ctx.has_bias)
    if torch.is_grad_enabled():
      # differentiable composition (second-order terms of a gradient penalty)
      dyf = AvgPool2BwdFn.apply(dy16)
      if need_x:
        dx = GConvFn.apply(dyf, w, None, None, None, gate_in, spec.adjoint(out_f32=ctx.dx_f32),
                           False, ctx.bt_pair)
      if need_w:
        dw = GWgradFn.apply(x, dyf, gate_in, None, spec)
      if want_b:
        db = K.colsum(dyf.reshape(-1, dyf.shape[-1]))
      return dx, dw, db, dr, None, None, None, None
    g = spec.geom
    gi = gate_in if spec.slope_in is not None else None
    if need_x:
      ag = K.geom_adjoint(g)
      bt_b = ctx.bt_pair[1] if ctx.bt_pair is not None else None
      if bt_b is None:
        _, bt_b = K.weight_prep(w, want_fwd=False, want_bwd=True)
      if K.gconv_fused_rows(ag) > 0:
        dx, _ = K.gconv_fused(ag, dy16, bt_b, gate_out=gi, slope_out=spec.slope_in or 0.0,
                              in_up=True, out_scale=0.25, out_f32=ctx.dx_f32)
      else:
        dx = K.gconv(ag, K.avgpool2_bwd(dy16), bt_b, gate_out=gi, slope_out=spec.slope_in or 0.0,
                     out_f32=ctx.dx_f32)
    if need_w or want_b:
      dup = _note_weight_contribution(w)
      dw, db = K.gwgrad_pooled(g, x, dy16, gate_in=gi, want_dbias=want_b, defer=_reduce_ctx(x))
      if dup or not ctx.w_late:
        flush_wgrads()
      if not need_w:
        dw = None
    return dx, dw, db, dr, None, None, None, None


def conv_pool(x, w, bias=None, residual_p=None, gate_in=None, spec=None, dx_f32=False,
              bt_pair=None):
  return ConvPoolFn.apply(x, w, bias, residual_p, gate_in, spec, dx_f32, bt_pair)


# ------------------------------------------------------------------------------------------------
# spectral norm
# ------------------------------------------------------------------------------------------------
class SpectralNormFn(torch.autograd.Function):
  """w_bar = w / sigma after one power iteration; updates u_var in place (arch_ops.py:479-535).

  First-order only: the gradient of the rank-one correction itself is not differentiated (no
  example config combines a gradient penalty with a spectrally normalised D)."""

  @staticmethod
  def forward(ctx, w, u_var, mode, eps):
    w2 = w.reshape(-1, w.shape[-1])
    v, sigma, inv_sigma = K.spectral_norm(w2, u_var.view(-1), mode, eps)
    wbar = K.scale_f32(w2, inv_sigma).reshape(w.shape)
    ctx.mode = mode
    ctx.save_for_backward(w2, u_var.detach().view(-1).clone(), v, sigma)
    return wbar

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, dwbar):
    join_wgrad_stream()
    w2, u_new, v, sigma = ctx.saved_tensors
    a_k, b_co = (u_new, v) if ctx.mode == 0 else (v, u_new)
    dw = K.sn_backward(dwbar.contiguous().reshape(w2.shape), w2, a_k, b_co, sigma)
    return dw.reshape(dwbar.shape), None, None, None


def spectral_norm(w, u_var, mode, eps):
  return SpectralNormFn.apply(w, u_var, mode, eps)


class SpectralNormBatchFn(torch.autograd.Function):
  """SpectralNormFn for ALL spectrally-normalised weights of a network call at once: five launches
  forward (two mat-vec passes, their normalisations, the 1/sigma scaling), two backward."""

  @staticmethod
  def forward(ctx, u_vars, modes, eps, *weights):
    w2 = [w.reshape(-1, w.shape[-1]) for w in weights]
    u_news, vs, sigmas, wbars = K.spectral_norm_multi(w2, [u.view(-1) for u in u_vars], modes, eps)
    ctx.modes = list(modes)
    ctx.n = len(weights)
    ctx.save_for_backward(*(w2 + u_news + vs + sigmas))
    ctx.set_materialize_grads(False)
    return tuple(wb.view(w.shape) for wb, w in zip(wbars, weights))

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, *dwbars):
    join_wgrad_stream()
    n = ctx.n
    saved = ctx.saved_tensors
    w2, u_news, vs, sigmas = saved[:n], saved[n:2 * n], saved[2 * n:3 * n], saved[3 * n:4 * n]
    idx = [i for i in range(n) if dwbars[i] is not None and ctx.needs_input_grad[3 + i]]
    out = [None] * n
    if idx:
      a_ks = [u_news[i] if ctx.modes[i] == 0 else vs[i] for i in idx]
      b_cos = [vs[i] if ctx.modes[i] == 0 else u_news[i] for i in idx]
      dws = K.sn_backward_multi([dwbars[i].contiguous().reshape(w2[i].shape) for i in idx],
                                [w2[i] for i in idx], a_ks, b_cos, [sigmas[i] for i in idx])
      for i, dw in zip(idx, dws):
        out[i] = dw.reshape(dwbars[i].shape)
    return (None, None, None) + tuple(out)


def spectral_norm_batch(weights, u_vars, modes, eps):
  return SpectralNormBatchFn.apply(tuple(u_vars), tuple(modes), eps, *weights)


# ------------------------------------------------------------------------------------------------
# batch norm (+ ReLU)
# ------------------------------------------------------------------------------------------------
class BatchNormActFn(torch.autograd.Function):
  """y = act((x - mean) * rsqrt(var + eps) * gamma + beta).

  training: mean/var are this batch's moments (returned for the moving-average / accumulator
  bookkeeping); eval: the given moments are used as constants."""

  @staticmethod
  def forward(ctx, x, gamma, beta, mean_in, var_in, eps, per_sample, relu, sync_fn, moving):
    shape = x.shape
    N, C = shape[0], shape[-1]
    x3 = x.contiguous().reshape(N, -1, C)
    batch_stats = mean_in is None
    if batch_stats:
      if sync_fn is None and moving is not None:
        # moving averages updated by the statistics kernel itself (arch_ops.py:105-114)
        mean, var = K.bn_stats(x3, moving[0], moving[1], moving[2])
      else:
        mean, var = K.bn_stats(x3)
        if sync_fn is not None:
          mean, var = sync_fn.forward_sync(mean, var)
        if moving is not None:
          K.bn_update_moving(moving[0], moving[1], mean, var, moving[2])
    else:
      mean, var = mean_in, var_in
    y3 = K.bn_apply(x3, mean, var, eps, gamma, beta, per_sample, relu)
    ctx.cfg = (eps, per_sample, relu, batch_stats, shape, sync_fn)
    ctx.save_for_backward(x3, y3, gamma, mean, var)
    ctx.mark_non_differentiable(mean, var)
    ctx.set_materialize_grads(False)   # no zero-fill launches for the moments' unused gradients
    return y3.reshape(shape), mean, var

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, dy, _dm, _dv):
    if dy is None:
      return (None,) * 10
    x3, y3, gamma, mean, var = ctx.saved_tensors
    eps, per_sample, relu, batch_stats, shape, sync_fn = ctx.cfg
    dy3 = _bf16(dy).reshape(x3.shape)
    dx, dgamma, dbeta = K.bn_backward(
        x3, y3, dy3, mean, var, eps, gamma, per_sample, relu, batch_stats,
        want_dgamma=ctx.needs_input_grad[1], want_dbeta=ctx.needs_input_grad[2],
        sync_fn=sync_fn.backward_sync if sync_fn is not None else None)
    return dx.reshape(shape), dgamma, dbeta, None, None, None, None, None, None, None


def batch_norm_act(x, gamma, beta, mean=None, var=None, eps=1e-5, per_sample=False, relu=False,
                   sync_fn=None, moving=None):
  """moving = (moving_mean, moving_var, decay): updated in place from this batch's moments."""
  return BatchNormActFn.apply(x, gamma, beta, mean, var, eps, per_sample, relu, sync_fn, moving)


class LayerNormBwdFn(torch.autograd.Function):
  """dx (and dgamma, dbeta) of layer_norm as a differentiable node: only built under
  create_graph=True, i.e. by a gradient penalty through D.layer_norm = True (resnet_ops.py:162-173
  under penalty_lib.py:59-82).  Its backward is cg_layer_norm_bwd_bwd (closed form in cg_ln.hip)."""

  @staticmethod
  def forward(ctx, x3, dy3, mean, rstd, gamma, want_params):
    dx, dg, db = K.layer_norm_bwd(x3, dy3, mean, rstd, gamma, want_params=want_params)
    ctx.save_for_backward(x3, dy3, mean, rstd, gamma)
    ctx.want_params = want_params
    if not want_params:
      dg = db = torch.zeros((gamma.numel(),), dtype=gamma.dtype, device=gamma.device)
    ctx.mark_non_differentiable(dg, db)   # (the penalty's inner backward discards them)
    return dx, dg, db

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, u, _ug, _ub):
    x3, dy3, mean, rstd, gamma = ctx.saved_tensors
    d_dy, d_x, d_g = K.layer_norm_bwd_bwd(x3, dy3, _bf16(u).reshape(x3.shape), mean, rstd, gamma,
                                          want_dgamma=ctx.needs_input_grad[4])
    return d_x, d_dy, None, None, d_g, None


class LayerNormFn(torch.autograd.Function):
  """tf.contrib.layers.layer_norm (arch_ops.py:448-450): statistics per sample over (H, W, C),
  gamma / beta per channel.  Twice differentiable with respect to its input (LayerNormBwdFn): the
  WGAN-GP discriminator normaliser."""

  @staticmethod
  def forward(ctx, x, gamma, beta, eps):
    shape = x.shape
    x3 = x.contiguous().reshape(shape[0], -1, shape[-1])
    y3, mean, rstd = K.layer_norm_fwd(x3, gamma.contiguous(), beta.contiguous(), eps)
    ctx.shape = shape
    # x itself (an input of the node), not the reshaped copy: under create_graph=True the second-order
    # gradient with respect to x has to flow on into the layers below
    ctx.save_for_backward(x, mean, rstd, gamma)
    return y3.reshape(shape)

  @staticmethod
  def backward(ctx, dy):
    x, mean, rstd, gamma = ctx.saved_tensors
    x3 = x.contiguous().reshape(ctx.shape[0], -1, ctx.shape[-1])
    want = bool((ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and not _SKIP_PARAM_GRADS[0])
    if torch.is_grad_enabled():
      dx, dg, db = LayerNormBwdFn.apply(x3, _bf16(dy).reshape(x3.shape), mean, rstd,
                                        gamma.contiguous(), want)
    else:
      dx, dg, db = K.layer_norm_bwd(x3, _bf16(dy).reshape(x3.shape), mean, rstd, gamma.contiguous(),
                                    want_params=want)
    if not want:
      dg = db = None
    return (dx.reshape(ctx.shape), dg if ctx.needs_input_grad[1] else None,
            db if ctx.needs_input_grad[2] else None, None)


def layer_norm(x, gamma, beta, eps=1e-12):
  return LayerNormFn.apply(x, gamma, beta, eps)


# ------------------------------------------------------------------------------------------------
# stand-alone leaky ReLU (only where the consumer cannot take an input gate)
# ------------------------------------------------------------------------------------------------
class LreluFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, slope):
    x = x.contiguous()
    ctx.slope = slope
    ctx.save_for_backward(x)
    return K.lrelu(x, slope)

  @staticmethod
  def backward(ctx, dy):
    x, = ctx.saved_tensors
    return GateFn.apply(_bf16(dy), x, ctx.slope), None


class GateFn(torch.autograd.Function):
  """t * D(gate): self-adjoint in t."""

  @staticmethod
  def forward(ctx, t, gate, slope):
    ctx.slope = slope
    ctx.save_for_backward(gate)
    return K.lrelu_bwd(gate, t.contiguous(), slope)

  @staticmethod
  def backward(ctx, d):
    gate, = ctx.saved_tensors
    return GateFn.apply(_bf16(d), gate, ctx.slope), None, None


# ------------------------------------------------------------------------------------------------
# pooling / reductions
# ------------------------------------------------------------------------------------------------
class AvgPool2Fn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x):
    return K.avgpool2(x.contiguous())

  @staticmethod
  def backward(ctx, dy):
    return AvgPool2BwdFn.apply(_bf16(dy))


class AvgPool2BwdFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, dy):
    return K.avgpool2_bwd(dy.contiguous())

  @staticmethod
  def backward(ctx, ddx):
    return AvgPool2Fn.apply(_bf16(ddx))


class UnpoolFn(torch.autograd.Function):
  """Zero-insertion 2x upsampling (+ residual): resnet_ops.unpool outside a convolution."""

  @staticmethod
  def forward(ctx, x, residual):
    ctx.has_res = residual is not None
    return K.unpool2(x.contiguous(), None if residual is None else residual.contiguous())

  @staticmethod
  def backward(ctx, dy):
    dy16 = _bf16(dy)
    return UnpoolBwdFn.apply(dy16), (dy16 if ctx.has_res else None)


class UnpoolBwdFn(torch.autograd.Function):
  """dx = dy[:, ::2, ::2, :]; its own gradient is the upsampling again."""

  @staticmethod
  def forward(ctx, dy):
    return K.unpool2_bwd(dy.contiguous())

  @staticmethod
  def backward(ctx, ddx):
    return UnpoolFn.apply(_bf16(ddx), None)


def unpool2(x, residual=None):
  return UnpoolFn.apply(x, residual)


class MaxPool2Fn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x):
    x = x.contiguous()
    ctx.save_for_backward(x)
    return K.maxpool2(x)

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, dy):
    x, = ctx.saved_tensors
    return K.maxpool2_bwd(x, _bf16(dy))


class SpatialReduceFn(torch.autograd.Function):
  """out[n,c] = scale * sum_hw x * D(gate), gate constant (gate = x gives relu + reduce)."""

  @staticmethod
  def forward(ctx, x, gate, scale):
    x = x.contiguous()
    ctx.scale, ctx.shape = scale, x.shape
    ctx.save_for_backward(gate)
    return K.spatial_reduce(x, gate, scale)

  @staticmethod
  def backward(ctx, dout):
    gate, = ctx.saved_tensors
    return SpatialReduceBwdFn.apply(_bf16(dout), gate, ctx.scale, ctx.shape), None, None


class SpatialReduceBwdFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, dout, gate, scale, shape):
    ctx.scale = scale
    ctx.save_for_backward(gate)
    return K.spatial_reduce_bwd(gate, dout.contiguous(), shape, scale)

  @staticmethod
  def backward(ctx, ddx):
    gate, = ctx.saved_tensors
    return SpatialReduceFn.apply(_bf16(ddx), gate, ctx.scale), None, None, None


class PooledHeadFn(torch.autograd.Function):
  """(logit [N,1] fp32, pooled [N,C] bf16) = head(x [N,...,C] bf16, w [C,1] fp32, bias [1] fp32):
  relu -> scale * sum over the spatial axes -> linear(C -> 1), one launch forward, one (+ a small
  reduction) backward (cg_pooled_head_*; resnet_cifar.py:154-157, arch_ops.py:538-556).  `pooled`
  is differentiable too (projection discriminators and the self-supervised heads read it).  Not
  twice differentiable: gradient penalties call the discriminator inside ops.twice_differentiable(),
  where the separate (differentiable) launches run instead."""

  @staticmethod
  def forward(ctx, x, w, bias, scale):
    n, c = x.shape[0], x.shape[-1]
    x3 = x.contiguous().reshape(n, -1, c)
    wv = w.contiguous().reshape(-1)
    logit, pooled = K.pooled_head_fwd(x3, wv, bias, scale)
    ctx.scale, ctx.shape = scale, x.shape
    ctx.save_for_backward(x3, wv, pooled)
    ctx.w_shape = w.shape
    ctx.has_bias = bias is not None
    ctx.set_materialize_grads(False)
    return logit, pooled

  @staticmethod
  def backward(ctx, dlogit, dpooled):
    if torch.is_grad_enabled():
      raise RuntimeError("PooledHeadFn is not twice differentiable: call the discriminator inside "
                         "ops.twice_differentiable() when its gradient will be differentiated again")
    x3, wv, pooled = ctx.saved_tensors
    if dlogit is None and dpooled is None:
      return None, None, None, None
    dl = dlogit.contiguous().reshape(-1).float() if dlogit is not None else None
    dp = _bf16(dpooled) if dpooled is not None else None
    need_w = ctx.needs_input_grad[1] and not _SKIP_PARAM_GRADS[0]
    need_b = ctx.has_bias and ctx.needs_input_grad[2] and not _SKIP_PARAM_GRADS[0]
    dx, dw, db = K.pooled_head_bwd(x3, wv, ctx.scale, pooled, dlogit=dl, dpooled=dp,
                                   want_dw=need_w, want_dbias=need_b)
    return (dx.reshape(ctx.shape) if ctx.needs_input_grad[0] else None,
            dw.reshape(ctx.w_shape) if dw is not None else None, db, None)


def avg_pool2(x):
  return AvgPool2Fn.apply(x)


def max_pool2(x):
  return MaxPool2Fn.apply(x)


def relu_mean(x):
  hw = x.numel() // (x.shape[0] * x.shape[-1])
  return SpatialReduceFn.apply(x, x.detach(), 1.0 / hw)


def relu_sum(x):
  return SpatialReduceFn.apply(x, x.detach(), 1.0)


# ------------------------------------------------------------------------------------------------
# heads, input staging, projection
# ------------------------------------------------------------------------------------------------
class HeadFn(torch.autograd.Function):
  """fp32 pre-activation -> image in [0,1]: kind 0 sigmoid, 1 (tanh+1)/2."""

  @staticmethod
  def forward(ctx, x, kind):
    y = K.head(x.contiguous(), kind)
    ctx.kind = kind
    ctx.save_for_backward(y)
    return y

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, dy):
    y, = ctx.saved_tensors
    return K.head_bwd(y, ctx.kind, dy.contiguous()), None


class StageImagesFn(torch.autograd.Function):
  """all_images = concat([images, generated]) * a + b, cast to bf16 (modular_gan.py:657;
  sndcgan.py:108 folds its `x * 2 - 1` here).  Either part may be None."""

  @staticmethod
  def forward(ctx, real, fake, a, b):
    parts = [t for t in (real, fake) if t is not None]
    n = sum(t.shape[0] for t in parts)
    out = torch.empty((n,) + tuple(parts[0].shape[1:]), dtype=BF16, device=parts[0].device)
    off = 0
    for t in parts:
      K.cast_f32_to_bf16(t.contiguous(), a, b, out=out[off:off + t.shape[0]])
      off += t.shape[0]
    ctx.a = a
    ctx.n_real = real.shape[0] if real is not None else 0
    ctx.has_fake = fake is not None
    return out

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, dy):
    dfake = None
    if ctx.has_fake and ctx.needs_input_grad[1]:
      d = dy[ctx.n_real:].contiguous()
      if d.dtype == BF16:
        d = K.cast_bf16_to_f32(d)
      dfake = K.scale_f32(d, None, ctx.a) if ctx.a != 1.0 else d
    return None, dfake, None, None


def stage_images(real, fake, a=1.0, b=0.0):
  return StageImagesFn.apply(real, fake, a, b)


class CastFn(torch.autograd.Function):
  """fp32 -> bf16 (z, embeddings) with a straight-through fp32 gradient."""

  @staticmethod
  def forward(ctx, x):
    return K.cast_f32_to_bf16(x.contiguous())

  @staticmethod
  def backward(ctx, dy):
    return K.cast_bf16_to_f32(dy.contiguous()) if dy.dtype == BF16 else dy


class ToF32Fn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x):
    return K.cast_bf16_to_f32(x.contiguous())

  @staticmethod
  def backward(ctx, dy):
    return _bf16(dy)


class RowDotFn(torch.autograd.Function):
  """out[b] = sum_c a[b,c] * h[b,c] (projection term, resnet_biggan.py:423)."""

  @staticmethod
  def forward(ctx, a, h):
    a, h = a.contiguous(), h.contiguous()
    ctx.save_for_backward(a, h)
    return K.rowdot(a, h)

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, dout):
    a, h = ctx.saved_tensors
    da, dh = K.rowdot_bwd(a, h, dout.contiguous(), ctx.needs_input_grad[0],
                          ctx.needs_input_grad[1])
    return da, dh


class AddF32Fn(torch.autograd.Function):
  """alpha * a + beta * b on small fp32 tensors (logit + projection; loss + lambda * penalty)."""

  @staticmethod
  def forward(ctx, a, b, alpha, beta):
    ctx.ab = (alpha, beta)
    return K.axpby_f32(a.contiguous(), alpha, b.contiguous(), beta)

  @staticmethod
  def backward(ctx, d):
    alpha, beta = ctx.ab
    d = d.contiguous()
    da = d if alpha == 1.0 else K.axpby_f32(d, alpha)
    db = d if beta == 1.0 else K.axpby_f32(d, beta)
    return da, db, None, None


def add_f32(a, b, alpha=1.0, beta=1.0):
  return AddF32Fn.apply(a, b, alpha, beta)


class ScaleF32Fn(torch.autograd.Function):
  """s * x on fp32 (linear, hence differentiable to any order through itself)."""

  @staticmethod
  def forward(ctx, x, s):
    ctx.s = s
    return K.scale_f32(x.contiguous(), None, s)

  @staticmethod
  def backward(ctx, d):
    return ScaleF32Fn.apply(d.contiguous() if d.dtype == F32 else K.cast_bf16_to_f32(d.contiguous()),
                            ctx.s), None


class HalfSumSqFn(torch.autograd.Function):
  """tf.nn.l2_loss(w) = sum(w^2) / 2 of an fp32 tensor (penalty_lib.py:99-102)."""

  @staticmethod
  def forward(ctx, w):
    w = w.contiguous()
    ctx.save_for_backward(w)
    return ScaleF32Fn.apply(K.moments_f32(w)[1:2], 0.5).reshape(())

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, up):
    w, = ctx.saved_tensors
    return K.scale_f32(w, up.reshape(1).to(F32).contiguous()).reshape(w.shape)


class L2LossMeanFn(torch.autograd.Function):
  """mean_k tf.nn.l2_loss(w_k) = (1 / K) sum_k sum(w_k^2) / 2 over a LIST of fp32 tensors
  (penalty_lib.py:85-102) in three launches whatever K is: one multi-tensor gather into a flat
  buffer, one sum of squares, one scale; the backward is one scale of the flat buffer, handed back
  as per-tensor views."""

  @staticmethod
  def forward(ctx, *ws):
    ws = [w.contiguous() for w in ws]
    flat = torch.empty(sum(w.numel() for w in ws), dtype=F32, device=ws[0].device)
    K.flatten_multi(ws, flat)
    ctx.shapes = [w.shape for w in ws]
    ctx.save_for_backward(flat)
    return K.scale_f32(K.moments_f32(flat)[1:2].contiguous(), None, 0.5 / len(ws)).reshape(())

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, up):
    flat, = ctx.saved_tensors
    g = K.scale_f32(flat, up.reshape(1).to(F32).contiguous(), 1.0 / len(ctx.shapes))
    out, off = [], 0
    for shp in ctx.shapes:
      n = 1
      for d in shp:
        n *= d
      out.append(g[off:off + n].view(shp))
      off += n
    return tuple(out)


class AttentionFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, theta, phi, g):
    theta, phi, g = theta.contiguous(), phi.contiguous(), g.contiguous()
    out, lse = K.attention_fwd(theta, phi, g)
    ctx.save_for_backward(theta, phi, g, out, lse)
    return out

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, dout):
    theta, phi, g, out, lse = ctx.saved_tensors
    return K.attention_bwd(theta, phi, g, out, lse, _bf16(dout))


class ScaleWeightFn(torch.autograd.Function):
  """w_eff = sigma * w for a trainable device scalar sigma: `x + sigma * conv(a, w)`
  (arch_ops.py:755-758) runs as conv(a, sigma * w) with x as the convolution's residual -- no pass
  over the activations for the scaled add, none for its gradients: d w = sigma * d w_eff,
  d sigma = <d w_eff, w> (a dot over the weight, not over the feature map)."""

  @staticmethod
  def forward(ctx, w, sigma):
    w = w.contiguous()
    ctx.save_for_backward(w, sigma)
    return K.scale_f32(w, sigma.reshape(1).contiguous())

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, dweff):
    w, sigma = ctx.saved_tensors
    dweff = dweff.contiguous()
    dw = K.scale_f32(dweff, sigma.reshape(1).contiguous()) if ctx.needs_input_grad[0] else None
    dsig = K.dot_f32(dweff, w).reshape(sigma.shape) if ctx.needs_input_grad[1] else None
    return dw, dsig


class ScaledResidualFn(torch.autograd.Function):
  """x + sigma * o with a trainable scalar sigma (arch_ops.py:755-758)."""

  @staticmethod
  def forward(ctx, x, o, sigma):
    x, o = x.contiguous(), o.contiguous()
    ctx.save_for_backward(o, sigma)
    return K.axpy_dev(x, o, sigma)

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, dy):
    o, sigma = ctx.saved_tensors
    dy16 = _bf16(dy)
    do = K.axpy_dev(None, dy16, sigma) if ctx.needs_input_grad[1] else None
    dsig = K.dot_bf16(dy16, o).reshape(sigma.shape) if ctx.needs_input_grad[2] else None
    return dy16, do, dsig


# ------------------------------------------------------------------------------------------------
# losses / penalties
# ------------------------------------------------------------------------------------------------
class GanLossFn(torch.autograd.Function):
  """(d_loss, d_loss_real, d_loss_fake, g_loss) from fp32 logits [2B,1] (real then fake)."""

  @staticmethod
  def forward(ctx, logits, kind):
    losses, dd, dg = K.gan_loss(kind, logits.contiguous())
    ctx.save_for_backward(dd, dg)
    ctx.shape = logits.shape
    ctx.set_materialize_grads(False)   # only one of the four is differentiated per sub-step
    return tuple(losses.unbind(0))

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, g_d, g_real, g_fake, g_g):
    dd, dg = ctx.saved_tensors
    # only d_loss and g_loss are ever differentiated (modular_gan.py:480-497)
    out = None
    for up, base in ((g_d, dd), (g_g, dg)):
      if up is None:
        continue
      term = K.scale_f32(base, up.reshape(1).to(F32).contiguous())
      out = term if out is None else K.axpby_f32(out, 1.0, term, 1.0)
    return (out.reshape(ctx.shape) if out is not None else None), None


class SumFn(torch.autograd.Function):
  """a + b (bf16 or fp32, same shape) as a HIP launch; linear, so differentiable to any order."""

  @staticmethod
  def forward(ctx, a, b):
    a, b = a.contiguous(), b.contiguous()
    if a.dtype != b.dtype:
      a = a if a.dtype == F32 else K.cast_bf16_to_f32(a)
      b = b if b.dtype == F32 else K.cast_bf16_to_f32(b)
    return K.axpby_f32(a, 1.0, b, 1.0) if a.dtype == F32 else K.axpby(a, 1.0, b, 1.0)

  @staticmethod
  def backward(ctx, d):
    return d, d


class ForkFn(torch.autograd.Function):
  """A tensor with TWO consumers (a residual block's input feeds the shortcut and the main branch,
  resnet_ops.py:156-181): returns two aliases of it, and sums the two gradient contributions with
  a HIP launch -- otherwise autograd accumulates them with a torch add, the one piece of hot-path
  arithmetic that would run outside libcgamd.so."""

  @staticmethod
  def forward(ctx, x):
    ctx.set_materialize_grads(False)
    return x.view_as(x), x.view_as(x)

  @staticmethod
  def backward(ctx, g1, g2):
    if g1 is None or g2 is None:
      return g1 if g2 is None else g2
    if torch.is_grad_enabled():
      return SumFn.apply(g1, g2)
    with torch.no_grad():
      return SumFn.forward(None, g1, g2)


class Sum4Fn(torch.autograd.Function):
  """a + b + c (+ d) of bf16 tensors in one launch (cg_sum4); linear, differentiable to any order."""

  @staticmethod
  def forward(ctx, *ts):
    ctx.n = len(ts)
    return K.sum4(*[t.contiguous() for t in ts])

  @staticmethod
  def backward(ctx, d):
    return (d,) * ctx.n


class ForkNFn(torch.autograd.Function):
  """A tensor with THREE or FOUR consumers (the input of the self-attention block, arch_ops.py:709-758:
  theta / phi / g projections and the residual path): aliases forward, one fused sum of the gradient
  contributions backward (autograd would chain bf16 torch adds: 3 passes of 2 reads + 1 write over the
  [N, H, W, C] map -- 27 launches, 2.4 ms of the BigGAN bs-256 step, r05_torch_ops_per_step.txt)."""

  @staticmethod
  def forward(ctx, x, n):
    ctx.set_materialize_grads(False)
    return tuple(x.view_as(x) for _ in range(n))

  @staticmethod
  def backward(ctx, *gs):
    live = [g for g in gs if g is not None]
    if not live:
      return None, None
    if len(live) == 1:
      return live[0], None
    same = all(g.dtype == BF16 for g in live)
    if len(live) == 2 or not same:
      acc = live[0]
      for g in live[1:]:
        acc = SumFn.apply(acc, g) if torch.is_grad_enabled() else SumFn.forward(None, acc, g)
      return acc, None
    if torch.is_grad_enabled():
      return Sum4Fn.apply(*live), None
    with torch.no_grad():
      return K.sum4(*[g.contiguous() for g in live]), None


def fork_n(x, n):
  """n aliases of x for n (3 or 4) consumers when a gradient will flow back; plain aliases otherwise."""
  if torch.is_tensor(x) and x.requires_grad and torch.is_grad_enabled() and not x.is_meta:
    return ForkNFn.apply(x, n)
  return (x,) * n


def fork(x):
  """(x, x) for two consumers when a gradient will flow back; plain aliases otherwise."""
  if torch.is_tensor(x) and x.requires_grad and torch.is_grad_enabled() and not x.is_meta:
    return ForkFn.apply(x)
  return x, x


class SoftmaxXentEpsFn(torch.autograd.Function):
  """-mean_i log(softmax(logits_i)[label_i] + eps) (ssgan.py:191-199), first-order."""

  @staticmethod
  def forward(ctx, logits, labels, eps):
    loss, dlogits = K.softmax_xent_eps(logits.contiguous(), labels, eps)
    ctx.save_for_backward(dlogits)
    return loss.reshape(())

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, up):
    (dlogits,) = ctx.saved_tensors
    return K.scale_f32(dlogits, up.reshape(1).to(F32).contiguous()), None, None


class SoftmaxXentWeightedFn(torch.autograd.Function):
  """tf.losses.softmax_cross_entropy(labels, logits, weights) (s3gan.py:311-313), first-order;
  labels and weights are constants (the reference stops their gradients)."""

  @staticmethod
  def forward(ctx, logits, labels, weights):
    loss, dlogits = K.softmax_xent_weighted(logits.contiguous(), labels.contiguous(),
                                            weights.contiguous())
    ctx.save_for_backward(dlogits)
    return loss.reshape(())

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, up):
    (dlogits,) = ctx.saved_tensors
    return K.scale_f32(dlogits, up.reshape(1).to(F32).contiguous()), None, None


class GradientPenaltyFn(torch.autograd.Function):
  """mean((sqrt(1e-4 + sum g^2) - 1)^2) over fp32 input gradients g [B, ...]."""

  @staticmethod
  def forward(ctx, g):
    g = g.contiguous()
    slopes, pen = K.gradient_penalty(g)
    ctx.save_for_backward(g, slopes)
    return pen.reshape(())

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, up):
    g, slopes = ctx.saved_tensors
    return K.gradient_penalty_bwd(g, slopes, up.reshape(1).to(F32).contiguous())
