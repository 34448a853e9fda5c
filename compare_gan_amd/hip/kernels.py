"""Thin tensor-level wrappers over the C-ABI (include/cgamd.h): argument validation, output and
workspace allocation on the current torch HIP stream.  No arithmetic happens in Python/torch here;
torch only owns the device memory and the stream."""
import ctypes
import os

import torch

from compare_gan_amd.hip import _lib
from compare_gan_amd.hip._lib import ConvGeom, WgradItem, check

BF16 = torch.bfloat16
F32 = torch.float32
F64 = torch.float64


def lib():
    return _lib.load()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _req(t, dtype, name, allow_none=False):
    if t is None:
        if allow_none:
            return
        raise ValueError("%s must not be None" % name)
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU (no CPU fallback exists)" % name)
    if t.dtype != dtype:
        raise ValueError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)


def _ws(nbytes, like):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=like.device)


class DeferCtx(object):
    """A caller-owned cgDeferCtx (include/cgamd.h, "Deferred reductions") plus the workspaces whose
    split partials its recorded reductions will read.  gwgrad / gwgrad_pooled called with
    `defer=ctx` launch the partial-sum kernels and record the reduction; their outputs are valid
    after ctx.flush() (on the stream current there).  One context per device / stream / backward
    pass; the library itself holds no deferral state."""

    def __init__(self):
        self._lib = lib()
        self.handle = self._lib.cg_defer_create()
        if not self.handle:
            raise MemoryError("cg_defer_create failed")
        self.keep = []

    def pending(self):
        return int(self._lib.cg_defer_pending(self.handle))

    def flush(self):
        """Runs every recorded reduction in one launch per kernel form; the context is empty (and
        reusable) afterwards."""
        if self.keep or self.pending():
            check(self._lib.cg_defer_flush(self.handle, _stream()), "cg_defer_flush")
        del self.keep[:]   # (stream-ordered: a later allocation reuses them after the launch)

    def abort(self):
        del self.keep[:]
        self._lib.cg_defer_abort(self.handle)

    def __del__(self):
        try:
            if self.handle:
                self._lib.cg_defer_destroy(self.handle)
                self.handle = None
        except Exception:  # pylint: disable=broad-except
            pass


# ------------------------------------------------------------------------------------------------
# geometry helpers
# ------------------------------------------------------------------------------------------------
def make_geom(N, Hin, Win, Ci, Ho, Wo, Co, kh, kw, S=1, U=1, pt=0, pl=0):
    return ConvGeom(N, Hin, Win, Ci, Ho, Wo, Co, kh, kw, S, U, pt, pl)


def geom_conv_same(N, H, W, Ci, Co, kh, kw, stride=1, up=1):
    """tf.nn.conv2d(padding='SAME') on the zero-inserted input."""
    Hv, Wv = H * up, W * up
    Ho, Wo = -(-Hv // stride), -(-Wv // stride)
    ph = max((Ho - 1) * stride + kh - Hv, 0)
    pw = max((Wo - 1) * stride + kw - Wv, 0)
    return make_geom(N, H, W, Ci, Ho, Wo, Co, kh, kw, stride, up, ph // 2, pw // 2)


def geom_adjoint(g):
    """Geometry of the data-gradient / transposed convolution of `g` (input = g's output)."""
    return make_geom(g.N, g.Ho, g.Wo, g.Co, g.Hin, g.Win, g.Ci, g.kh, g.kw, g.U, g.S,
                     g.kh - 1 - g.pt, g.kw - 1 - g.pl)


# ------------------------------------------------------------------------------------------------
# convolution family
# ------------------------------------------------------------------------------------------------
def _prep_elems(kh, kw, Ci, Co, which):
    """bf16 elements of a weight operand buffer (which: 0 forward, 1 backward image)."""
    return int(lib().cg_weight_prep_elems(int(kh), int(kw), int(Ci), int(Co), int(which)))


def weight_prep(w, scale=None, want_fwd=True, want_bwd=False):
    """w fp32 [kh,kw,Ci,Co] -> (bt_fwd [Co, Kp] bf16, bt_bwd [Ci, Kbp] bf16)."""
    _req(w, F32, "w")
    _req(scale, F32, "scale", True)
    kh, kw, Ci, Co = w.shape
    bt_f = bt_b = None
    if want_fwd:
        Kp = (kh * kw * Ci + 7) // 8 * 8
        buf = torch.empty(_prep_elems(kh, kw, Ci, Co, 0), dtype=BF16, device=w.device)
        bt_f = buf[:Co * Kp].view(Co, Kp)
    if want_bwd:
        Kbp = (kh * kw * Co + 7) // 8 * 8
        buf = torch.empty(_prep_elems(kh, kw, Ci, Co, 1), dtype=BF16, device=w.device)
        bt_b = buf[:Ci * Kbp].view(Ci, Kbp)
    check(lib().cg_weight_prep(_p(w), kh, kw, Ci, Co, _p(scale), _p(bt_f), _p(bt_b), _stream()),
          "cg_weight_prep")
    return bt_f, bt_b


def _leaky_prepass(geom):
    """Leaky self-gates are applied by an element-wise pass in front of the convolution wherever an
    MFMA-tiled kernel can take the launch afterwards (channel counts in multiples of 32; image
    inputs and linear layers keep the fused gate of the generic / VALU kernels)."""
    return geom.Ci % 32 == 0 and geom.Hin * geom.Win > 1 and os.environ.get("CGAMD_LEAKY_PREPASS", "1") != "0"


def gconv(geom, x, bt, bias=None, gate_in=None, slope_in=0.0, gate_out=None, slope_out=0.0,
          residual=None, out_f32=False, act_out=None):
    """act_out: leaky-ReLU slope applied to (conv + bias) itself (0.0 = ReLU); exclusive with
    gate_out."""
    _req(x, BF16, "x")
    _req(bt, BF16, "bt")
    _req(bias, F32, "bias", True)
    _req(gate_in, BF16, "gate_in", True)
    _req(gate_out, BF16, "gate_out", True)
    _req(residual, BF16, "residual", True)
    if x.numel() != geom.N * geom.Hin * geom.Win * geom.Ci:
        raise ValueError("x has %d elements, geometry expects %s" % (x.numel(), geom.key()))
    Kp = (geom.kh * geom.kw * geom.Ci + 7) // 8 * 8
    if tuple(bt.shape) != (geom.Co, Kp):
        raise ValueError("bt shape %s != (%d, %d)" % (tuple(bt.shape), geom.Co, Kp))
    oshape = (geom.N, geom.Ho, geom.Wo, geom.Co)
    for t, nm in ((gate_out, "gate_out"), (residual, "residual")):
        if t is not None and t.numel() != geom.N * geom.Ho * geom.Wo * geom.Co:
            raise ValueError("%s has the wrong number of elements" % nm)
    if gate_in is not None and gate_in.numel() != x.numel():
        raise ValueError("gate_in has the wrong number of elements")
    if bias is not None and bias.numel() != geom.Co:
        raise ValueError("bias has the wrong number of elements")
    if gate_in is not None and gate_in.data_ptr() != x.data_ptr():
        # a gate that is a separate tensor (second-order terms of the gradient penalty,
        # penalty_lib.py:74-82) is only understood by the generic kernel (~85 TFLOP/s): one
        # element-wise pass x * D(gate) and the MFMA-tiled kernels apply (exact for ReLU gates)
        x, gate_in = lrelu_bwd(gate_in, x, slope_in), None
    elif gate_in is not None and float(slope_in) != 0.0 and _leaky_prepass(geom):
        # y = conv(lrelu_slope(x)) (sndcgan.py:97-122, dcgan.py:108-124): the MFMA-tiled kernels gate
        # their input with an integer max (ReLU only); the leaky form runs on the generic gather
        # kernel at ~200 TFLOP/s.  One element-wise pass in front (x read + written once) and the
        # tiled kernels apply: 347 -> ~90 us on 64 x 128^2 x 64 -> 64^2 x 128, 4x4 / stride 2
        x, gate_in = lrelu_bwd(x, x, slope_in), None
    out = torch.empty(oshape, dtype=F32 if out_f32 else BF16, device=x.device)
    if act_out is not None:
        if gate_out is not None:
            raise ValueError("act_out and gate_out are mutually exclusive")
        gate_out, slope_out = out, act_out   # gate_out == out: the value itself is the gate
    check(lib().cg_gconv(ctypes.byref(geom), _p(x), _p(bt), _p(out), int(out_f32), _p(bias),
                         _p(gate_in), float(slope_in), _p(gate_out), float(slope_out),
                         _p(residual), _stream()), "cg_gconv")
    return out


def _pixel_pitch(t, c, name):
    """Elements between consecutive pixels of an NHWC tensor or of a channel slice t[..., a:a+c] of
    one; the pixels themselves must be densely packed at that pitch."""
    if t.dim() != 4 or t.shape[3] != c or t.stride(3) != 1:
        raise ValueError("%s must be [N, H, W, %d] with unit channel stride" % (name, c))
    ld = t.stride(2)
    if t.stride(1) != ld * t.shape[2] or t.stride(0) != ld * t.shape[2] * t.shape[1]:
        raise ValueError("%s: pixels are not a dense grid at pitch %d" % (name, ld))
    return ld


def gconv_ld_supported(geom, in_ld, out_ld):
    return bool(lib().cg_gconv_ld_supported(ctypes.byref(geom), int(in_ld), int(out_ld)))


def gconv_ld(geom, x, bt, out, bias=None, relu=False):
    """cg_gconv_ld: conv (+ bias, + ReLU) reading a channel slice `x` of a wider NHWC tensor and
    writing into the channel slice `out` of another (bf16, or fp32 out) -- both torch views.  relu:
    True / False, or the number of leading output channels that get it (a multiple of 8)."""
    for t, dt, nm in ((x, BF16, "x"), (bt, BF16, "bt")):
        if not t.is_cuda or t.dtype != dt:
            raise ValueError("%s must be %s on the GPU" % (nm, dt))
    if not out.is_cuda or out.dtype not in (BF16, F32):
        raise ValueError("out must be bf16 or fp32 on the GPU")
    _req(bias, F32, "bias", True)
    if tuple(x.shape) != (geom.N, geom.Hin, geom.Win, geom.Ci):
        raise ValueError("x shape %s, geometry expects %s" % (tuple(x.shape), geom.key()))
    if tuple(out.shape) != (geom.N, geom.Ho, geom.Wo, geom.Co):
        raise ValueError("out shape %s, geometry expects %s" % (tuple(out.shape), geom.key()))
    Kp = (geom.kh * geom.kw * geom.Ci + 7) // 8 * 8
    if tuple(bt.shape) != (geom.Co, Kp) or not bt.is_contiguous():
        raise ValueError("bt shape %s != (%d, %d)" % (tuple(bt.shape), geom.Co, Kp))
    if bias is not None and bias.numel() != geom.Co:
        raise ValueError("bias has the wrong number of elements")
    relu_cols = geom.Co if relu is True else (0 if relu is False else int(relu))
    check(lib().cg_gconv_ld(ctypes.byref(geom), _p(x), _pixel_pitch(x, geom.Ci, "x"), _p(bt), _p(out),
                            _pixel_pitch(out, geom.Co, "out"), int(out.dtype == F32), _p(bias),
                            relu_cols, _stream()), "cg_gconv_ld")
    return out


def pool2d_ld(x, k, s, p, kind, ho, wo, out, bias=None, relu=False):
    """cg_pool2d_ld: pooling (kind 0 max / 1 avg with TF 'SAME' counts) of a channel slice `x` into the
    channel slice `out` (both bf16 torch views of NHWC tensors), then + bias (fp32 [C]) and ReLU."""
    for t, nm in ((x, "x"), (out, "out")):
        if not t.is_cuda or t.dtype != BF16:
            raise ValueError("%s must be bf16 on the GPU" % nm)
    _req(bias, F32, "bias", True)
    n, h, w, c = x.shape
    if tuple(out.shape) != (n, ho, wo, c):
        raise ValueError("out shape %s, expected %s" % (tuple(out.shape), (n, ho, wo, c)))
    if bias is not None and bias.numel() != c:
        raise ValueError("bias has the wrong number of elements")
    check(lib().cg_pool2d_ld(_p(x), _pixel_pitch(x, c, "x"), n, h, w, c, int(k), int(s), int(p),
                             int(kind), int(ho), int(wo), _p(out), _pixel_pitch(out, c, "out"),
                             _p(bias), int(bool(relu)), _stream()), "cg_pool2d_ld")
    return out


def gconv_fused_rows(geom):
    """Rows of batch-norm partial sums the fused convolution kernel emits for `geom`; 0 when the
    geometry is not covered (use gconv + bn_stats / bn_apply then)."""
    return int(lib().cg_gconv_fused_rows(ctypes.byref(geom)))


def gconv_fused_prologue_supported(geom):
    """True when gconv_fused covers `geom` with the batch-norm prologue alone (also the RGB-output
    convolutions, for which gconv_fused_rows is 0: no statistics epilogue there)."""
    return bool(lib().cg_gconv_fused_prologue_supported(ctypes.byref(geom)))


def gconv_pool_supported(geom):
    """True when the pooled-output convolution (gconv_fused(pool=True)) and its pooled-gradient
    weight gradient (gwgrad_pooled) cover `geom`."""
    return bool(lib().cg_gconv_pool_supported(ctypes.byref(geom)))


def gconv_fused_phases(geom):
    """Phase blocks of the statistics rows cg_gconv_fused emits for `geom` (bn_finalize's `phases`)."""
    return int(lib().cg_gconv_fused_phases(ctypes.byref(geom)))


def gconv_fused(geom, x, bt, bias=None, residual=None, out_f32=False, bn=None, want_stats=False,
                gate_in=None, gate_out=None, slope_out=0.0, pool=False, in_up=False,
                out_scale=1.0):
    """cg_gconv_fused: the halo-staged convolution with its fusions --
      bn = (mean, var, gamma, beta, eps, per_sample): batch norm + ReLU applied to the input in LDS;
      want_stats: per-channel partial sums of the output (for the next batch norm);
      pool: 2x2 average pooling of (conv + bias) in the epilogue, `residual` at the pooled size;
      in_up: x is [N, Hin/2, Win/2, Ci], read as its nearest-neighbour up-sampling; out_scale
             multiplies the convolution sum (pool gradient: in_up with out_scale = 0.25);
      gate_in: ReLU of the input itself (must be x); gate_out / slope_out as gconv.
    Returns (out, partials or None)."""
    from compare_gan_amd.hip._lib import ConvFusion
    _req(x, BF16, "x")
    _req(bt, BF16, "bt")
    _req(bias, F32, "bias", True)
    _req(residual, BF16, "residual", True)
    _req(gate_out, BF16, "gate_out", True)
    if gate_in is not None and gate_in.data_ptr() != x.data_ptr():
        raise ValueError("gconv_fused: gate_in must be the input itself")
    us = 2 if in_up else 1
    if x.numel() * us * us != geom.N * geom.Hin * geom.Win * geom.Ci:
        raise ValueError("x has %d elements, geometry expects %s" % (x.numel(), geom.key()))
    fu = ConvFusion()
    keep = []
    if bn is not None:
        mean, var, gamma, beta, eps, per_sample = bn
        for t, nm in ((mean, "bn mean"), (var, "bn var")):
            _req(t, F32, nm)
        fu.bn_stat_group = stat_group_of(mean, geom.N, geom.Ci)   # [Ci] or [groups, Ci]
        if var.numel() != mean.numel():
            raise ValueError("bn var has the wrong number of elements")
        want = geom.N * geom.Ci if per_sample else geom.Ci
        for t, nm in ((gamma, "bn gamma"), (beta, "bn beta")):
            _req(t, F32, nm, True)
            if t is not None and t.numel() != want:
                raise ValueError("%s has the wrong number of elements" % nm)
        fu.bn_mean, fu.bn_var, fu.bn_gamma, fu.bn_beta = _p(mean), _p(var), _p(gamma), _p(beta)
        fu.bn_eps, fu.bn_per_sample = float(eps), int(bool(per_sample))
        keep = [mean, var, gamma, beta]
    stats = None
    if want_stats:
        rows = gconv_fused_rows(geom)
        if rows <= 0:
            raise ValueError("geometry %s is not covered by the fused kernel" % (geom.key(),))
        stats = torch.empty((rows, 2 * geom.Co), dtype=F32, device=x.device)
        fu.stats_out = _p(stats)
    fu.pool_out, fu.in_up, fu.out_scale = int(bool(pool)), int(bool(in_up)), float(out_scale)
    ps = 2 if pool else 1
    oshape = (geom.N, geom.Ho // ps, geom.Wo // ps, geom.Co)
    if residual is not None and residual.numel() != oshape[0] * oshape[1] * oshape[2] * oshape[3]:
        raise ValueError("residual has the wrong number of elements")
    out = torch.empty(oshape, dtype=F32 if out_f32 else BF16, device=x.device)
    check(lib().cg_gconv_fused(ctypes.byref(geom), _p(x), _p(bt), _p(out), int(out_f32), _p(bias),
                               _p(gate_in), 0.0, _p(gate_out), float(slope_out), _p(residual),
                               ctypes.byref(fu), _stream()), "cg_gconv_fused")
    del keep
    return out, stats


def gwgrad_pooled(geom, x, dy_pooled, gate_in=None, want_dbias=False, defer=None):
    """dw (+ dbias) of a convolution whose output was 2x2 average-pooled; dy_pooled is the
    gradient w.r.t. the pooled output [N, Ho/2, Wo/2, Co].  defer: a DeferCtx that records the
    split reduction (outputs valid after its flush)."""
    _req(x, BF16, "x")
    _req(dy_pooled, BF16, "dy_pooled")
    if gate_in is not None and gate_in.data_ptr() != x.data_ptr():
        raise ValueError("gwgrad_pooled: gate_in must be the input itself")
    if dy_pooled.numel() * 4 != geom.N * geom.Ho * geom.Wo * geom.Co:
        raise ValueError("dy_pooled has the wrong number of elements for %s" % (geom.key(),))
    dw = torch.empty((geom.kh, geom.kw, geom.Ci, geom.Co), dtype=F32, device=x.device)
    dbias = torch.empty((geom.Co,), dtype=F32, device=x.device) if want_dbias else None
    ws = _ws(lib().cg_gwgrad_workspace_bytes(ctypes.byref(geom)), x)
    if defer is not None:
        defer.keep.append(ws)
        check(lib().cg_gwgrad_pooled_deferred(ctypes.byref(geom), _p(x), _p(gate_in), 0.0,
                                              _p(dy_pooled), _p(dw), 0, _p(dbias), _p(ws),
                                              ws.numel(), _stream(), defer.handle),
              "cg_gwgrad_pooled_deferred")
        return dw, dbias
    check(lib().cg_gwgrad_pooled(ctypes.byref(geom), _p(x), _p(gate_in), 0.0, _p(dy_pooled),
                                 _p(dw), 0, _p(dbias), _p(ws), ws.numel(), _stream()),
          "cg_gwgrad_pooled")
    return dw, dbias


def bn_finalize(partials, count, moving_mean=None, moving_var=None, decay=0.0, groups=1, phases=1):
    """mean / var [C] from partial sums [rows][2C] over `count` values per channel; optionally the
    moving-average update of cg_bn_stats.  groups > 1: mean / var [groups, C] over count / groups
    values each (rows = [phases][rows / phases], each phase split in order over the groups)."""
    _req(partials, F32, "partials")
    rows, c2 = partials.shape
    c = c2 // 2
    if groups > 1:
        mean = torch.empty((groups, c), dtype=F32, device=partials.device)
        var = torch.empty((groups, c), dtype=F32, device=partials.device)
        check(lib().cg_bn_finalize_groups(_p(partials), int(rows), int(c), int(count) // groups,
                                          int(groups), int(phases), _p(mean), _p(var),
                                          _p(moving_mean), _p(moving_var), float(decay), _stream()),
              "cg_bn_finalize_groups")
        return mean, var
    mean = torch.empty((c,), dtype=F32, device=partials.device)
    var = torch.empty((c,), dtype=F32, device=partials.device)
    check(lib().cg_bn_finalize(_p(partials), int(rows), int(c), int(count), _p(mean), _p(var),
                               _p(moving_mean), _p(moving_var), float(decay), _stream()),
          "cg_bn_finalize")
    return mean, var


def gwgrad(geom, x, dy, gate_in=None, slope_in=0.0, gate_dy=None, slope_dy=0.0, want_dbias=False,
           out=None, accumulate=False, defer=None):
    """dw fp32 [kh,kw,Ci,Co] (+ dbias [Co]) of the gather convolution `geom`.  defer: a DeferCtx
    that records the split reduction (outputs valid after its flush)."""
    _req(x, BF16, "x")
    _req(dy, BF16, "dy")
    _req(gate_in, BF16, "gate_in", True)
    _req(gate_dy, BF16, "gate_dy", True)
    if x.numel() != geom.N * geom.Hin * geom.Win * geom.Ci:
        raise ValueError("x has the wrong number of elements for %s" % (geom.key(),))
    if dy.numel() != geom.N * geom.Ho * geom.Wo * geom.Co:
        raise ValueError("dy has the wrong number of elements for %s" % (geom.key(),))
    # separate gate tensors: see gconv
    if gate_in is not None and gate_in.data_ptr() != x.data_ptr():
        x, gate_in = lrelu_bwd(gate_in, x, slope_in), None
    elif gate_in is not None and float(slope_in) != 0.0 and _leaky_prepass(geom):
        x, gate_in = lrelu_bwd(x, x, slope_in), None   # see gconv
    if gate_dy is not None:
        dy, gate_dy = lrelu_bwd(gate_dy, dy, slope_dy), None
    dw = out if out is not None else torch.empty((geom.kh, geom.kw, geom.Ci, geom.Co), dtype=F32,
                                                  device=x.device)
    _req(dw, F32, "dw")
    dbias = torch.empty((geom.Co,), dtype=F32, device=x.device) if want_dbias else None
    nbytes = lib().cg_gwgrad_workspace_bytes(ctypes.byref(geom))
    ws = _ws(nbytes, x)
    if defer is not None:
        defer.keep.append(ws)
        check(lib().cg_gwgrad_deferred(ctypes.byref(geom), _p(x), _p(gate_in), float(slope_in),
                                       _p(dy), _p(gate_dy), float(slope_dy), _p(dw),
                                       int(accumulate), _p(dbias), _p(ws), ws.numel(), _stream(),
                                       defer.handle), "cg_gwgrad_deferred")
        return dw, dbias
    check(lib().cg_gwgrad(ctypes.byref(geom), _p(x), _p(gate_in), float(slope_in), _p(dy),
                          _p(gate_dy), float(slope_dy), _p(dw), int(accumulate), _p(dbias),
                          _p(ws), ws.numel(), _stream()), "cg_gwgrad")
    return dw, dbias


_GROUPABLE = {}


def gwgrad_groupable(geom):
    """True when gwgrad_multi runs this weight gradient in a launch shared with other layers."""
    key = geom.key()
    if key not in _GROUPABLE:
        _GROUPABLE[key] = bool(lib().cg_gwgrad_groupable(ctypes.byref(geom)))
    return _GROUPABLE[key]


def gwgrad_multi(jobs):
    """Several weight gradients in one call (cg_gwgrad_multi): jobs = [(geom, x, dy, relu_in,
    dw, dbias)], x / dy bf16, relu_in = the input carries its own ReLU gate, dw fp32
    [kh,kw,Ci,Co] and dbias fp32 [Co] (or None) are written.  The 3x3 layers on small maps share
    launches; the rest run one by one."""
    if not jobs:
        return
    items = (WgradItem * len(jobs))()
    need = 256
    for it, (geom, x, dy, relu_in, dw, dbias) in zip(items, jobs):
        _req(x, BF16, "x")
        _req(dy, BF16, "dy")
        _req(dw, F32, "dw")
        _req(dbias, F32, "dbias", True)
        if x.numel() != geom.N * geom.Hin * geom.Win * geom.Ci:
            raise ValueError("x has the wrong number of elements for %s" % (geom.key(),))
        if dy.numel() != geom.N * geom.Ho * geom.Wo * geom.Co:
            raise ValueError("dy has the wrong number of elements for %s" % (geom.key(),))
        it.geom = geom
        it.in_ = _p(x)
        it.gate_in = _p(x) if relu_in else None
        it.slope_in = 0.0
        it.accumulate = 0
        it.dy = _p(dy)
        it.dw = _p(dw)
        it.dbias = _p(dbias)
        need = max(need, int(lib().cg_gwgrad_workspace_bytes(ctypes.byref(it.geom))))
    ws = _ws(need, jobs[0][1])
    check(lib().cg_gwgrad_multi(items, len(jobs), _p(ws), ws.numel(), _stream()), "cg_gwgrad_multi")


# ------------------------------------------------------------------------------------------------
# spectral norm
# ------------------------------------------------------------------------------------------------
def spectral_norm(w2d, u, mode, eps=1e-12):
    """One power iteration; updates u IN PLACE.  Returns (v, sigma, inv_sigma)."""
    _req(w2d, F32, "w")
    _req(u, F32, "u")
    K, Co = w2d.shape
    if u.numel() != (K if mode == 0 else Co):
        raise ValueError("u has %d elements for mode %d of a [%d,%d] matrix" % (u.numel(), mode, K, Co))
    v = torch.empty((Co if mode == 0 else K,), dtype=F32, device=w2d.device)
    sig = torch.empty((2,), dtype=F32, device=w2d.device)
    ws = _ws(lib().cg_spectral_norm_workspace_bytes(K, Co), w2d)
    check(lib().cg_spectral_norm(_p(w2d), K, Co, mode, float(eps), _p(u), _p(u), _p(v),
                                 ctypes.c_void_p(sig.data_ptr()),
                                 ctypes.c_void_p(sig.data_ptr() + 4), _p(ws), ws.numel(),
                                 _stream()), "cg_spectral_norm")
    return v, sig[0:1], sig[1:2]


def sn_backward(dwbar2d, w2d, a_k, b_co, sigma):
    _req(dwbar2d, F32, "dwbar")
    _req(w2d, F32, "w")
    K, Co = w2d.shape
    dw = torch.empty_like(w2d)
    ws = _ws(lib().cg_sn_backward_workspace_bytes(K, Co), w2d)
    check(lib().cg_sn_backward(_p(dwbar2d), _p(w2d), K, Co, _p(a_k), _p(b_co), _p(sigma), _p(dw),
                               _p(ws), ws.numel(), _stream()), "cg_sn_backward")
    return dw


def _carve(flat, sizes):
    """Views into one flat buffer (one allocation per network call instead of one per tensor)."""
    out, off = [], 0
    for n in sizes:
        out.append(flat[off:off + n])
        off += n
    return out


def spectral_norm_multi(weights2d, us, modes, eps=1e-12, want_wbar=True):
    """One power-iteration round for a list of [K, Co] weights; every u is updated IN PLACE.
    Returns (u_news, vs, sigmas [each a 2-vector: sigma, 1/sigma], wbars or None)."""
    n = len(weights2d)
    dev = weights2d[0].device
    vec_sizes, ws_sizes, wb_sizes = [], [], []
    for w, u, m in zip(weights2d, us, modes):
        _req(w, F32, "w")
        _req(u, F32, "u")
        K, Co = w.shape
        lu, lv = (K, Co) if m == 0 else (Co, K)
        if u.numel() != lu:
            raise ValueError("u has %d elements for mode %d of a [%d,%d] matrix" % (u.numel(), m, K, Co))
        vec_sizes += [lu, lv, 2]
        ws_sizes.append(int(lib().cg_spectral_norm_multi_workspace_floats(K, Co)))
        wb_sizes.append(K * Co)
    vecs = _carve(torch.empty(sum(vec_sizes), dtype=F32, device=dev), vec_sizes)
    wss = _carve(torch.empty(sum(ws_sizes), dtype=F32, device=dev), ws_sizes)
    wbars = _carve(torch.empty(sum(wb_sizes), dtype=F32, device=dev), wb_sizes) if want_wbar else None
    items = (_lib.SNItem * n)()
    u_news, vs, sigmas = [], [], []
    for i, (w, u, m) in enumerate(zip(weights2d, us, modes)):
        it = items[i]
        u_new, v, sig = vecs[3 * i], vecs[3 * i + 1], vecs[3 * i + 2]
        it.w, it.u, it.u_out, it.v_out, it.sigma = (w.data_ptr(), u.data_ptr(), u_new.data_ptr(),
                                                    v.data_ptr(), sig.data_ptr())
        it.wbar = wbars[i].data_ptr() if want_wbar else None
        it.ws = wss[i].data_ptr()
        it.K, it.Co, it.mode = w.shape[0], w.shape[1], m
        u_news.append(u_new); vs.append(v); sigmas.append(sig)
    check(lib().cg_spectral_norm_multi(ctypes.cast(items, ctypes.c_void_p), n, float(eps),
                                       _stream()), "cg_spectral_norm_multi")
    if want_wbar:
        wbars = [wb.view(w.shape) for wb, w in zip(wbars, weights2d)]
    return u_news, vs, sigmas, wbars


def sn_backward_multi(dwbars, weights2d, a_ks, b_cos, sigmas):
    """dw_i = (dwbar_i - <dwbar_i, w_i>/sigma_i a_i b_i^T) / sigma_i for a list of weights."""
    n = len(dwbars)
    dev = weights2d[0].device
    sizes = [w.numel() for w in weights2d]
    ws_sizes = [int(lib().cg_sn_backward_multi_workspace_floats(w.shape[0], w.shape[1]))
                for w in weights2d]
    dws = _carve(torch.empty(sum(sizes), dtype=F32, device=dev), sizes)
    wss = _carve(torch.empty(sum(ws_sizes), dtype=F32, device=dev), ws_sizes)
    items = (_lib.SNBwdItem * n)()
    for i in range(n):
        _req(dwbars[i], F32, "dwbar")
        it = items[i]
        it.dwbar, it.w, it.a_k, it.b_co = (dwbars[i].data_ptr(), weights2d[i].data_ptr(),
                                           a_ks[i].data_ptr(), b_cos[i].data_ptr())
        it.sigma, it.dw, it.ws = sigmas[i].data_ptr(), dws[i].data_ptr(), wss[i].data_ptr()
        it.K, it.Co = weights2d[i].shape
    check(lib().cg_sn_backward_multi(ctypes.cast(items, ctypes.c_void_p), n, _stream()),
          "cg_sn_backward_multi")
    return [d.view(w.shape) for d, w in zip(dws, weights2d)]


def weight_prep_multi(weights4d, want_fwd=True, want_bwd=False):
    """fp32 [kh,kw,Ci,Co] weights -> lists of (bt_fwd [Co,Kp], bt_bwd [Ci,Kbp]) bf16 images."""
    n = len(weights4d)
    dev = weights4d[0].device
    f_sizes, b_sizes = [], []
    for w in weights4d:
        _req(w, F32, "w")
        kh, kw, Ci, Co = w.shape
        f_sizes.append(_prep_elems(kh, kw, Ci, Co, 0) if want_fwd else 0)
        b_sizes.append(_prep_elems(kh, kw, Ci, Co, 1) if want_bwd else 0)
    # every image starts 16-byte aligned (sizes are multiples of 8 bf16 elements)
    fbuf = _carve(torch.empty(sum(f_sizes), dtype=BF16, device=dev), f_sizes)
    bbuf = _carve(torch.empty(sum(b_sizes), dtype=BF16, device=dev), b_sizes)
    items = (_lib.PrepItem * n)()
    bt_f, bt_b = [], []
    for i, w in enumerate(weights4d):
        kh, kw, Ci, Co = w.shape
        it = items[i]
        it.w = w.data_ptr()
        it.T, it.Ci, it.Co = kh * kw, Ci, Co
        f = fbuf[i][:Co * ((kh * kw * Ci + 7) // 8 * 8)].view(Co, -1) if want_fwd else None
        b = bbuf[i][:Ci * ((kh * kw * Co + 7) // 8 * 8)].view(Ci, -1) if want_bwd else None
        it.bt_fwd = f.data_ptr() if f is not None else None
        it.bt_bwd = b.data_ptr() if b is not None else None
        bt_f.append(f); bt_b.append(b)
    check(lib().cg_weight_prep_multi(ctypes.cast(items, ctypes.c_void_p), n, _stream()),
          "cg_weight_prep_multi")
    return bt_f, bt_b


def flatten_multi(tensors, flat):
    """Copies a list of fp32 tensors back to back into `flat` (one launch per 32 tensors)."""
    _req(flat, F32, "flat")
    n = len(tensors)
    ptrs = (ctypes.c_void_p * n)()
    sizes = (ctypes.c_int64 * n)()
    total = 0
    for i, t in enumerate(tensors):
        _req(t, F32, "tensor")
        ptrs[i] = t.data_ptr()
        sizes[i] = t.numel()
        total += t.numel()
    if total != flat.numel():
        raise ValueError("flat has %d elements, the tensors %d" % (flat.numel(), total))
    check(lib().cg_flatten_multi(ctypes.cast(ptrs, ctypes.c_void_p),
                                 ctypes.cast(sizes, ctypes.c_void_p), n, _p(flat), _stream()),
          "cg_flatten_multi")
    return flat


def scale_f32(x, scale_dev=None, scale_host=1.0):
    _req(x, F32, "x")
    out = torch.empty_like(x)
    check(lib().cg_scale_f32(_p(x), _p(scale_dev), float(scale_host), _p(out), x.numel(),
                             _stream()), "cg_scale_f32")
    return out


def scale_count_nan(x, scale, out, nan_count):
    """out = x * scale (fp32, same shape, may be a slice of a larger buffer) and nan_count[0] +=
    NaNs of x (int32 device tensor): eval_utils.FakeImageSink."""
    _req(x, F32, "x")
    _req(out, F32, "out")
    _req(nan_count, torch.int32, "nan_count")
    if out.numel() != x.numel():
        raise ValueError("out must have as many elements as x")
    check(lib().cg_scale_count_nan_f32(_p(x), float(scale), _p(out), x.numel(), _p(nan_count),
                                       _stream()), "cg_scale_count_nan_f32")
    return out


# ------------------------------------------------------------------------------------------------
# batch norm
# ------------------------------------------------------------------------------------------------
def bn_stats(x3, moving_mean=None, moving_var=None, decay=0.0, groups=1):
    """x3 [N, HW, C] bf16 -> mean, var fp32 [C]; optionally updates the moving averages.
    groups > 1: independent statistics for `groups` consecutive blocks of N / groups samples ->
    mean, var [groups, C]; the moving averages take the groups' updates in order."""
    _req(x3, BF16, "x")
    _req(moving_mean, F32, "moving_mean", True)
    _req(moving_var, F32, "moving_var", True)
    N, HW, C = x3.shape
    if groups > 1:
        if N % groups:
            raise ValueError("bn_stats: %d samples do not split into %d groups" % (N, groups))
        mean = torch.empty((groups, C), dtype=F32, device=x3.device)
        var = torch.empty((groups, C), dtype=F32, device=x3.device)
        ws = _ws(lib().cg_bn_stats_groups_workspace_bytes(N * HW, C, groups), x3)
        check(lib().cg_bn_stats_groups(_p(x3), N * HW, C, groups, _p(mean), _p(var),
                                       _p(moving_mean), _p(moving_var), float(decay), _p(ws),
                                       ws.numel(), _stream()), "cg_bn_stats_groups")
        return mean, var
    mean = torch.empty((C,), dtype=F32, device=x3.device)
    var = torch.empty((C,), dtype=F32, device=x3.device)
    ws = _ws(lib().cg_bn_stats_workspace_bytes(N * HW, C), x3)
    check(lib().cg_bn_stats(_p(x3), N * HW, C, _p(mean), _p(var), _p(moving_mean), _p(moving_var),
                            float(decay), _p(ws), ws.numel(), _stream()), "cg_bn_stats")
    return mean, var


def stat_group_of(mean, n, c):
    """Samples per statistics group for mean / var of shape [C] (0) or [groups, C]."""
    if mean.numel() == c:
        return 0
    groups = mean.numel() // c
    if groups * c != mean.numel() or n % groups:
        raise ValueError("statistics of %d elements do not fit %d samples x %d channels" % (
            mean.numel(), n, c))
    return n // groups


def bn_apply(x3, mean, var, eps, gamma=None, beta=None, per_sample=False, relu=False):
    """mean / var [C], or [groups, C] for group statistics (bn_stats(groups=...))."""
    _req(x3, BF16, "x")
    N, HW, C = x3.shape
    for t, nm in ((mean, "mean"), (var, "var")):
        _req(t, F32, nm)
    _req(gamma, F32, "gamma", True)
    _req(beta, F32, "beta", True)
    y = torch.empty_like(x3)
    check(lib().cg_bn_apply_groups(_p(x3), N, HW, C, _p(mean), _p(var), float(eps), _p(gamma),
                                   _p(beta), int(per_sample), stat_group_of(mean, N, C),
                                   int(relu), _p(y), _stream()), "cg_bn_apply")
    return y


def bn_backward(x3, y3, dy3, mean, var, eps, gamma=None, per_sample=False, relu=False,
                batch_stats=True, want_dgamma=True, want_dbeta=True, sync_fn=None):
    """Two-stage BN backward; sync_fn(m12) all-reduces the per-channel means across replicas."""
    _req(x3, BF16, "x")
    _req(dy3, BF16, "dy")
    _req(y3, BF16, "y", not relu)
    N, HW, C = x3.shape
    dx = torch.empty_like(x3)
    pshape = (N, C) if per_sample else (C,)
    dgamma = torch.empty(pshape, dtype=F32, device=x3.device) if want_dgamma else None
    dbeta = torch.empty(pshape, dtype=F32, device=x3.device) if want_dbeta else None
    m12 = torch.empty((2 * C,), dtype=F32, device=x3.device)
    ws = _ws(lib().cg_bn_backward_workspace_bytes(N, HW, C), x3)
    check(lib().cg_bn_backward_reduce(_p(x3), _p(y3), _p(dy3), N, HW, C, _p(mean), _p(var),
                                      float(eps), _p(gamma), int(per_sample), int(relu),
                                      _p(dgamma), _p(dbeta), _p(m12), _p(ws), ws.numel(),
                                      _stream()), "cg_bn_backward_reduce")
    if sync_fn is not None and batch_stats:
        m12 = sync_fn(m12)
    check(lib().cg_bn_backward_apply(_p(x3), _p(y3), _p(dy3), N, HW, C, _p(mean), _p(var),
                                     float(eps), _p(gamma), int(per_sample), int(relu),
                                     int(batch_stats), _p(m12), _p(dx), _stream()),
          "cg_bn_backward_apply")
    return dx, dgamma, dbeta


def bn_accumulate(accu_mean, accu_var, accu_counter, mean, var):
    """accu_mean += mean; accu_var += var; accu_counter += 1 (in place, on the device)."""
    for t, nm in ((accu_mean, "accu_mean"), (accu_var, "accu_var"), (accu_counter, "accu_counter"),
                  (mean, "mean"), (var, "var")):
        _req(t, F32, nm)
    check(lib().cg_bn_accumulate(_p(accu_mean), _p(accu_var), _p(accu_counter), _p(mean), _p(var),
                                 mean.numel(), _stream()), "cg_bn_accumulate")


def bn_accumulated_moments(accu_mean, accu_var, accu_counter):
    """(accu_mean / accu_counter, accu_var / accu_counter) without a host round trip."""
    for t, nm in ((accu_mean, "accu_mean"), (accu_var, "accu_var"), (accu_counter, "accu_counter")):
        _req(t, F32, nm)
    mean, var = torch.empty_like(accu_mean), torch.empty_like(accu_var)
    check(lib().cg_bn_accumulated_moments(_p(accu_mean), _p(accu_var), _p(accu_counter), _p(mean),
                                          _p(var), accu_mean.numel(), _stream()),
          "cg_bn_accumulated_moments")
    return mean, var


def bn_moments_convert(mean, second, to_variance, scale=1.0):
    _req(mean, F32, "mean")
    _req(second, F32, "second")
    check(lib().cg_bn_moments_convert(_p(mean), _p(second), mean.numel(), int(to_variance),
                                      float(scale), _stream()), "cg_bn_moments_convert")


def bn_update_moving(moving_mean, moving_var, mean, var, decay):
    check(lib().cg_bn_update_moving(_p(moving_mean), _p(moving_var), _p(mean), _p(var),
                                    mean.numel(), float(decay), _stream()), "cg_bn_update_moving")


# ------------------------------------------------------------------------------------------------
# element-wise / pooling
# ------------------------------------------------------------------------------------------------
def lrelu(x, slope):
    _req(x, BF16, "x")
    y = torch.empty_like(x)
    check(lib().cg_lrelu(_p(x), float(slope), _p(y), x.numel(), _stream()), "cg_lrelu")
    return y


def lrelu_bwd(x, dy, slope):
    _req(x, BF16, "x")
    _req(dy, BF16, "dy")
    dx = torch.empty_like(x)
    check(lib().cg_lrelu_bwd(_p(x), _p(dy), float(slope), _p(dx), x.numel(), _stream()),
          "cg_lrelu_bwd")
    return dx


def axpby(a, alpha, b=None, beta=0.0):
    _req(a, BF16, "a")
    _req(b, BF16, "b", True)
    out = torch.empty_like(a)
    check(lib().cg_axpby(_p(a), float(alpha), _p(b), float(beta), _p(out), a.numel(), _stream()),
          "cg_axpby")
    return out


def sum4(a, b, c, d=None):
    """a + b + c (+ d): bf16 tensors of one shape, fp32 sum with one rounding (cg_sum4)."""
    for t, nm in ((a, "a"), (b, "b"), (c, "c")):
        _req(t, BF16, nm)
    _req(d, BF16, "d", True)
    out = torch.empty_like(a)
    check(lib().cg_sum4(_p(a), _p(b), _p(c), _p(d), _p(out), a.numel(), _stream()), "cg_sum4")
    return out


def axpby_f32(a, alpha, b=None, beta=0.0):
    _req(a, F32, "a")
    _req(b, F32, "b", True)
    out = torch.empty_like(a)
    check(lib().cg_axpby_f32(_p(a), float(alpha), _p(b), float(beta), _p(out), a.numel(),
                             _stream()), "cg_axpby_f32")
    return out


def axpy_dev(x, o, sigma):
    """x + sigma * o (bf16) with sigma a 1-element fp32 device tensor; x may be None."""
    _req(o, BF16, "o")
    _req(x, BF16, "x", True)
    _req(sigma, F32, "sigma")
    out = torch.empty_like(o)
    check(lib().cg_axpy_dev(_p(x), _p(o), _p(sigma), _p(out), o.numel(), _stream()),
          "cg_axpy_dev")
    return out


def dot_bf16(a, b):
    _req(a, BF16, "a")
    _req(b, BF16, "b")
    out = torch.empty((1,), dtype=F32, device=a.device)
    ws = _ws(lib().cg_dot_bf16_workspace_bytes(a.numel()), a)
    check(lib().cg_dot_bf16(_p(a), _p(b), a.numel(), _p(out), _p(ws), ws.numel(), _stream()),
          "cg_dot_bf16")
    return out


def _pool(fn, name, x):
    _req(x, BF16, "x")
    N, H, W, C = x.shape
    y = torch.empty((N, H // 2, W // 2, C), dtype=BF16, device=x.device)
    check(fn(_p(x), N, H, W, C, _p(y), _stream()), name)
    return y


def avgpool2(x):
    return _pool(lib().cg_avgpool2, "cg_avgpool2", x)


def maxpool2(x):
    return _pool(lib().cg_maxpool2, "cg_maxpool2", x)


def unpool2(x, residual=None):
    """Zero-insertion upsampling x [N,H,W,C] -> [N,2H,2W,C] (+ residual of that shape)."""
    _req(x, BF16, "x")
    N, H, W, C = x.shape
    if residual is not None:
        _req(residual, BF16, "residual")
        if tuple(residual.shape) != (N, 2 * H, 2 * W, C):
            raise ValueError("residual shape %r does not match the upsampled %r" % (
                tuple(residual.shape), (N, 2 * H, 2 * W, C)))
    y = torch.empty((N, 2 * H, 2 * W, C), dtype=BF16, device=x.device)
    check(lib().cg_unpool2(_p(x), _p(residual), N, H, W, C, _p(y), _stream()), "cg_unpool2")
    return y


def unpool2_bwd(dy):
    """dy [N,2H,2W,C] -> dx [N,H,W,C] = dy[:, ::2, ::2, :]."""
    _req(dy, BF16, "dy")
    N, Ho, Wo, C = dy.shape
    if (Ho | Wo) & 1:
        raise ValueError("unpool2_bwd needs even spatial sizes, got %r" % (tuple(dy.shape),))
    dx = torch.empty((N, Ho // 2, Wo // 2, C), dtype=BF16, device=dy.device)
    check(lib().cg_unpool2_bwd(_p(dy), N, Ho // 2, Wo // 2, C, _p(dx), _stream()),
          "cg_unpool2_bwd")
    return dx


def avgpool2_bwd(dy):
    """dy [N,H/2,W/2,C] -> dx [N,H,W,C]."""
    _req(dy, BF16, "dy")
    N, Ho, Wo, C = dy.shape
    dx = torch.empty((N, Ho * 2, Wo * 2, C), dtype=BF16, device=dy.device)
    check(lib().cg_avgpool2_bwd(_p(dy), N, Ho * 2, Wo * 2, C, _p(dx), _stream()),
          "cg_avgpool2_bwd")
    return dx


def maxpool2_bwd(x, dy):
    _req(x, BF16, "x")
    _req(dy, BF16, "dy")
    N, H, W, C = x.shape
    dx = torch.empty_like(x)
    check(lib().cg_maxpool2_bwd(_p(x), _p(dy), N, H, W, C, _p(dx), _stream()), "cg_maxpool2_bwd")
    return dx


def spatial_reduce(x, gate, scale):
    """x [N,H,W,C] (or [N,HW,C]) -> [N,C]: scale * sum_hw x * (gate>0)."""
    _req(x, BF16, "x")
    _req(gate, BF16, "gate", True)
    N, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * C)
    out = torch.empty((N, C), dtype=BF16, device=x.device)
    check(lib().cg_spatial_reduce(_p(x), _p(gate), N, HW, C, float(scale), _p(out), _stream()),
          "cg_spatial_reduce")
    return out


def dot_f32(a, b):
    """sum(a * b) of two fp32 tensors of one shape -> fp32 [1] (one workgroup: weight-sized inputs)."""
    _req(a, F32, "a")
    _req(b, F32, "b")
    if a.numel() != b.numel():
        raise ValueError("dot_f32: operands differ in size")
    out = torch.empty((1,), dtype=F32, device=a.device)
    check(lib().cg_dot_f32(_p(a), _p(b), a.numel(), _p(out), _stream()), "cg_dot_f32")
    return out


def pooled_head_supported(HW, C):
    return bool(lib().cg_pooled_head_supported(int(HW), int(C)))


def pooled_head_fwd(x3, w, bias, scale):
    """x3 [N,HW,C] bf16, w [C] fp32, bias [1] fp32 or None -> (logit [N,1] fp32, pooled [N,C] bf16):
    relu -> scale * sum over HW -> linear(C -> 1) in one launch (cg_pooled_head_fwd)."""
    _req(x3, BF16, "x")
    _req(w, F32, "w")
    _req(bias, F32, "bias", True)
    N, HW, C = x3.shape
    if w.numel() != C:
        raise ValueError("w must have %d elements" % C)
    pooled = torch.empty((N, C), dtype=BF16, device=x3.device)
    logit = torch.empty((N, 1), dtype=F32, device=x3.device)
    check(lib().cg_pooled_head_fwd(_p(x3), N, HW, C, float(scale), _p(w), _p(bias), _p(pooled),
                                   _p(logit), _stream()), "cg_pooled_head_fwd")
    return logit, pooled


def pooled_head_bwd(x3, w, scale, pooled, dlogit=None, dpooled=None, want_dw=True, want_dbias=True):
    """-> (dx [N,HW,C] bf16, dw [C] fp32 or None, dbias [1] fp32 or None)."""
    _req(x3, BF16, "x")
    _req(w, F32, "w")
    _req(pooled, BF16, "pooled")
    _req(dlogit, F32, "dlogit", True)
    _req(dpooled, BF16, "dpooled", True)
    N, HW, C = x3.shape
    want_dw = bool(want_dw and dlogit is not None)
    want_dbias = bool(want_dbias and dlogit is not None)
    dx = torch.empty_like(x3)
    dw = torch.empty((C,), dtype=F32, device=x3.device) if want_dw else None
    db = torch.empty((1,), dtype=F32, device=x3.device) if want_dbias else None
    ws = _ws(lib().cg_pooled_head_bwd_workspace_bytes(N, C), x3)
    check(lib().cg_pooled_head_bwd(_p(x3), N, HW, C, float(scale), _p(w), _p(dlogit), _p(dpooled),
                                   _p(pooled), _p(dx), _p(dw), _p(db), _p(ws), ws.numel(), _stream()),
          "cg_pooled_head_bwd")
    return dx, dw, db


def spatial_reduce_bwd(gate, dout, shape, scale):
    _req(dout, BF16, "dout")
    _req(gate, BF16, "gate", True)
    N, C = shape[0], shape[-1]
    HW = 1
    for s in shape[1:-1]:
        HW *= s
    dx = torch.empty(shape, dtype=BF16, device=dout.device)
    check(lib().cg_spatial_reduce_bwd(_p(gate), _p(dout), N, HW, C, float(scale), _p(dx),
                                      _stream()), "cg_spatial_reduce_bwd")
    return dx


def head(x, kind):
    _req(x, F32, "x")
    y = torch.empty_like(x)
    check(lib().cg_head(_p(x), kind, _p(y), x.numel(), _stream()), "cg_head")
    return y


def head_bwd(y, kind, dy):
    _req(y, F32, "y")
    if dy.dtype not in (F32, BF16):
        raise ValueError("dy must be fp32 or bf16")
    dx = torch.empty(y.shape, dtype=BF16, device=y.device)
    check(lib().cg_head_bwd(_p(y), kind, _p(dy.contiguous()), int(dy.dtype == F32), _p(dx),
                            y.numel(), _stream()), "cg_head_bwd")
    return dx


def cast_f32_to_bf16(x, a=1.0, b=0.0, out=None):
    _req(x, F32, "x")
    y = out if out is not None else torch.empty(x.shape, dtype=BF16, device=x.device)
    if a == 1.0 and b == 0.0:
        check(lib().cg_cast_f32_to_bf16(_p(x), _p(y), x.numel(), _stream()), "cg_cast_f32_to_bf16")
    else:
        check(lib().cg_affine_f32_to_bf16(_p(x), float(a), float(b), _p(y), x.numel(), _stream()),
              "cg_affine_f32_to_bf16")
    return y


def cast_bf16_to_f32(x):
    _req(x, BF16, "x")
    y = torch.empty(x.shape, dtype=F32, device=x.device)
    check(lib().cg_cast_bf16_to_f32(_p(x), _p(y), x.numel(), _stream()), "cg_cast_bf16_to_f32")
    return y


def layer_norm_fwd(x3, gamma, beta, eps=1e-12):
    """x3 bf16 [N, M, C] -> (y, mean [N], rstd [N])."""
    _req(x3, BF16, "x")
    _req(gamma, F32, "gamma")
    _req(beta, F32, "beta")
    N, M, C = x3.shape
    y = torch.empty_like(x3)
    mean = torch.empty((N,), dtype=F32, device=x3.device)
    rstd = torch.empty((N,), dtype=F32, device=x3.device)
    check(lib().cg_layer_norm_fwd(_p(x3), N, M, C, _p(gamma), _p(beta), float(eps), _p(y), _p(mean),
                                  _p(rstd), _stream()), "cg_layer_norm_fwd")
    return y, mean, rstd


def layer_norm_bwd(x3, dy3, mean, rstd, gamma, want_params=True):
    """-> (dx, dgamma, dbeta); the parameter gradients are None unless want_params."""
    _req(x3, BF16, "x")
    _req(dy3, BF16, "dy")
    N, M, C = x3.shape
    dx = torch.empty_like(x3)
    dg = torch.empty((C,), dtype=F32, device=x3.device) if want_params else None
    db = torch.empty((C,), dtype=F32, device=x3.device) if want_params else None
    ws = _ws(lib().cg_layer_norm_bwd_workspace_bytes(N, C), x3)
    check(lib().cg_layer_norm_bwd(_p(x3), _p(dy3), _p(mean), _p(rstd), _p(gamma), N, M, C, _p(dx),
                                  _p(dg), _p(db), _p(ws), ws.numel(), _stream()),
          "cg_layer_norm_bwd")
    return dx, dg, db


def layer_norm_bwd_bwd(x3, dy3, u3, mean, rstd, gamma, want_dgamma=True):
    """Second order of layer_norm_bwd: upstream u = dL/d(dx) -> (d_dy, d_x, d_gamma or None)."""
    for t, nm in ((x3, "x"), (dy3, "dy"), (u3, "u")):
        _req(t, BF16, nm)
    N, M, C = x3.shape
    d_dy = torch.empty_like(x3)
    d_x = torch.empty_like(x3)
    dg = torch.empty((C,), dtype=F32, device=x3.device) if want_dgamma else None
    ws = _ws(lib().cg_layer_norm_bwd_bwd_workspace_bytes(N, C), x3)
    check(lib().cg_layer_norm_bwd_bwd(_p(x3), _p(dy3), _p(u3), _p(mean), _p(rstd), _p(gamma), N, M, C,
                                      _p(d_dy), _p(d_x), _p(dg), _p(ws), ws.numel(), _stream()),
          "cg_layer_norm_bwd_bwd")
    return d_dy, d_x, dg


def colsum(x2):
    _req(x2, BF16, "x")
    rows, C = x2.shape
    out = torch.empty((C,), dtype=F32, device=x2.device)
    ws = _ws(lib().cg_colsum_workspace_bytes(rows, C), x2)
    check(lib().cg_colsum(_p(x2), rows, C, _p(out), _p(ws), ws.numel(), _stream()), "cg_colsum")
    return out


def rowdot(a, b):
    _req(a, BF16, "a")
    _req(b, BF16, "b")
    B, C = a.shape
    out = torch.empty((B, 1), dtype=F32, device=a.device)
    check(lib().cg_rowdot(_p(a), _p(b), B, C, _p(out), _stream()), "cg_rowdot")
    return out


def rowdot_bwd(a, b, dout, want_da=True, want_db=True):
    _req(dout, F32, "dout")
    B, C = a.shape
    da = torch.empty_like(a) if want_da else None
    db = torch.empty_like(b) if want_db else None
    check(lib().cg_rowdot_bwd(_p(a), _p(b), _p(dout), B, C, _p(da), _p(db), _stream()),
          "cg_rowdot_bwd")
    return da, db


def one_hot(labels, K):
    if labels.dtype != torch.int32 or not labels.is_cuda:
        raise ValueError("labels must be int32 on the GPU")
    B = labels.numel()
    out = torch.empty((B, K), dtype=BF16, device=labels.device)
    check(lib().cg_one_hot(_p(labels.contiguous()), B, K, _p(out), _stream()), "cg_one_hot")
    return out


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def attention_fwd(theta, phi, g):
    for t, nm in ((theta, "theta"), (phi, "phi"), (g, "g")):
        _req(t, BF16, nm)
    B, Lq, Dk = theta.shape
    _, Lk, Dv = g.shape
    out = torch.empty((B, Lq, Dv), dtype=BF16, device=theta.device)
    lse = torch.empty((B, Lq), dtype=F32, device=theta.device)
    check(lib().cg_attention_fwd(_p(theta), _p(phi), _p(g), B, Lq, Lk, Dk, Dv, _p(out), _p(lse),
                                 _stream()), "cg_attention_fwd")
    return out, lse


def attention_bwd(theta, phi, g, out, lse, dout):
    _req(dout, BF16, "dout")
    B, Lq, Dk = theta.shape
    _, Lk, Dv = g.shape
    dtheta, dphi, dg = torch.empty_like(theta), torch.empty_like(phi), torch.empty_like(g)
    ws = _ws(lib().cg_attention_bwd_workspace_bytes(B, Lq, Lk, Dk, Dv), theta)
    check(lib().cg_attention_bwd(_p(theta), _p(phi), _p(g), _p(out), _p(lse), _p(dout), B, Lq, Lk,
                                 Dk, Dv, _p(dtheta), _p(dphi), _p(dg), _p(ws), ws.numel(),
                                 _stream()), "cg_attention_bwd")
    return dtheta, dphi, dg


# ------------------------------------------------------------------------------------------------
# losses / penalties
# ------------------------------------------------------------------------------------------------
LOSS_KINDS = {"non_saturating": 0, "wasserstein": 1, "least_squares": 2, "hinge": 3}


def gan_loss(kind, logits):
    """logits fp32 [2B(,1)] -> (losses[4], dlogits_d [2B], dlogits_g [2B])."""
    _req(logits, F32, "logits")
    B = logits.numel() // 2
    losses = torch.empty((4,), dtype=F32, device=logits.device)
    dd = torch.empty((2 * B,), dtype=F32, device=logits.device)
    dg = torch.empty((2 * B,), dtype=F32, device=logits.device)
    check(lib().cg_gan_loss(kind, _p(logits), B, _p(losses), _p(dd), _p(dg), _stream()),
          "cg_gan_loss")
    return losses, dd, dg


def softmax_xent_eps(logits, labels, eps=1e-10):
    """(-mean log(softmax(logits)[label] + eps) as a [1] tensor, dlogits [n,k])."""
    _req(logits, F32, "logits")
    if labels.dtype != torch.int32:
        raise TypeError("labels must be int32")
    n, k = logits.shape
    loss = torch.empty((1,), dtype=F32, device=logits.device)
    dlogits = torch.empty_like(logits)
    check(lib().cg_softmax_xent_eps(_p(logits), _p(labels), n, k, float(eps), _p(loss),
                                    _p(dlogits), _stream()), "cg_softmax_xent_eps")
    return loss, dlogits


def s3gan_labels(aux_logits, y, soft):
    """(y_out bf16 [n,k], is_label_available fp32 [n]): see cg_s3gan_labels."""
    _req(y, BF16, "y")
    _req(aux_logits, F32, "aux_logits", True)
    n, k = y.shape
    y_out = torch.empty_like(y)
    avail = torch.empty((n,), dtype=F32, device=y.device)
    check(lib().cg_s3gan_labels(_p(aux_logits), _p(y), n, k, int(bool(soft)), _p(y_out),
                                _p(avail), _stream()), "cg_s3gan_labels")
    return y_out, avail


def softmax_xent_weighted(logits, labels, weights):
    """(loss [1], dlogits [n,k]) of tf.losses.softmax_cross_entropy(labels, logits, weights)."""
    _req(logits, F32, "logits")
    _req(labels, BF16, "labels")
    _req(weights, F32, "weights")
    n, k = logits.shape
    loss = torch.empty((1,), dtype=F32, device=logits.device)
    dlogits = torch.empty_like(logits)
    check(lib().cg_softmax_xent_weighted(_p(logits), _p(labels), _p(weights), n, k, _p(loss),
                                         _p(dlogits), _stream()), "cg_softmax_xent_weighted")
    return loss, dlogits


def interpolate(x, x_fake, alpha):
    _req(x, F32, "x")
    _req(x_fake, F32, "x_fake")
    _req(alpha, F32, "alpha")
    B = x.shape[0]
    per = x.numel() // B
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(lib().cg_interpolate(_p(x), _p(x_fake), _p(alpha), B, per, _p(out), _stream()),
          "cg_interpolate")
    return out


def moments_f32(x):
    """[sum x, sum x^2] of an fp32 tensor (device tensor of 2 floats)."""
    _req(x, F32, "x")
    sums = torch.empty((2,), dtype=F32, device=x.device)
    ws = _ws(lib().cg_moments_workspace_bytes(), x)
    check(lib().cg_moments_f32(_p(x), x.numel(), _p(sums), _p(ws), ws.numel(), _stream()),
          "cg_moments_f32")
    return sums


def dragan_perturb(x, u, sums, a=1.0, b=0.0):
    """clip(x + std(x) * (u - 0.5), 0, 1) * a + b -> bf16 (penalty_lib.py:46-49)."""
    _req(x, F32, "x")
    _req(u, F32, "u")
    _req(sums, F32, "sums")
    if u.numel() != x.numel():
        raise ValueError("dragan_perturb: u has the wrong number of elements")
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(lib().cg_dragan_perturb(_p(x), _p(u), _p(sums), x.numel(), float(a), float(b), _p(out),
                                  _stream()), "cg_dragan_perturb")
    return out


def gradient_penalty(g):
    _req(g, F32, "g")
    B = g.shape[0]
    per = g.numel() // B
    slopes = torch.empty((B,), dtype=F32, device=g.device)
    pen = torch.empty((1,), dtype=F32, device=g.device)
    check(lib().cg_gradient_penalty(_p(g), B, per, _p(slopes), _p(pen), _stream()),
          "cg_gradient_penalty")
    return slopes, pen


def gradient_penalty_bwd(g, slopes, upstream):
    _req(upstream, F32, "upstream", True)
    B = g.shape[0]
    per = g.numel() // B
    dg = torch.empty(g.shape, dtype=BF16, device=g.device)
    check(lib().cg_gradient_penalty_bwd(_p(g), _p(slopes), _p(upstream), B, per, _p(dg),
                                        _stream()), "cg_gradient_penalty_bwd")
    return dg


# ------------------------------------------------------------------------------------------------
# optimiser / counters / rng
# ------------------------------------------------------------------------------------------------
class AdamTable(object):
    """Device-resident cgAdamEntry table for a fixed list of (param, grad, m, v, ema) tensors.

    The table lives at a stable device address; set_grads() re-points the gradient column (the
    autograd engine hands out fresh gradient tensors every step) and re-uploads it.  While a
    hipGraph is being captured no host-to-device copy may be recorded, so the upload is deferred:
    the kernels only read the table when the graph is replayed -- call flush() after capture.  The
    table buffer of a captured update must be allocated BEFORE the capture (`table=`)."""

    def __init__(self, params, grads, ms, vs, emas=None, table=None):
        n = len(params)
        self.entries = (_lib.AdamEntry * n)()
        chunk = 0
        self._offs = []
        off = 0
        for i in range(n):
            for t in (params[i], ms[i], vs[i]):
                _req(t, F32, "adam tensor")
            e = self.entries[i]
            e.param = params[i].data_ptr()
            e.m, e.v = ms[i].data_ptr(), vs[i].data_ptr()
            e.ema = emas[i].data_ptr() if emas is not None and emas[i] is not None else None
            e.n = params[i].numel()
            e.chunk_begin = chunk
            chunk += (e.n + _lib.ADAM_CHUNK - 1) // _lib.ADAM_CHUNK
            self._offs.append(off)
            off += e.n
        self.n = n
        self.total_chunks = chunk
        self.total_elems = off
        self.device = params[0].device
        if table is None:
            if torch.cuda.is_current_stream_capturing():
                # memory handed out during a capture is recycled among the captured kernels: an
                # earlier temporary of the same graph may own these bytes, and replaying it would
                # overwrite the (uploaded-once) table
                raise RuntimeError("AdamTable: pass a table buffer allocated before the capture")
            table = torch.empty(ctypes.sizeof(self.entries), dtype=torch.uint8, device=self.device)
        if table.numel() < ctypes.sizeof(self.entries) or table.dtype != torch.uint8:
            raise ValueError("AdamTable: table buffer too small")
        self.table = table
        self._offsets = None
        self._keep = (params, ms, vs, emas)
        self._grad_ptrs = None
        self.dirty = False
        self.set_grads(grads)

    @staticmethod
    def table_bytes(n_entries):
        return ctypes.sizeof(_lib.AdamEntry) * n_entries

    @property
    def offsets(self):
        if self._offsets is None:
            self._offsets = torch.tensor(self._offs, dtype=torch.int64).to(self.device)
        return self._offsets

    def set_grads(self, grads):
        ptrs = tuple(g.data_ptr() for g in grads)
        for g, e in zip(grads, self.entries):
            _req(g, F32, "grad")
            if g.numel() != e.n:
                raise ValueError("gradient has %d elements, parameter %d" % (g.numel(), e.n))
        self._keep_grads = grads
        if ptrs == self._grad_ptrs:
            return
        for p, e in zip(ptrs, self.entries):
            e.grad = p
        self._grad_ptrs = ptrs
        self.dirty = True
        if not torch.cuda.is_current_stream_capturing():
            self.flush()

    def flush(self):
        if self.dirty:
            host = torch.frombuffer(bytearray(bytes(self.entries)), dtype=torch.uint8)
            self.table.copy_(host)
            self.dirty = False

    def adam(self, lr, beta1, beta2, eps, grad_scale, step, ema_decay=0.0, ema_start=0):
        check(lib().cg_adam_multi(_p(self.table), self.n, self.total_chunks, float(lr),
                                  float(beta1), float(beta2), float(eps), float(grad_scale),
                                  _p(step), float(ema_decay), int(ema_start), _stream()),
              "cg_adam_multi")

    def gather(self, flat):
        _req(flat, F32, "flat")
        check(lib().cg_multi_gather(_p(self.table), _p(self.offsets), self.n, self.total_chunks,
                                    _p(flat), _stream()), "cg_multi_gather")

    def scatter(self, flat):
        _req(flat, F32, "flat")
        check(lib().cg_multi_scatter(_p(self.table), _p(self.offsets), self.n, self.total_chunks,
                                     _p(flat), _stream()), "cg_multi_scatter")


def counter_add(counter, inc=1):
    if counter.dtype != torch.int64:
        raise ValueError("counter must be int64")
    check(lib().cg_counter_add(_p(counter), int(inc), _stream()), "cg_counter_add")


def random(kind, lo, hi, seed, op_id, stream_id, step, shape, device):
    """kind 0 uniform[lo,hi), 1 normal(mean=lo, std=hi)."""
    out = torch.empty(shape, dtype=F32, device=device)
    check(lib().cg_random(kind, float(lo), float(hi), int(seed) & (2 ** 64 - 1), int(op_id),
                          int(stream_id), _p(step), _p(out), out.numel(), _stream()), "cg_random")
    return out


def random_labels(K, seed, op_id, stream_id, step, n, device):
    out = torch.empty((n,), dtype=torch.int32, device=device)
    check(lib().cg_random_labels(int(K), int(seed) & (2 ** 64 - 1), int(op_id), int(stream_id),
                                 _p(step), _p(out), n, _stream()), "cg_random_labels")
    return out


# ------------------------------------------------------------------------------------------------
# FID / IS statistics
# ------------------------------------------------------------------------------------------------
def mean_cov_f64(x):
    _req(x, F32, "x")
    n, d = x.shape
    mean = torch.empty((d,), dtype=F64, device=x.device)
    cov = torch.empty((d, d), dtype=F64, device=x.device)
    ws = _ws(lib().cg_mean_cov_workspace_bytes(n, d), x)
    check(lib().cg_mean_cov_f64(_p(x), n, d, _p(mean), _p(cov), _p(ws), ws.numel(), _stream()),
          "cg_mean_cov_f64")
    return mean, cov


def gemm_f64(a, b, ta=False, tb=False, alpha=1.0, eye=0.0):
    """alpha * op(a) @ op(b) + eye * I (fp64)."""
    _req(a, F64, "a")
    _req(b, F64, "b")
    m, k = (a.shape[1], a.shape[0]) if ta else a.shape
    n = b.shape[0] if tb else b.shape[1]
    c = torch.empty((m, n), dtype=F64, device=a.device)
    if alpha == 1.0 and eye == 0.0:
        check(lib().cg_gemm_f64(_p(a), _p(b), _p(c), m, n, k, int(ta), int(tb), _stream()),
              "cg_gemm_f64")
    else:
        check(lib().cg_gemm_f64_ex(_p(a), _p(b), _p(c), m, n, k, int(ta), int(tb), float(alpha),
                                   float(eye), _stream()), "cg_gemm_f64_ex")
    return c


def axpby_eye_f64(a, alpha, eye):
    """alpha * a + eye * I for a square fp64 matrix."""
    _req(a, F64, "a")
    out = torch.empty_like(a)
    check(lib().cg_axpby_eye_f64(_p(a), float(alpha), float(eye), _p(out), a.shape[0], _stream()),
          "cg_axpby_eye_f64")
    return out


def mat_stats_f64(a):
    """Device tensor [trace(a), sum(a * a), min diagonal entry] of a square fp64 matrix."""
    _req(a, F64, "a")
    out = torch.empty((3,), dtype=F64, device=a.device)
    ws = _ws(lib().cg_mat_stats_workspace_bytes(), a)
    check(lib().cg_mat_stats_f64(_p(a), a.shape[0], _p(out), _p(ws), ws.numel(), _stream()),
          "cg_mat_stats_f64")
    return out


def poly3_kernel_sums_f64(gram, dim):
    """[sum, trace] of (gram / dim + 1)^3 for an fp64 Gram block (device tensor of 2 doubles)."""
    _req(gram, F64, "gram")
    m, n = gram.shape
    out = torch.empty((2,), dtype=F64, device=gram.device)
    ws = _ws(lib().cg_poly3_kernel_workspace_bytes(), gram)
    check(lib().cg_poly3_kernel_sums_f64(_p(gram), m, n, 1.0 / float(dim), _p(out), _p(ws),
                                         ws.numel(), _stream()), "cg_poly3_kernel_sums_f64")
    return out


def rowscale_f64(a, scale):
    _req(a, F64, "a")
    _req(scale, F64, "scale")
    out = torch.empty_like(a)
    check(lib().cg_rowscale_f64(_p(a), _p(scale), _p(out), a.shape[0], a.shape[1], _stream()),
          "cg_rowscale_f64")
    return out


def spectral_sqrt_f64(w, eps, want_values=True):
    """(f, sum f) with f = sign(w) * (|w| if |w| < eps else sqrt|w|), both on the device."""
    _req(w, F64, "w")
    f = torch.empty_like(w) if want_values else None
    total = torch.empty((1,), dtype=F64, device=w.device)
    check(lib().cg_spectral_sqrt_f64(_p(w), w.numel(), float(eps), _p(f), _p(total), _stream()),
          "cg_spectral_sqrt_f64")
    return f, total


def spectral_root_scale_f64(w, eps):
    """f(|w|) / w^2 (0 where w = 0) on the device: the row weights of G^T diag(.) G (cg_spectral_root_scale_f64)."""
    _req(w, F64, "w")
    d = torch.empty_like(w)
    check(lib().cg_spectral_root_scale_f64(_p(w), w.numel(), float(eps), _p(d), _stream()),
          "cg_spectral_root_scale_f64")
    return d


def fid_combine_f64(sigma, sigma_v, mean, mean_v, sqrt_trace):
    """tr(sigma) + tr(sigma_v) - 2 sqrt_trace + |mean - mean_v|^2 as a device scalar [1]."""
    for t, nm in ((sigma, "sigma"), (sigma_v, "sigma_v"), (mean, "mean"), (mean_v, "mean_v"),
                  (sqrt_trace, "sqrt_trace")):
        _req(t, F64, nm)
    out = torch.empty((1,), dtype=F64, device=sigma.device)
    check(lib().cg_fid_combine_f64(_p(sigma), _p(sigma_v), _p(mean), _p(mean_v), sigma.shape[0],
                                   _p(sqrt_trace), _p(out), _stream()), "cg_fid_combine_f64")
    return out


def syevj_f64(a, max_sweeps=30, tol=1e-15, want_vectors=True):
    """Destroys `a`.  Returns (w [d], v [d,d] with eigenvectors as ROWS).  want_vectors=False (block
    form only: d >= 256, d % 64 == 0; otherwise the vectors are computed and dropped): (|w|, None)."""
    _req(a, F64, "a")
    d = a.shape[0]
    w = torch.empty((d,), dtype=F64, device=a.device)
    if not want_vectors and not (d >= 256 and d % 64 == 0):
        want_vectors = True
    v = torch.empty((d, d), dtype=F64, device=a.device) if want_vectors else None
    ws = _ws(lib().cg_syevj_workspace_bytes(d), a)
    check(lib().cg_syevj_f64(_p(a), d, _p(w), _p(v), int(max_sweeps), float(tol), _p(ws),
                             ws.numel(), _stream()), "cg_syevj_f64")
    return w, v


def sytrd_eigvals_f64(a):
    """Destroys `a` ([n, n] symmetric fp64, n <= 4096).  Returns (w [n] ascending, |a|_F [1]): the
    eigenvalues by Householder tridiagonalisation + bisection -- absolute accuracy c n u |a|_F."""
    _req(a, F64, "a")
    n = a.shape[0]
    if a.dim() != 2 or a.shape[1] != n:
        raise ValueError("a must be square")
    w = torch.empty((n,), dtype=F64, device=a.device)
    fro = torch.empty((1,), dtype=F64, device=a.device)
    ws = _ws(lib().cg_sytrd_eigvals_workspace_bytes(n), a)
    check(lib().cg_sytrd_eigvals_f64(_p(a), n, _p(w), _p(fro), _p(ws), ws.numel(), _stream()),
          "cg_sytrd_eigvals_f64")
    return w, fro


def spectral_sqrt_bound_f64(w, eps, delta_f_rel, delta_2_rel, fro):
    """[sum_i f(|w_i|), bound on its error when w are the eigenvalues of A + E with |E|_F <= delta_f_rel *
    fro and |E|_2 <= delta_2_rel * fro] (device, fp64); f(s) = s < eps ? s : sqrt(s)."""
    _req(w, F64, "w")
    _req(fro, F64, "fro")
    out = torch.empty((2,), dtype=F64, device=w.device)
    check(lib().cg_spectral_sqrt_bound_f64(_p(w), w.numel(), float(eps), float(delta_f_rel),
                                           float(delta_2_rel), _p(fro), _p(out), _stream()),
          "cg_spectral_sqrt_bound_f64")
    return out


def inception_score_f64(logits):
    _req(logits, F32, "logits")
    n, k = logits.shape
    score = torch.empty((1,), dtype=F64, device=logits.device)
    ws = _ws(lib().cg_inception_score_workspace_bytes(n, k), logits)
    check(lib().cg_inception_score_f64(_p(logits), n, k, _p(score), _p(ws), ws.numel(),
                                       _stream()), "cg_inception_score_f64")
    return score


def inception_preprocess(x, Ho=299, Wo=299):
    _req(x, F32, "x")
    N, H, W, C = x.shape
    y = torch.empty((N, Ho, Wo, C), dtype=BF16, device=x.device)
    check(lib().cg_inception_preprocess(_p(x), N, H, W, C, Ho, Wo, _p(y), _stream()),
          "cg_inception_preprocess")
    return y


def pool2d(x, k, s, p, kind, Ho, Wo):
    _req(x, BF16, "x")
    N, H, W, C = x.shape
    y = torch.empty((N, Ho, Wo, C), dtype=BF16, device=x.device)
    check(lib().cg_pool2d(_p(x), N, H, W, C, k, s, p, kind, Ho, Wo, _p(y), _stream()), "cg_pool2d")
    return y


# ------------------------------------------------------------------------------------------------
# calibration microbenchmarks (bench.py: measured roofline denominators)
# ------------------------------------------------------------------------------------------------
def calibrate(device, mfma_ms=50.0, copy_mb=1024):
    """{"mfma_bf16_tflops", "hbm_copy_gbs"} of THIS box, now: a pure MFMA loop of about `mfma_ms`
    milliseconds on random operands and a float4 copy of `copy_mb` MiB (read + write counted),
    each timed with stream events after one untimed run."""
    sink = torch.zeros(4, dtype=F32, device=device)
    fl = ctypes.c_double(0.0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def mfma(iters):
        check(lib().cg_calib_mfma_bf16(2048, int(iters), _p(sink), ctypes.byref(fl), _stream()),
              "cg_calib_mfma_bf16")
    mfma(200)
    torch.cuda.synchronize()
    # 2048 blocks x 4 waves x 8 MFMAs x 32 cycles per round over 1024 SIMDs: 2048 cycles per round
    iters = max(200, int(mfma_ms * 1e-3 * 2.0e9 / 2048))
    ev[0].record()
    mfma(iters)
    ev[1].record()
    n = int(copy_mb) << 20
    src = torch.empty(n, dtype=torch.uint8, device=device).random_(0, 255)
    dst = torch.empty_like(src)
    check(lib().cg_calib_copy(_p(src), _p(dst), n, _stream()), "cg_calib_copy")
    ev[2].record()
    for _ in range(4):
        check(lib().cg_calib_copy(_p(src), _p(dst), n, _stream()), "cg_calib_copy")
    ev[3].record()
    torch.cuda.synchronize()
    t_m = ev[0].elapsed_time(ev[1]) * 1e-3
    t_c = ev[2].elapsed_time(ev[3]) * 1e-3
    out = {"mfma_bf16_tflops": fl.value / t_m / 1e12, "mfma_ms": t_m * 1e3,
           "hbm_copy_gbs": 4 * 2.0 * n / t_c / 1e9, "copy_mb": int(copy_mb)}
    del src, dst
    # (1) the same MFMA loop on ZERO operands; (2) what this library's own main loop reaches on a
    # plain GEMM: 8192 x 8192 x 8192 as a 1x1 convolution (8192 pixels, 8192 -> 8192 channels)
    fz = ctypes.c_double(0.0)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record()
    check(lib().cg_calib_mfma_bf16_zero(2048, int(iters), _p(sink), ctypes.byref(fz), _stream()),
          "cg_calib_mfma_bf16_zero")
    e[1].record()
    geom = geom_conv_same(8, 32, 32, 8192, 8192, 1, 1, 1, 1)
    x = torch.randn(8, 32, 32, 8192, device=device).to(BF16)
    w = (torch.randn(8192, 8192, device=device) * 0.01).to(BF16).reshape(8192, 8192)   # [Co][K] image
    gconv(geom, x, w)
    e[2].record()
    for _ in range(3):
        gconv(geom, x, w)
    e[3].record()
    torch.cuda.synchronize()
    out["mfma_zero_tflops"] = fz.value / (e[0].elapsed_time(e[1]) * 1e-3) / 1e12
    out["gemm_tflops"] = 3 * 2.0 * 8192.0 ** 3 / (e[2].elapsed_time(e[3]) * 1e-3) / 1e12
    return out


# ------------------------------------------------------------------------------------------------
# kernel-family timing (bench.py roofline leg)
# ------------------------------------------------------------------------------------------------
def prof_enable(on):
    check(lib().cg_prof_enable(int(on)), "cg_prof_enable")


def prof_reset():
    check(lib().cg_prof_reset(), "cg_prof_reset")


def prof_collect():
    """{kernel family: dict(ms, launches, flops, bytes)} accumulated since the last reset."""
    out = {}
    for i in range(lib().cg_prof_family_count()):
        name = lib().cg_prof_family_name(i).decode()
        ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        n = ctypes.c_int64()
        check(lib().cg_prof_collect(i, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl),
                                    ctypes.byref(by)), "cg_prof_collect")
        out[name] = {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}
    return out
