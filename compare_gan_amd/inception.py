"""Batched InceptionV3 feature extractor for FID / Inception score on the HIP kernels.

Reference: compare_gan/eval_utils.py:41-49,165-206 -- the reference runs tfgan.eval.run_inception on
the frozen 2015 Inception graph (`inceptionv1_for_inception_score.pb`, downloaded at run time) and
reads `pool_3:0` ([B, 2048]) and `logits:0` ([B, 1008]) after tfgan.eval.preprocess_image
(bilinear resize to 299x299, (x - 128) / 128).

What is reproduced here: the preprocessing (cg_inception_preprocess, TF1 legacy bilinear) and the
architecture of that graph (stem, mixed .. mixed_10, 8x8 average pool, 1008-way logits; batch norm
folded into conv + bias + ReLU as in the frozen graph).  What cannot be reproduced offline: the
trained weights (no network, no file on disk -- SURVEY.md section 8c).  Weights are therefore
He-normal draws from a fixed seed: FID / IS values are self-consistent (same extractor for real and
fake images) but not comparable with published numbers -- "parity unpinned" for real activations.
`load_weights()` accepts a {name: array} dict in the layouts below, should a weight file exist.

Every arithmetic op is a HIP kernel: cg_gconv (conv + bias + ReLU fused, act_out), cg_pool2d,
cg_spatial_reduce (global average pool), cg_gconv as the logits GEMM.  torch.cat only moves bytes.
"""
import math
import os

import torch

from compare_gan_amd.hip import kernels as K

BF16 = torch.bfloat16
F32 = torch.float32

# ---- architecture table ------------------------------------------------------------------------
# op tuples: ("conv", name, cout, kh, kw, stride, padding) | ("max", k, s) | ("avg3",) and
# ("mixed", name, [branches]) where every branch is a list of ops applied to the block input.


def _c(name, cout, kh=1, kw=1, stride=1, padding="SAME"):
  return ("conv", name, cout, kh, kw, stride, padding)


def _mixed_35(name, pool_proj):
  return ("mixed", name, [
      [_c(name + "/b0_1x1", 64)],
      [_c(name + "/b1_1x1", 48), _c(name + "/b1_5x5", 64, 5, 5)],
      [_c(name + "/b2_1x1", 64), _c(name + "/b2_3x3a", 96, 3, 3), _c(name + "/b2_3x3b", 96, 3, 3)],
      [("avg3",), _c(name + "/b3_pool_1x1", pool_proj)],
  ])


def _mixed_17(name, c7):
  return ("mixed", name, [
      [_c(name + "/b0_1x1", 192)],
      [_c(name + "/b1_1x1", c7), _c(name + "/b1_1x7", c7, 1, 7), _c(name + "/b1_7x1", 192, 7, 1)],
      [_c(name + "/b2_1x1", c7), _c(name + "/b2_7x1a", c7, 7, 1), _c(name + "/b2_1x7a", c7, 1, 7),
       _c(name + "/b2_7x1b", c7, 7, 1), _c(name + "/b2_1x7b", 192, 1, 7)],
      [("avg3",), _c(name + "/b3_pool_1x1", 192)],
  ])


def _mixed_8(name, pool):
  return ("mixed", name, [
      [_c(name + "/b0_1x1", 320)],
      [_c(name + "/b1_1x1", 384), ("split", [[_c(name + "/b1_1x3", 384, 1, 3)],
                                             [_c(name + "/b1_3x1", 384, 3, 1)]])],
      [_c(name + "/b2_1x1", 448), _c(name + "/b2_3x3", 384, 3, 3),
       ("split", [[_c(name + "/b2_1x3", 384, 1, 3)], [_c(name + "/b2_3x1", 384, 3, 1)]])],
      [(pool,), _c(name + "/b3_pool_1x1", 192)],
  ])


SPEC = [
    _c("conv", 32, 3, 3, 2, "VALID"), _c("conv_1", 32, 3, 3, 1, "VALID"), _c("conv_2", 64, 3, 3),
    ("max", 3, 2),
    _c("conv_3", 80), _c("conv_4", 192, 3, 3, 1, "VALID"),
    ("max", 3, 2),
    _mixed_35("mixed", 32), _mixed_35("mixed_1", 64), _mixed_35("mixed_2", 64),
    ("mixed", "mixed_3", [
        [_c("mixed_3/b0_3x3", 384, 3, 3, 2, "VALID")],
        [_c("mixed_3/b1_1x1", 64), _c("mixed_3/b1_3x3a", 96, 3, 3),
         _c("mixed_3/b1_3x3b", 96, 3, 3, 2, "VALID")],
        [("max", 3, 2)],
    ]),
    _mixed_17("mixed_4", 128), _mixed_17("mixed_5", 160), _mixed_17("mixed_6", 160),
    _mixed_17("mixed_7", 192),
    ("mixed", "mixed_8", [
        [_c("mixed_8/b0_1x1", 192), _c("mixed_8/b0_3x3", 320, 3, 3, 2, "VALID")],
        [_c("mixed_8/b1_1x1", 192), _c("mixed_8/b1_1x7", 192, 1, 7), _c("mixed_8/b1_7x1", 192, 7, 1),
         _c("mixed_8/b1_3x3", 192, 3, 3, 2, "VALID")],
        [("max", 3, 2)],
    ]),
    _mixed_8("mixed_9", "avg3"), _mixed_8("mixed_10", "max3s1"),
]
_CHUNK = int(os.environ.get("CGAMD_INCEPTION_BATCH", "512"))   # images per call of transform()
_USE_LD = os.environ.get("CGAMD_INCEPTION_LD", "1") != "0"    # branches write into the block output
_MERGE_HEADS = os.environ.get("CGAMD_INCEPTION_HEADS", "1") != "0"   # sibling 1x1 convolutions as one
_SWAP_POOL = os.environ.get("CGAMD_INCEPTION_SWAP_POOL", "1") != "0"  # avg_pool -> 1x1 as 1x1 -> avg_pool
POOL3_DIM = 2048
NUM_LOGITS = 1008
INPUT_SIZE = 299


def conv_shapes(spec=None, cin=3):
  """[(name, [kh, kw, cin, cout])] of every convolution in execution order, and the final width."""
  out = []

  def run(ops, c):
    for op in ops:
      if op[0] == "conv":
        _, name, cout, kh, kw, _, _ = op
        out.append((name, [kh, kw, c, cout]))
        c = cout
      elif op[0] == "mixed":
        c = sum(run(b, c) for b in op[2])
      elif op[0] == "split":
        c = sum(run(b, c) for b in op[1])
    return c
  c_final = run(SPEC if spec is None else spec, cin)
  return out, c_final


def make_weights(seed=2015, device="cpu"):
  """Seeded He-normal conv kernels (HWIO fp32), small biases, and the logits layer.  device: where
  the draws are made (the stand-in extractor of an offline evaluation draws its 24 M values on the
  GPU: 1.2 s of host sampling + 190 host-to-device copies otherwise, bench.py `extractor_setup_s`);
  the host and device generators give DIFFERENT values for one seed -- tests that compare with the
  oracle pass the same dictionary to both."""
  device = torch.device(device)
  g = torch.Generator(device=device).manual_seed(seed)
  w = {}
  shapes, c_final = conv_shapes()
  assert c_final == POOL3_DIM
  for name, (kh, kw, ci, co) in shapes:
    std = math.sqrt(2.0 / (kh * kw * ci))
    w[name + "/kernel"] = torch.randn((kh, kw, ci, co), generator=g, dtype=F32, device=device) * std
    w[name + "/bias"] = torch.randn((co,), generator=g, dtype=F32, device=device) * 0.05
  w["logits/kernel"] = torch.randn((POOL3_DIM, NUM_LOGITS), generator=g, dtype=F32,
                                   device=device) * (1.0 / math.sqrt(POOL3_DIM))
  w["logits/bias"] = torch.zeros((NUM_LOGITS,), dtype=F32, device=device)
  return w


class InceptionV3(object):
  """pool_3 / logits of [B, H, W, 3] images in [0, 255] (eval_utils.inception_transform)."""

  def __init__(self, device, weights=None, seed=2015):
    self.device = torch.device(device)
    self.load_weights(weights if weights is not None else make_weights(seed, self.device))

  # (producer, consumer) pairs whose intermediate channel count is not a multiple of 32 (48 / 80):
  # the MFMA-tiled kernels slice K in 64-channel blocks and need Ci % 32 == 0 (cg_gconv falls back
  # to the generic gather kernel otherwise, ~10x slower).  The producer gets zero output channels
  # (zero kernel columns, zero bias: relu(0) = 0, and max-pooling zeros gives zeros), the consumer
  # zero kernel rows for them -- the arithmetic on the real channels is unchanged.
  _PAD_PAIRS = [("conv_3", "conv_4")] + [("%s/b1_1x1" % m, "%s/b1_5x5" % m)
                                         for m in ("mixed", "mixed_1", "mixed_2")]

  @classmethod
  def _pad_channels(cls, weights):
    if os.environ.get("CGAMD_INCEPTION_PAD", "1") == "0":
      return weights, {}
    w = dict(weights)
    padded = {}
    for prod, cons in cls._PAD_PAIRS:
      kp, kc = w[prod + "/kernel"], w[cons + "/kernel"]
      co = kp.shape[3]
      cp = (co + 31) // 32 * 32
      if cp == co or kc.shape[2] != co:
        continue
      w[prod + "/kernel"] = torch.cat([kp, kp.new_zeros(kp.shape[:3] + (cp - co,))], dim=3)
      w[prod + "/bias"] = torch.cat([w[prod + "/bias"], kp.new_zeros((cp - co,))])
      w[cons + "/kernel"] = torch.cat(
          [kc, kc.new_zeros(kc.shape[:2] + (cp - co, kc.shape[3]))], dim=2)
      padded[prod] = cp
    return w, padded

  def load_weights(self, weights):
    weights, self._padded_cout = self._pad_channels(
        {k: v.to(F32) for k, v in weights.items()})
    self.weights = {k: v.to(F32).to(self.device).contiguous() for k, v in weights.items()}
    # frozen weights: the MFMA operand images are built once
    self._bt = {}
    for name, v in self.weights.items():
      if name.endswith("/kernel") and v.dim() == 4:
        self._bt[name] = K.weight_prep(v, want_fwd=True, want_bwd=False)[0]
    wl = self.weights["logits/kernel"]
    self._bt["logits/kernel"] = K.weight_prep(wl.reshape(1, 1, *wl.shape), want_fwd=True)[0]
    # Sibling 1x1 convolutions: the branches of a block that START with a unit-stride 1x1 convolution
    # of the block input (three of four in every mixed block) run as ONE convolution with their kernels
    # side by side -- the input is read once instead of three times and the column tiles fill up
    # (17x17x768 -> 192 + 160 + 160: five 128-wide tiles, two of them 62 % full, become 512 columns);
    # the next layer of each branch reads its channel slice of the result (K.gconv_ld).
    self._heads = {}
    if _MERGE_HEADS:
      for op in SPEC:
        if op[0] != "mixed":
          continue
        heads = [(i, b[0]) for i, b in enumerate(op[2])
                 if b[0][0] == "conv" and b[0][3:6] == (1, 1, 1)]
        if len(heads) < 2:
          continue
        # ... and the `avg_pool 3x3 -> 1x1 conv -> ReLU` branch: the pooling (a per-channel linear map,
        # TF 'SAME' counts included) commutes with the 1x1 convolution, so the convolution joins the
        # heads (no bias, no ReLU on its columns: K.gconv_ld's relu_cols) and the pooling runs behind it
        # on a quarter of the channels, adding bias and ReLU (K.pool2d_ld)
        pooled = [(i, b[1]) for i, b in enumerate(op[2])
                  if _SWAP_POOL and len(b) == 2 and b[0] == ("avg3",) and b[1][0] == "conv"
                  and b[1][3:6] == (1, 1, 1)]
        ks = [self.weights[h[1] + "/kernel"] for _, h in heads + pooled]
        bs = [self.weights[h[1] + "/bias"] for _, h in heads] + \
             [torch.zeros_like(self.weights[h[1] + "/bias"]) for _, h in pooled]
        name = op[1] + "/heads"
        self._bt[name + "/kernel"] = K.weight_prep(torch.cat(ks, dim=3).contiguous(), want_fwd=True)[0]
        self.weights[name + "/bias"] = torch.cat(bs).contiguous()
        cols, off = {}, 0
        for (i, h), k in zip(heads + pooled, ks):
          cols[i] = (off, int(k.shape[3]), h[1] if (i, h) in pooled else None)
          off += int(k.shape[3])
        relu_cols = sum(int(self.weights[h[1] + "/kernel"].shape[3]) for _, h in heads)
        self._heads[op[1]] = (cols, off, relu_cols)

  # -- ops -----------------------------------------------------------------------------------------
  # Every block of the graph ends in a concatenation along the channels.  Here the block's output is
  # allocated once and each branch's LAST op writes its channels straight into its slice of it
  # (K.gconv_ld: the one-tap MFMA kernel with a pixel pitch on its output; a pooling branch copies)
  # -- the 2.1 ms of concatenation copies per 512-image batch are gone.  _emit(ops, x, dst) runs a
  # list of ops; with dst (a channel slice) given, the result of the last one lands there.
  def _geom(self, x, op):
    _, name, cout, kh, kw, stride, padding = op
    cout = self._padded_cout.get(name, cout)
    n, h, w_, ci = x.shape
    if padding == "SAME":
      return K.geom_conv_same(n, h, w_, ci, cout, kh, kw, stride, 1)
    ho, wo = (h - kh) // stride + 1, (w_ - kw) // stride + 1
    return K.make_geom(n, h, w_, ci, ho, wo, cout, kh, kw, stride, 1, 0, 0)

  def _conv(self, x, op, dst=None):
    name = op[1]
    geom = self._geom(x, op)
    bt, bias = self._bt[name + "/kernel"], self.weights[name + "/bias"]
    if not x.is_contiguous() and geom.kh == 3 and geom.kw == 3 and geom.Hin == 8:
      x = x.contiguous()     # 8x8 maps: the four-image halo kernel (dense input) beats the one-tap one
    if dst is None and x.is_contiguous():
      return K.gconv(geom, x, bt, bias=bias, act_out=0.0)
    if dst is None:
      dst = torch.empty((geom.N, geom.Ho, geom.Wo, geom.Co), dtype=BF16, device=x.device)
    if _USE_LD and K.gconv_ld_supported(geom, x.stride(2), dst.stride(2)):
      return K.gconv_ld(geom, x, bt, dst, bias=bias, relu=True)
    dst.copy_(K.gconv(geom, x.contiguous(), bt, bias=bias, act_out=0.0))
    return dst

  @staticmethod
  def _pool(x, kind, k, s, same, dst=None):
    n, h, w_, c = x.shape
    if same:
      p = (k - 1) // 2
      ho, wo = -(-h // s), -(-w_ // s)
    else:
      p = 0
      ho, wo = (h - k) // s + 1, (w_ - k) // s + 1
    if dst is not None and _USE_LD and c % 8 == 0:
      return K.pool2d_ld(x, k, s, p, kind, ho, wo, dst)   # the pooling branch of a reduction block
    y = K.pool2d(x.contiguous(), k, s, p, kind, ho, wo)
    if dst is not None:
      dst.copy_(y)
      return dst
    return y

  def _shape_after(self, ops, h, w_, c):
    """(h, w, c) after a list of ops (no launches): the size of a block's output buffer."""
    for op in ops:
      if op[0] == "conv":
        _, name, cout, kh, kw, stride, padding = op
        c = self._padded_cout.get(name, cout)
        if padding == "SAME":
          h, w_ = -(-h // stride), -(-w_ // stride)
        else:
          h, w_ = (h - kh) // stride + 1, (w_ - kw) // stride + 1
      elif op[0] == "max":
        h, w_ = (h - op[1]) // op[2] + 1, (w_ - op[1]) // op[2] + 1
      elif op[0] in ("mixed", "split"):
        outs = [self._shape_after(b, h, w_, c) for b in (op[2] if op[0] == "mixed" else op[1])]
        h, w_, c = outs[0][0], outs[0][1], sum(o[2] for o in outs)
    return h, w_, c

  def _concat(self, branches, x, dst, block=None):
    n, h, w_, c = x.shape
    outs = [self._shape_after(b, h, w_, c) for b in branches]
    if dst is None:
      dst = torch.empty((n, outs[0][0], outs[0][1], sum(o[2] for o in outs)), dtype=BF16,
                        device=x.device)
    cols, heads = {}, None
    if block in self._heads and x.is_contiguous() and _USE_LD:
      cols, width, relu_cols = self._heads[block]
      geom = K.make_geom(n, h, w_, c, h, w_, width, 1, 1)
      if K.gconv_ld_supported(geom, c, width):
        heads = torch.empty((n, h, w_, width), dtype=BF16, device=x.device)
        K.gconv_ld(geom, x, self._bt[block + "/heads/kernel"], heads,
                   bias=self.weights[block + "/heads/bias"], relu=relu_cols)
      else:
        cols = {}
    off = 0
    for i, (b, o) in enumerate(zip(branches, outs)):
      d = dst[..., off:off + o[2]]
      if i in cols:
        xb = heads[..., cols[i][0]:cols[i][0] + cols[i][1]]
        if cols[i][2] is not None:   # the pooled branch: its 1x1 convolution ran with the heads
          K.pool2d_ld(xb, 3, 1, 1, 1, h, w_, d, bias=self.weights[cols[i][2] + "/bias"], relu=True)
        elif len(b) == 1:
          d.copy_(xb)           # the branch IS its 1x1 convolution: its columns move into the block
        else:
          self._emit(b[1:], xb, d)
      else:
        self._emit(b, x, d)
      off += o[2]
    return dst

  def _emit(self, ops, x, dst=None):
    for i, op in enumerate(ops):
      d = dst if i == len(ops) - 1 else None
      if op[0] == "conv":
        x = self._conv(x, op, d)
        continue
      if op[0] == "mixed":
        x = self._concat(op[2], x, d, block=op[1])
        continue
      if op[0] == "split":
        x = self._concat(op[1], x, d)
        continue
      if op[0] == "max":
        y = self._pool(x, 0, op[1], op[2], False, d)
      elif op[0] == "avg3":
        y = self._pool(x, 1, 3, 1, True, d)
      elif op[0] == "max3s1":
        y = self._pool(x, 0, 3, 1, True, d)
      else:
        raise ValueError("unknown op %r" % (op,))
      x = y
    return x

  def _run(self, ops, x):
    return self._emit(ops, x)

  def features(self, images_0_255):
    """images [B, H, W, 3] fp32 in [0, 255] on the device -> (pool_3 [B, 2048], logits [B, 1008]),
    both fp32."""
    if images_0_255.shape[-1] != 3:
      raise ValueError("Inception expects 3 colour channels, got %d" % images_0_255.shape[-1])
    x = K.inception_preprocess(images_0_255.contiguous(), INPUT_SIZE, INPUT_SIZE)
    x = self._run(SPEC, x)
    n, h, w_, c = x.shape
    pool3 = K.spatial_reduce(x, None, 1.0 / (h * w_))            # [B, 2048] bf16
    geom = K.make_geom(n, 1, 1, c, 1, 1, NUM_LOGITS, 1, 1)
    logits = K.gconv(geom, pool3.reshape(n, 1, 1, c), self._bt["logits/kernel"],
                     bias=self.weights["logits/bias"], out_f32=True).reshape(n, NUM_LOGITS)
    return K.cast_bf16_to_f32(pool3), logits

  def transform(self, images_0_255, batch_size=64):
    """eval_utils.inception_transform_np: batched over a [N, H, W, 3] array (host or device).
    The frozen graph has no cross-sample operation, so the features of an image do not depend on
    how the set is cut into batches: at least _CHUNK images go through per call (the reference's
    batches of 64 leave the 17x17 and 8x8 stages at a few hundred workgroups per launch)."""
    feats, logits = [], []
    n = images_0_255.shape[0]
    batch_size = max(int(batch_size), _CHUNK)
    for i in range(0, n, batch_size):
      batch = images_0_255[i:i + batch_size]
      if not torch.is_tensor(batch):
        batch = torch.from_numpy(batch)
      f, l = self.features(batch.to(self.device, dtype=F32))
      feats.append(f)
      logits.append(l)
    return torch.cat(feats, dim=0), torch.cat(logits, dim=0)
