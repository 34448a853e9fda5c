"""ORACLE -- test infrastructure only (never imported by the product path).

CPU restatement of the reference's generator / discriminator definitions:
  architectures/abstract_arch.py:48-146, resnet_ops.py:70-182, resnet_cifar.py:34-167,
  resnet5.py:36-145, resnet_biggan.py:80-425, resnet_biggan_deep.py:62-433, dcgan.py:39-129,
  sndcgan.py:36-127.
Pinned by the reference's structural tests (variable names/shapes and parameter counts:
architectures/resnet_norm_test.py:39-369, resnet_biggan_test.py:112-154); the forward numerics
are "parity unpinned" (no reference test asserts a value) -- see tests/test_oracle_pins.py.
"""
import math

import torch

from oracle import arch_ops as ops


class ArchConfig(object):
  """The gin surface of G / D (abstract_arch.py:48-56,101-109) + the L1 op configs."""

  def __init__(self, batch_norm_fn=None, spectral_norm=False, layer_norm=False,
               bn_cfg=None, sn_cfg=None, cbn_use_bias=False, sbn_num_hidden=32, **arch_kwargs):
    self.batch_norm_fn = batch_norm_fn      # None | "batch_norm" | "conditional_batch_norm" | ...
    self.spectral_norm = spectral_norm
    self.layer_norm = layer_norm
    self.bn_cfg = bn_cfg or ops.BNConfig()
    self.sn_cfg = sn_cfg or ops.SNConfig()
    self.cbn_use_bias = cbn_use_bias
    self.sbn_num_hidden = sbn_num_hidden
    self.arch_kwargs = arch_kwargs


def _batch_norm(vs, cfg, x, name, z=None, y=None, is_training=True, use_sn=None, relu=False):
  """abstract_arch.py:76-83 / :121-128 dispatch via call_with_accepted_args; relu=True is the
  tf.nn.relu that follows the call in every block (kept inside so that emulate_bf16 rounds once)."""
  fn = cfg.batch_norm_fn
  if fn is None or fn == "no_batch_norm":
    return torch.relu(x) if relu else x
  if use_sn is None:
    use_sn = cfg.spectral_norm
  if fn == "batch_norm":
    return ops.batch_norm(vs, x, is_training, name, cfg.bn_cfg, relu=relu)
  if fn == "conditional_batch_norm":
    return ops.conditional_batch_norm(vs, x, y, is_training, use_sn, name, cfg.bn_cfg, cfg.sn_cfg,
                                      cfg.cbn_use_bias, relu=relu)
  if fn == "self_modulated_batch_norm":
    return ops.self_modulated_batch_norm(vs, x, z, is_training, use_sn, name, cfg.bn_cfg,
                                         cfg.sn_cfg, cfg.sbn_num_hidden, relu=relu)
  raise ValueError("unknown batch_norm_fn %r" % fn)


def _norm_relu(vs, cfg, x, scope, bn_name, ln_name, z, y, is_training):
  """batch_norm -> [layer_norm] -> relu (resnet_ops.py:159-165,169-175; resnet_biggan.py:120-136)."""
  if not cfg.layer_norm:
    return _batch_norm(vs, cfg, x, scope + "/" + bn_name, z=z, y=y, is_training=is_training,
                       relu=True)
  out = _batch_norm(vs, cfg, x, scope + "/" + bn_name, z=z, y=y, is_training=is_training)
  out = ops.layer_norm(vs, out, is_training, scope + "/" + ln_name)
  return torch.relu(out)


def _get_conv(vs, cfg, x, in_ch, out_ch, scale, suffix, scope, kernel=(3, 3), residual=None,
              pool=True):
  """resnet_ops.py:112-134.  residual / pool=False let the caller form pool(conv2 + shortcut),
  which equals the reference's pool(conv2) + pool(shortcut) (average pooling is linear)."""
  if x.shape[-1] != in_ch:
    raise ValueError("Unexpected number of input channels.")
  if scale not in ("up", "down", "none"):
    raise ValueError("Scale: got {}, expected 'up', 'down', or 'none'.".format(scale))
  out = x
  if scale == "up":
    out = ops.unpool(out)
  name = "{}/{}_{}".format(scope, "same" if scale == "none" else scale, suffix)
  out = ops.conv2d(vs, out, out_ch, kernel[0], kernel[1], 1, 1, name, cfg.sn_cfg,
                   use_sn=cfg.spectral_norm, residual=residual)
  if scale == "down" and pool:
    out = vs.q(ops.avg_pool2(out))
  return out


def resnet_block(vs, cfg, x, scope, in_ch, out_ch, scale, is_gen_block, z, y, is_training):
  """resnet_ops.py:136-182 (3x3 shortcut conv, created FIRST)."""
  if x.shape[-1] != in_ch:
    raise ValueError("Unexpected number of input channels.")
  scale1 = scale if is_gen_block else "none"
  scale2 = "none" if is_gen_block else scale
  shortcut = _get_conv(vs, cfg, x, in_ch, out_ch, scale, "conv_shortcut", scope, pool=False)
  out = _norm_relu(vs, cfg, x, scope, "bn1", "ln1", z, y, is_training)
  out = _get_conv(vs, cfg, out, in_ch, out_ch, scale1, "conv1", scope)
  out = _norm_relu(vs, cfg, out, scope, "bn2", "ln2", z, y, is_training)
  # output += shortcut (resnet_ops.py:181), pooled once when the block downsamples
  out = _get_conv(vs, cfg, out, out_ch, out_ch, scale2, "conv2", scope, residual=shortcut,
                  pool=False)
  if scale == "down":
    out = vs.q(ops.avg_pool2(out))
  return out


def biggan_block(vs, cfg, x, scope, in_ch, out_ch, scale, is_gen_block, z, y, is_training,
                 add_shortcut=True):
  """resnet_biggan.py:99-151 (1x1 shortcut, created LAST, optional)."""
  if x.shape[-1] != in_ch:
    raise ValueError("Unexpected number of input channels (expected {}, got {}).".format(
        in_ch, x.shape[-1]))
  scale1 = scale if is_gen_block else "none"
  scale2 = "none" if is_gen_block else scale
  out = _norm_relu(vs, cfg, x, scope, "bn1", "ln1", z, y, is_training)
  out = _get_conv(vs, cfg, out, in_ch, out_ch, scale1, "conv1", scope)
  out = _norm_relu(vs, cfg, out, scope, "bn2", "ln2", z, y, is_training)
  out = _get_conv(vs, cfg, out, out_ch, out_ch, scale2, "conv2", scope)
  if add_shortcut:
    sc_in = x
    if scale == "down":
      # pool(conv1x1(x)) == conv1x1(pool(x)): average pooling commutes with a 1x1 convolution
      sc_in = vs.q(ops.avg_pool2(x))
    out = _get_conv(vs, cfg, sc_in, in_ch, out_ch, scale, "conv_shortcut", scope, kernel=(1, 1),
                    residual=out, pool=False)
  return out


# ------------------------------------------------------------------------------------------------
# resnet_cifar (resnet_cifar.py:34-167)
# ------------------------------------------------------------------------------------------------
def resnet_cifar_generator(vs, cfg, z, y, is_training, image_shape=(32, 32, 3)):
  kw = cfg.arch_kwargs
  hierarchical_z, embed_z, embed_y = (kw.get("hierarchical_z", False), kw.get("embed_z", False),
                                      kw.get("embed_y", False))
  assert image_shape[0] == 32 and image_shape[1] == 32
  s = "generator"
  num_blocks = 3
  z_dim = z.shape[1]
  if embed_z:
    z = ops.linear(vs, z, z_dim, s + "/embed_z", cfg.sn_cfg, use_sn=cfg.spectral_norm)
  if embed_y:
    y = ops.linear(vs, y, z_dim, s + "/embed_y", cfg.sn_cfg, use_sn=cfg.spectral_norm)
  y_per_block = num_blocks * [y]
  if hierarchical_z:
    chunks = torch.chunk(z, num_blocks + 1, dim=1)
    z0, z_per_block = chunks[0], list(chunks[1:])
    if y is not None:
      y_per_block = [torch.cat([zi, y], 1) for zi in z_per_block]
  else:
    z0, z_per_block = z, num_blocks * [z]
  out = ops.linear(vs, z0, 4 * 4 * 256, s + "/fc_noise", cfg.sn_cfg, use_sn=cfg.spectral_norm)
  out = out.reshape(-1, 4, 4, 256)
  for b in range(3):
    out = resnet_block(vs, cfg, out, "%s/B%d" % (s, b + 1), 256, 256, "up", True,
                       z_per_block[b], y_per_block[b], is_training)
  out = _batch_norm(vs, cfg, out, s + "/final_norm", z=z, y=y, is_training=is_training, relu=True)
  out = ops.conv2d(vs, out, image_shape[2], 3, 3, 1, 1, s + "/final_conv", cfg.sn_cfg,
                   use_sn=cfg.spectral_norm, out_f32=True)
  return torch.sigmoid(out)


def resnet_cifar_discriminator(vs, cfg, x, y, is_training):
  project_y = cfg.arch_kwargs.get("project_y", False)
  s = "discriminator"
  colors = x.shape[3]
  if colors not in (1, 3):
    raise ValueError("Number of color channels not supported: {}".format(colors))
  out = x
  for b in range(4):
    out = resnet_block(vs, cfg, out, "%s/B%d" % (s, b + 1), colors if b == 0 else 128, 128,
                       "down" if b <= 1 else "none", False, None, y, is_training)
  out = torch.relu(out)
  h = vs.q(out.mean(dim=(1, 2)))
  logit = ops.linear(vs, h, 1, s + "/disc_final_fc", cfg.sn_cfg, use_sn=cfg.spectral_norm,
                     out_f32=True)
  if project_y:
    if y is None:
      raise ValueError("You must provide class information y to project.")
    emb = ops.linear(vs, y, 128, s + "/embedding_fc", cfg.sn_cfg, use_sn=cfg.spectral_norm,
                     use_bias=False)
    logit = logit + (emb * h).sum(dim=1, keepdim=True)
  return torch.sigmoid(logit), logit, h


# ------------------------------------------------------------------------------------------------
# resnet5 (resnet5.py:36-145)
# ------------------------------------------------------------------------------------------------
def resnet5_generator(vs, cfg, z, y, is_training, image_shape=(128, 128, 3)):
  ch = cfg.arch_kwargs.get("ch", 64)
  channels = cfg.arch_kwargs.get("channels", (8, 8, 4, 4, 2, 1))
  s = "generator"
  seed = 4
  net = ops.linear(vs, z, ch * channels[0] * seed * seed, s + "/fc_noise", cfg.sn_cfg)
  net = net.reshape(-1, seed, seed, ch * channels[0])
  up_layers = math.log2(float(image_shape[0]) / seed)
  if not float(up_layers).is_integer():
    raise ValueError("log2({}/{}) must be an integer.".format(image_shape[0], seed))
  if up_layers < 0 or up_layers > 5:
    raise ValueError("Invalid image_size {}.".format(image_shape[0]))
  up_layers = int(up_layers)
  for b in range(5):
    net = resnet_block(vs, cfg, net, "%s/B%d" % (s, b + 1), ch * channels[b],
                       ch * channels[b + 1], "up" if b < up_layers else "none", True, z, y,
                       is_training)
  net = _batch_norm(vs, cfg, net, s + "/final_norm", z=z, y=y, is_training=is_training, relu=True)
  net = ops.conv2d(vs, net, image_shape[2], 3, 3, 1, 1, s + "/final_conv", cfg.sn_cfg,
                   out_f32=True)
  return torch.sigmoid(net)


def resnet5_discriminator(vs, cfg, x, y, is_training):
  ch = cfg.arch_kwargs.get("ch", 64)
  channels = cfg.arch_kwargs.get("channels", (1, 2, 4, 4, 8, 8))
  s = "discriminator"
  colors = x.shape[3]
  if colors not in (1, 3):
    raise ValueError("Number of color channels not supported: {}".format(colors))
  out = resnet_block(vs, cfg, x, s + "/B0", colors, ch, "down", False, None, y, is_training)
  for b in range(5):
    out = resnet_block(vs, cfg, out, "%s/B%d" % (s, b + 1), ch * channels[b],
                       ch * channels[b + 1], "down", False, None, y, is_training)
  out = torch.relu(out)
  pre = vs.q(out.mean(dim=(1, 2)))
  logit = ops.linear(vs, pre, 1, s + "/disc_final_fc", cfg.sn_cfg, use_sn=cfg.spectral_norm,
                     out_f32=True)
  return torch.sigmoid(logit), logit, pre


# ------------------------------------------------------------------------------------------------
# resnet_biggan (resnet_biggan.py:154-425)
# ------------------------------------------------------------------------------------------------
_G_MULT = {512: [16, 16, 8, 8, 4, 2, 1, 1], 256: [16, 16, 8, 8, 4, 2, 1], 128: [16, 16, 8, 4, 2, 1],
           64: [16, 16, 8, 4, 2], 32: [4, 4, 4, 4]}
_D_MULT = {512: [1, 1, 2, 4, 8, 8, 16, 16], 256: [1, 2, 4, 8, 8, 16, 16], 128: [1, 2, 4, 8, 16, 16],
           64: [2, 4, 8, 16, 16], 32: [2, 2, 2, 2]}


def biggan_generator(vs, cfg, z, y, is_training, image_shape=(128, 128, 3)):
  kw = cfg.arch_kwargs
  ch = kw.get("ch", 96)
  attn_blocks = set(kw.get("blocks_with_attention", "B4").split(","))
  hierarchical_z = kw.get("hierarchical_z", True)
  embed_z, embed_y = kw.get("embed_z", False), kw.get("embed_y", True)
  embed_y_dim, embed_bias = kw.get("embed_y_dim", 128), kw.get("embed_bias", False)
  s = "generator"
  res = image_shape[0]
  if res not in _G_MULT:
    raise ValueError("Unsupported resolution: {}".format(res))
  mult = _G_MULT[res]
  in_ch = [ch * c for c in mult[:-1]]
  out_ch = [ch * c for c in mult[1:]]
  nb = len(in_ch)
  z_dim = z.shape[1]
  if embed_z:
    z = ops.linear(vs, z, z_dim, s + "/embed_z", cfg.sn_cfg, use_sn=False, use_bias=embed_bias)
  if embed_y:
    y = ops.linear(vs, y, embed_y_dim, s + "/embed_y", cfg.sn_cfg, use_sn=False,
                   use_bias=embed_bias)
  y_per_block = nb * [y]
  if hierarchical_z:
    chunks = torch.chunk(z, nb + 1, dim=1)
    z0, z_per_block = chunks[0], list(chunks[1:])
    if y is not None:
      y_per_block = [torch.cat([zi, y], 1) for zi in z_per_block]
  else:
    z0, z_per_block = z, nb * [z]
  net = ops.linear(vs, z0, in_ch[0] * 16, s + "/fc_noise", cfg.sn_cfg, use_sn=cfg.spectral_norm)
  net = net.reshape(-1, 4, 4, in_ch[0])
  for b in range(nb):
    name = "B%d" % (b + 1)
    net = biggan_block(vs, cfg, net, s + "/" + name, in_ch[b], out_ch[b], "up", True,
                       z_per_block[b], y_per_block[b], is_training)
    if name in attn_blocks:
      net = ops.non_local_block(vs, net, s + "/non_local_block", cfg.spectral_norm, cfg.sn_cfg)
  net = ops.batch_norm(vs, net, is_training, s + "/final_norm", cfg.bn_cfg, relu=True)  # :295
  net = ops.conv2d(vs, net, image_shape[2], 3, 3, 1, 1, s + "/final_conv", cfg.sn_cfg,
                   use_sn=cfg.spectral_norm, out_f32=True)
  return (torch.tanh(net) + 1.0) / 2.0


def biggan_discriminator(vs, cfg, x, y, is_training):
  kw = cfg.arch_kwargs
  ch = kw.get("ch", 96)
  attn_blocks = set(kw.get("blocks_with_attention", "B1").split(","))
  project_y = kw.get("project_y", True)
  s = "discriminator"
  colors, res = x.shape[-1], x.shape[1]
  if colors not in (1, 3):
    raise ValueError("Unsupported color channels: {}".format(colors))
  if res not in _D_MULT:
    raise ValueError("Unsupported resolution: {}".format(res))
  out_ch = [ch * c for c in _D_MULT[res]]
  in_ch = [colors] + out_ch[:-1]
  nb = len(in_ch)
  net = x
  for b in range(nb):
    name = "B%d" % (b + 1)
    last = b == nb - 1
    net = biggan_block(vs, cfg, net, s + "/" + name, in_ch[b], out_ch[b],
                       "none" if last else "down", False, None, y, is_training,
                       add_shortcut=in_ch[b] != out_ch[b])
    if name in attn_blocks:
      net = ops.non_local_block(vs, net, s + "/non_local_block", cfg.spectral_norm, cfg.sn_cfg)
  net = torch.relu(net)
  h = vs.q(net.sum(dim=(1, 2)))
  logit = ops.linear(vs, h, 1, s + "/final_fc", cfg.sn_cfg, use_sn=cfg.spectral_norm,
                     out_f32=True)
  if project_y:
    if y is None:
      raise ValueError("You must provide class information y to project.")
    kernel = vs.get(s + "/embedding_fc/kernel", (y.shape[1], out_ch[-1]), vs.glorot_normal_init())
    if cfg.spectral_norm:
      kernel = ops.spectral_norm(vs, kernel, s + "/embedding_fc/kernel", cfg.sn_cfg.epsilon,
                                 cfg.sn_cfg.singular_value)
    logit = logit + (vs.q(vs.q(y) @ vs.qw(kernel)) * h).sum(dim=1, keepdim=True)
  return torch.sigmoid(logit), logit, h


# ------------------------------------------------------------------------------------------------
# BigGAN-deep (resnet_biggan_deep.py:62-433); pinned by the parameter totals of
# resnet_biggan_deep_test.py:56-60 through the product's variable shapes, numerics "parity unpinned"
# ------------------------------------------------------------------------------------------------
_G_MULT_DEEP = {512: 4 * [16] + 4 * [8] + [4, 4, 2, 2, 1, 1, 1], 256: 4 * [16] + 4 * [8] + [4, 4, 2, 2, 1],
                128: 4 * [16] + 2 * [8] + [4, 4, 2, 2, 1], 64: 4 * [16] + 2 * [8] + [4, 4, 2],
                32: 8 * [4]}
_D_MULT_DEEP = {512: [1, 1, 1, 2, 2, 4, 4] + 4 * [8] + 4 * [16], 256: [1, 2, 2, 4, 4] + 4 * [8] + 4 * [16],
                128: [1, 2, 2, 4, 4] + 2 * [8] + 4 * [16], 64: [2, 4, 4] + 2 * [8] + 4 * [16],
                32: 8 * [2]}


def biggan_deep_block(vs, cfg, x, scope, in_ch, out_ch, scale, z, y, is_training):
  """resnet_biggan_deep.py:131-196: bn-relu-1x1, bn-relu-[unpool]-3x3, bn-relu-3x3,
  bn-relu-[avgpool]-1x1, plus the parameter-free shortcut of :90-117."""
  if x.shape[-1] != in_ch:
    raise ValueError("Unexpected number of input channels (expected {}, got {}).".format(
        in_ch, x.shape[-1]))
  bott = max(in_ch, out_ch) // 4
  sn, sc = cfg.spectral_norm, cfg.sn_cfg
  bn = lambda t, name: _batch_norm(vs, cfg, t, scope + "/" + name + "/bn", z=z, y=y,
                                   is_training=is_training, relu=True)
  out = ops.conv2d(vs, bn(x, "conv1"), bott, 1, 1, 1, 1, scope + "/conv1/1x1_conv", sc, use_sn=sn)
  out = bn(out, "conv2")
  if scale == "up":
    out = ops.unpool(out)
  out = ops.conv2d(vs, out, bott, 3, 3, 1, 1, scope + "/conv2/3x3_conv", sc, use_sn=sn)
  out = ops.conv2d(vs, bn(out, "conv3"), bott, 3, 3, 1, 1, scope + "/conv3/3x3_conv", sc, use_sn=sn)
  out = bn(out, "conv4")
  if scale == "down":
    out = vs.q(ops.avg_pool2(out))
  # shortcut (:90-117)
  shortcut = x
  if in_ch > out_ch:
    assert scale == "up"
    shortcut = shortcut[:, :, :, :out_ch]
  if scale == "up":
    shortcut = ops.unpool(shortcut)
  if scale == "down":
    shortcut = vs.q(ops.avg_pool2(shortcut))
  if in_ch < out_ch:
    assert scale == "down"
    added = ops.conv2d(vs, shortcut, out_ch - in_ch, 1, 1, 1, 1, scope + "/shortcut/add_channels",
                       sc, use_sn=sn)
    shortcut = torch.cat([shortcut, added], dim=-1)
  if scale == "up":
    # the HIP path stores conv4's output, then the upsampling kernel adds the shortcut to it
    out = ops.conv2d(vs, out, out_ch, 1, 1, 1, 1, scope + "/conv4/1x1_conv", sc, use_sn=sn)
    return vs.q(out + shortcut)
  return ops.conv2d(vs, out, out_ch, 1, 1, 1, 1, scope + "/conv4/1x1_conv", sc, use_sn=sn,
                    residual=shortcut)


def biggan_deep_generator(vs, cfg, z, y, is_training, image_shape=(128, 128, 3)):
  """resnet_biggan_deep.py:201-311."""
  kw = cfg.arch_kwargs
  ch = kw.get("ch", 128)
  embed_y, embed_y_dim = kw.get("embed_y", True), kw.get("embed_y_dim", 128)
  s = "generator"
  res = image_shape[0]
  if res not in _G_MULT_DEEP:
    raise ValueError("Unsupported resolution: {}".format(res))
  mult = _G_MULT_DEEP[res]
  in_ch = [ch * c for c in mult[:-1]]
  out_ch = [ch * c for c in mult[1:]]
  if embed_y:
    y = ops.linear(vs, y, embed_y_dim, s + "/embed_y", cfg.sn_cfg, use_sn=False, use_bias=False)
  if y is not None:
    y = torch.cat([z, y], 1)
    z = y
  net = ops.linear(vs, z, in_ch[0] * 16, s + "/fc_noise", cfg.sn_cfg, use_sn=cfg.spectral_norm)
  net = net.reshape(-1, 4, 4, in_ch[0])
  for b in range(len(in_ch)):
    scale = "none" if b % 2 == 0 else "up"
    net = biggan_deep_block(vs, cfg, net, s + "/B%d" % (b + 1), in_ch[b], out_ch[b], scale, z, y,
                            is_training)
    if scale == "up" and net.shape[1] == 64:
      net = ops.non_local_block(vs, net, s + "/non_local_block", cfg.spectral_norm, cfg.sn_cfg)
  net = ops.batch_norm(vs, net, is_training, s + "/final_norm", cfg.bn_cfg, relu=True)
  net = ops.conv2d(vs, net, image_shape[2], 3, 3, 1, 1, s + "/final_conv", cfg.sn_cfg,
                   use_sn=cfg.spectral_norm, out_f32=True)
  return (torch.tanh(net) + 1.0) / 2.0


def biggan_deep_discriminator(vs, cfg, x, y, is_training):
  """resnet_biggan_deep.py:314-433."""
  kw = cfg.arch_kwargs
  ch = kw.get("ch", 128)
  project_y = kw.get("project_y", True)
  s = "discriminator"
  colors, res = x.shape[-1], x.shape[1]
  if colors not in (1, 3):
    raise ValueError("Unsupported color channels: {}".format(colors))
  if res not in _D_MULT_DEEP:
    raise ValueError("Unsupported resolution: {}".format(res))
  mult = _D_MULT_DEEP[res]
  in_ch = [ch * c for c in mult[:-1]]
  out_ch = [ch * c for c in mult[1:]]
  net = ops.conv2d(vs, x, in_ch[0], 3, 3, 1, 1, s + "/initial_conv", cfg.sn_cfg,
                   use_sn=cfg.spectral_norm)
  for b in range(len(in_ch)):
    scale = "down" if b % 2 == 0 else "none"
    net = biggan_deep_block(vs, cfg, net, s + "/B%d" % (b + 1), in_ch[b], out_ch[b], scale, None,
                            y, is_training)
    if scale == "none" and net.shape[1] == 64:
      net = ops.non_local_block(vs, net, s + "/non_local_block", cfg.spectral_norm, cfg.sn_cfg)
  net = torch.relu(net)
  h = vs.q(net.sum(dim=(1, 2)))
  logit = ops.linear(vs, h, 1, s + "/final_fc", cfg.sn_cfg, use_sn=cfg.spectral_norm,
                     out_f32=True)
  if project_y:
    if y is None:
      raise ValueError("You must provide class information y to project.")
    kernel = vs.get(s + "/embedding_fc/kernel", (y.shape[1], out_ch[-1]), vs.glorot_normal_init())
    if cfg.spectral_norm:
      kernel = ops.spectral_norm(vs, kernel, s + "/embedding_fc/kernel", cfg.sn_cfg.epsilon,
                                 cfg.sn_cfg.singular_value)
    logit = logit + (vs.q(vs.q(y) @ vs.qw(kernel)) * h).sum(dim=1, keepdim=True)
  return torch.sigmoid(logit), logit, h


# ------------------------------------------------------------------------------------------------
# dcgan (dcgan.py:39-129) and sndcgan (sndcgan.py:36-127)
# ------------------------------------------------------------------------------------------------
def _half(size):
  return int(math.ceil(float(size) / 2.0))


def dcgan_generator(vs, cfg, z, y, is_training, image_shape=(64, 64, 3)):
  s = "generator"
  gf = 64
  bs = z.shape[0]
  h, w, colors = image_shape
  h2, w2 = _half(h), _half(w)
  h4, w4 = _half(h2), _half(w2)
  h8, w8 = _half(h4), _half(w4)
  h16, w16 = _half(h8), _half(w8)
  net = ops.linear(vs, z, gf * 8 * h16 * w16, s + "/g_fc1", cfg.sn_cfg)
  net = net.reshape(-1, h16, w16, gf * 8)
  bn = lambda t, nm: _batch_norm(vs, cfg, t, s + nm, z=z, y=y, is_training=is_training, relu=True)
  net = bn(net, "/g_bn1")
  net = ops.deconv2d(vs, net, [bs, h8, w8, gf * 4], 5, 5, 2, 2, s + "/g_dc1", cfg.sn_cfg)
  net = bn(net, "/g_bn2")
  net = ops.deconv2d(vs, net, [bs, h4, w4, gf * 2], 5, 5, 2, 2, s + "/g_dc2", cfg.sn_cfg)
  net = bn(net, "/g_bn3")
  net = ops.deconv2d(vs, net, [bs, h2, w2, gf], 5, 5, 2, 2, s + "/g_dc3", cfg.sn_cfg)
  net = bn(net, "/g_bn4")
  net = ops.deconv2d(vs, net, [bs, h, w, colors], 5, 5, 2, 2, s + "/g_dc4", cfg.sn_cfg,
                     out_f32=True)
  return 0.5 * torch.tanh(net) + 0.5


def dcgan_discriminator(vs, cfg, x, y, is_training):
  s = "discriminator"
  df = 64
  bs = x.shape[0]
  sn = cfg.spectral_norm
  net = ops.lrelu(ops.conv2d(vs, x, df, 5, 5, 2, 2, s + "/d_conv1", cfg.sn_cfg, use_sn=sn))
  net = ops.conv2d(vs, net, df * 2, 5, 5, 2, 2, s + "/d_conv2", cfg.sn_cfg, use_sn=sn)
  net = ops.lrelu(_batch_norm(vs, cfg, net, s + "/d_bn1", y=y, is_training=is_training))
  net = ops.conv2d(vs, net, df * 4, 5, 5, 2, 2, s + "/d_conv3", cfg.sn_cfg, use_sn=sn)
  net = ops.lrelu(_batch_norm(vs, cfg, net, s + "/d_bn2", y=y, is_training=is_training))
  net = ops.conv2d(vs, net, df * 8, 5, 5, 2, 2, s + "/d_conv4", cfg.sn_cfg, use_sn=sn)
  net = ops.lrelu(_batch_norm(vs, cfg, net, s + "/d_bn3", y=y, is_training=is_training))
  logit = ops.linear(vs, net.reshape(bs, -1), 1, s + "/d_fc4", cfg.sn_cfg, use_sn=sn,
                     out_f32=True)
  return torch.sigmoid(logit), logit, net


def sndcgan_generator(vs, cfg, z, y, is_training, image_shape=(128, 128, 3)):
  s = "generator"
  bs = z.shape[0]
  h, w, colors = image_shape
  h2, w2 = _half(h), _half(w)
  h4, w4 = _half(h2), _half(w2)
  h8, w8 = _half(h4), _half(w4)
  net = ops.linear(vs, z, h8 * w8 * 512, s + "/g_fc1", cfg.sn_cfg)
  bn = lambda t, nm: _batch_norm(vs, cfg, t, s + nm, z=z, y=y, is_training=is_training, relu=True)
  net = bn(net, "/g_bn1")
  net = net.reshape(bs, h8, w8, 512)
  net = ops.deconv2d(vs, net, [bs, h4, w4, 256], 4, 4, 2, 2, s + "/g_dc2", cfg.sn_cfg)
  net = bn(net, "/g_bn2")
  net = ops.deconv2d(vs, net, [bs, h2, w2, 128], 4, 4, 2, 2, s + "/g_dc3", cfg.sn_cfg)
  net = bn(net, "/g_bn3")
  net = ops.deconv2d(vs, net, [bs, h, w, 64], 4, 4, 2, 2, s + "/g_dc4", cfg.sn_cfg)
  net = bn(net, "/g_bn4")
  net = ops.deconv2d(vs, net, [bs, h, w, colors], 3, 3, 1, 1, s + "/g_dc5", cfg.sn_cfg,
                     out_f32=True)
  return (torch.tanh(net) + 1.0) / 2.0


def sndcgan_discriminator(vs, cfg, x, y, is_training):
  s = "discriminator"
  sn = cfg.spectral_norm
  x = vs.q(x * 2.0 - 1.0)
  spec = [(64, 3, 1), (128, 4, 2), (128, 3, 1), (256, 4, 2), (256, 3, 1), (512, 4, 2), (512, 3, 1)]
  net = x
  for i, (co, k, st) in enumerate(spec):
    net = ops.conv2d(vs, net, co, k, k, st, st, "%s/d_conv%d" % (s, i + 1), cfg.sn_cfg, use_sn=sn)
    net = ops.lrelu(net, 0.1)
  bs = x.shape[0]
  net = net.reshape(bs, -1)
  logit = ops.linear(vs, net, 1, s + "/d_fc1", cfg.sn_cfg, use_sn=sn, out_f32=True)
  return torch.sigmoid(logit), logit, net


# ------------------------------------------------------------------------------------------------
# infogan (infogan.py:35-100), resnet_stl (resnet_stl.py:33-108), resnet30 (resnet30.py:36-143)
# ------------------------------------------------------------------------------------------------
def infogan_generator(vs, cfg, z, y, is_training, image_shape=(32, 32, 3)):
  s = "generator"
  h, w, c = image_shape
  bs = z.shape[0]
  # the reference calls arch_ops.batch_norm directly (infogan.py:53-59): always plain batch norm
  bn = lambda t, nm: ops.batch_norm(vs, t, is_training, s + nm, cfg.bn_cfg)
  net = ops.linear(vs, z, 1024, s + "/g_fc1", cfg.sn_cfg)
  net = ops.lrelu(bn(net, "/g_bn1"))
  net = ops.linear(vs, net, 128 * (h // 4) * (w // 4), s + "/g_fc2", cfg.sn_cfg)
  net = ops.lrelu(bn(net, "/g_bn2"))
  net = net.reshape(bs, h // 4, w // 4, 128)
  net = ops.deconv2d(vs, net, [bs, h // 2, w // 2, 64], 4, 4, 2, 2, s + "/g_dc3", cfg.sn_cfg)
  net = ops.lrelu(bn(net, "/g_bn3"))
  net = ops.deconv2d(vs, net, [bs, h, w, c], 4, 4, 2, 2, s + "/g_dc4", cfg.sn_cfg, out_f32=True)
  return torch.sigmoid(net)


def infogan_discriminator(vs, cfg, x, y, is_training):
  s = "discriminator"
  sn = cfg.spectral_norm
  bs = x.shape[0]
  net = ops.lrelu(ops.conv2d(vs, x, 64, 4, 4, 2, 2, s + "/d_conv1", cfg.sn_cfg, use_sn=sn))
  net = ops.conv2d(vs, net, 128, 4, 4, 2, 2, s + "/d_conv2", cfg.sn_cfg, use_sn=sn)
  net = ops.lrelu(_batch_norm(vs, cfg, net, s + "/d_bn2", y=y, is_training=is_training))
  net = net.reshape(bs, -1)
  net = ops.linear(vs, net, 1024, s + "/d_fc3", cfg.sn_cfg, use_sn=sn)
  net = ops.lrelu(_batch_norm(vs, cfg, net, s + "/d_bn3", y=y, is_training=is_training))
  logit = ops.linear(vs, net, 1, s + "/d_fc4", cfg.sn_cfg, use_sn=sn, out_f32=True)
  return torch.sigmoid(logit), logit, net


def resnet_stl_generator(vs, cfg, z, y, is_training, image_shape=(48, 48, 3)):
  s = "generator"
  ch = 64
  magic = [(8, 4), (4, 2), (2, 1)]
  out = ops.linear(vs, z, 6 * 6 * 512, s + "/fc_noise", cfg.sn_cfg)
  out = out.reshape(-1, 6, 6, 512)
  for b in range(3):
    out = resnet_block(vs, cfg, out, "%s/B%d" % (s, b + 1), ch * magic[b][0], ch * magic[b][1],
                       "up", True, z, y, is_training)
  # scope="final_norm" is dropped by call_with_accepted_args (resnet_stl.py:60-61): the variables
  # live under the batch norm function's default name
  out = _batch_norm(vs, cfg, out, s + "/batch_norm", z=z, y=y, is_training=is_training, relu=True)
  out = ops.conv2d(vs, out, image_shape[2], 3, 3, 1, 1, s + "/final_conv", cfg.sn_cfg,
                   out_f32=True)
  return torch.sigmoid(out)


def resnet_stl_discriminator(vs, cfg, x, y, is_training):
  s = "discriminator"
  colors = x.shape[3]
  if colors not in (1, 3):
    raise ValueError("Number of color channels unknown: %s" % colors)
  ch = 64
  out = resnet_block(vs, cfg, x, s + "/B0", colors, ch, "down", False, None, y, is_training)
  magic = [(1, 2), (2, 4), (4, 8), (8, 16)]
  for b in range(4):
    out = resnet_block(vs, cfg, out, "%s/B%d" % (s, b + 1), ch * magic[b][0], ch * magic[b][1],
                       "down" if b < 3 else "none", False, None, y, is_training)
  out = torch.relu(out)
  pre = vs.q(out.mean(dim=(1, 2)))
  logit = ops.linear(vs, pre, 1, s + "/disc_final_fc", cfg.sn_cfg, use_sn=cfg.spectral_norm,
                     out_f32=True)
  return torch.sigmoid(logit), logit, pre


def resnet30_generator(vs, cfg, z, y, is_training, image_shape=(128, 128, 3)):
  s = "generator"
  ch = 64
  out = ops.linear(vs, z, 4 * 4 * 8 * ch, s + "/fc_noise", cfg.sn_cfg)
  out = out.reshape(-1, 4, 4, 8 * ch)
  cin, cout = 8 * ch, 4 * ch
  for sb in range(6):
    for i in range(5):
      out = resnet_block(vs, cfg, out, "%s/B_%d_%d" % (s, sb, i), cin, cin, "none", True, z, y,
                         is_training)
    if sb < 5:
      out = resnet_block(vs, cfg, out, "%s/B_%d_up" % (s, sb), cin, cout, "up", True, z, y,
                         is_training)
    cin, cout = cin // 2, cout // 2
  out = ops.conv2d(vs, out, image_shape[2], 3, 3, 1, 1, s + "/final_conv", cfg.sn_cfg,
                   out_f32=True)
  return torch.sigmoid(out)


def resnet30_discriminator(vs, cfg, x, y, is_training):
  s = "discriminator"
  ch = 64
  out = ops.conv2d(vs, x, ch // 4, 3, 3, 1, 1, s + "/color_conv", cfg.sn_cfg)
  cin, cout = ch // 4, ch // 2
  for sb in range(6):
    for i in range(5):
      out = resnet_block(vs, cfg, out, "%s/B_%d_%d" % (s, sb, i), cin, cin, "none", False, None,
                         y, is_training)
    if sb < 5:
      out = resnet_block(vs, cfg, out, "%s/B_%d_up" % (s, sb), cin, cout, "down", False, None, y,
                         is_training)
    cin, cout = cin * 2, cout * 2
  out = out.reshape(-1, 4 * 4 * 8 * ch)
  logit = ops.linear(vs, out, 1, s + "/disc_final_fc", cfg.sn_cfg, use_sn=cfg.spectral_norm,
                     out_f32=True)
  return torch.sigmoid(logit), logit, out


GENERATORS = {
    "infogan_arch": infogan_generator, "resnet_stl_arch": resnet_stl_generator,
    "resnet30_arch": resnet30_generator,
    "resnet_cifar_arch": resnet_cifar_generator, "resnet5_arch": resnet5_generator,
    "resnet_biggan_arch": biggan_generator, "dcgan_arch": dcgan_generator,
    "resnet_biggan_deep_arch": biggan_deep_generator,
    "sndcgan_arch": sndcgan_generator,
}
DISCRIMINATORS = {
    "infogan_arch": infogan_discriminator, "resnet_stl_arch": resnet_stl_discriminator,
    "resnet30_arch": resnet30_discriminator,
    "resnet_cifar_arch": resnet_cifar_discriminator, "resnet5_arch": resnet5_discriminator,
    "resnet_biggan_arch": biggan_discriminator, "dcgan_arch": dcgan_discriminator,
    "resnet_biggan_deep_arch": biggan_deep_discriminator,
    "sndcgan_arch": sndcgan_discriminator,
}
