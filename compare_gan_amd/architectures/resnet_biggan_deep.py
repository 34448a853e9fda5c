"""BigGAN-deep ResNet, 32..512 px (reference: architectures/resnet_biggan_deep.py:62-433).

Against resnet_biggan: four-convolution bottleneck blocks (1x1 -> 3x3 -> 3x3 -> 1x1, bottleneck
width max(in, out) / 4), parameter-free shortcuts (G drops surplus channels and zero-insertion
upsamples, D average-pools and appends `out - in` channels computed by one 1x1 convolution), two
blocks per resolution ("none" then "up" in G, "down" then "none" in D), no hierarchical z: every
conditional batch norm sees concat(z, embed(y)), self-attention at 64x64.
Parameter counts at 128 px (resnet_biggan_deep_test.py:56-60): G 50,244,484 / D 34,590,210.

Every arithmetic step is one of the engine's kernels: the block's last 1x1 convolution takes the
shortcut as its fused residual; G's "up" shortcut is cg_unpool2 with the main branch as residual.
"""
import torch

from compare_gan_amd import gin
from compare_gan_amd.architectures import abstract_arch
from compare_gan_amd.architectures import arch_ops as ops
from compare_gan_amd.architectures import resnet_ops
from compare_gan_amd.hip import functional as Fn


@gin.configurable
class BigGanDeepResNetBlock(object):
  """ResNet block with bottleneck and identity preserving skip connections."""

  def __init__(self, name, in_channels, out_channels, scale, spectral_norm=False,
               batch_norm=None, batch_norm_relu=None):
    if scale not in ["up", "down", "none"]:
      raise ValueError("Scale: got {}, expected 'up', 'down', or 'none'.".format(scale))
    self._name = name
    self._in_channels = in_channels
    self._out_channels = out_channels
    self._scale = scale
    self._spectral_norm = spectral_norm
    self.batch_norm = batch_norm
    self.batch_norm_relu = batch_norm_relu

  def __call__(self, inputs, z, y, is_training):
    return self.apply(inputs=inputs, z=z, y=y, is_training=is_training)

  def _shortcut_down(self, inputs):
    """D: average pool, then append the missing channels (resnet_biggan_deep.py:104-116)."""
    with ops.variable_scope("shortcut"):
      shortcut = inputs
      num_channels = inputs.shape[-1]
      if self._scale == "down":
        shortcut = ops.avg_pool2(shortcut)
      if num_channels < self._out_channels:
        if self._scale != "down":
          raise ValueError("channels can only be added in a 'down' block")
        added = ops.conv1x1(shortcut, self._out_channels - num_channels, name="add_channels",
                            use_sn=self._spectral_norm)
        shortcut = torch.cat([shortcut, added], dim=-1)
      return shortcut

  def apply(self, inputs, z, y, is_training):
    if inputs.shape[-1] != self._in_channels:
      raise ValueError("Unexpected number of input channels (expected {}, got {}).".format(
          self._in_channels, inputs.shape[-1]))
    num_channels = inputs.shape[-1]
    if num_channels > self._out_channels and self._scale != "up":
      raise ValueError("channels can only be dropped in an 'up' block")
    bottleneck = max(self._in_channels, self._out_channels) // 4
    sn = self._spectral_norm
    bn_relu = lambda t: self.batch_norm_relu(t, z=z, y=y, is_training=is_training, name="bn")
    with ops.variable_scope(self._name):
      with ops.variable_scope("conv1"):
        outputs = ops.conv1x1(bn_relu(inputs), bottleneck, name="1x1_conv", use_sn=sn)
      with ops.variable_scope("conv2"):
        outputs = ops.conv2d(bn_relu(outputs), bottleneck, k_h=3, k_w=3, d_h=1, d_w=1,
                             name="3x3_conv", use_sn=sn, upsample=(self._scale == "up"))
      with ops.variable_scope("conv3"):
        outputs = ops.conv2d(bn_relu(outputs), bottleneck, k_h=3, k_w=3, d_h=1, d_w=1,
                             name="3x3_conv", use_sn=sn)
      with ops.variable_scope("conv4"):
        outputs = bn_relu(outputs)
        if self._scale == "down":
          outputs = ops.avg_pool2(outputs)
        if self._scale == "up":
          # the shortcut needs its own kernel (zero-insertion of the leading out_channels
          # channels): it takes the main branch as residual
          outputs = ops.conv1x1(outputs, self._out_channels, name="1x1_conv", use_sn=sn)
          shortcut = inputs
          if num_channels > self._out_channels:
            shortcut = inputs[:, :, :, :self._out_channels].contiguous()
          with ops.variable_scope("shortcut"):
            return ops.unpool(shortcut, residual=outputs)
        # "none" / "down": the shortcut is the fused residual of the last convolution (its
        # variables, shortcut/add_channels, are therefore created just before conv4/1x1_conv's;
        # names are the reference's, only the creation order of these two differs)
        pre = outputs
      shortcut = self._shortcut_down(inputs)
      with ops.variable_scope("conv4"):
        return ops.conv1x1(pre, self._out_channels, name="1x1_conv", use_sn=sn, residual=shortcut)


_G_CHANNEL_MULTIPLIERS = {512: 4 * [16] + 4 * [8] + [4, 4, 2, 2, 1, 1, 1],
                          256: 4 * [16] + 4 * [8] + [4, 4, 2, 2, 1],
                          128: 4 * [16] + 2 * [8] + [4, 4, 2, 2, 1],
                          64: 4 * [16] + 2 * [8] + [4, 4, 2],
                          32: 8 * [4]}
_D_CHANNEL_MULTIPLIERS = {512: [1, 1, 1, 2, 2, 4, 4] + 4 * [8] + 4 * [16],
                          256: [1, 2, 2, 4, 4] + 4 * [8] + 4 * [16],
                          128: [1, 2, 2, 4, 4] + 2 * [8] + 4 * [16],
                          64: [2, 4, 4] + 2 * [8] + 4 * [16],
                          32: 8 * [2]}


@gin.configurable
class Generator(abstract_arch.AbstractGenerator):
  """ResNet-based generator supporting resolutions 32, 64, 128, 256, 512."""

  def __init__(self, ch=128, embed_y=True, embed_y_dim=128, experimental_fast_conv_to_rgb=False,
               **kwargs):
    super(Generator, self).__init__(**kwargs)
    self._ch = ch
    self._embed_y = embed_y
    self._embed_y_dim = embed_y_dim
    # the reference's TPU trick (128 output channels, sliced to `colors`) changes the variable
    # shapes; the narrow-output convolution kernels make it pointless here
    if experimental_fast_conv_to_rgb:
      raise NotImplementedError("experimental_fast_conv_to_rgb is a TPU workaround; not offered")

  def _resnet_block(self, name, in_channels, out_channels, scale):
    if scale not in ["up", "none"]:
      raise ValueError("Unknown generator ResNet block scaling: {}.".format(scale))
    return BigGanDeepResNetBlock(name=name, in_channels=in_channels, out_channels=out_channels,
                                 scale=scale, spectral_norm=self._spectral_norm,
                                 batch_norm=self.batch_norm, batch_norm_relu=self.batch_norm_relu)

  def _get_in_out_channels(self):
    resolution = self._image_shape[0]
    if resolution not in _G_CHANNEL_MULTIPLIERS:
      raise ValueError("Unsupported resolution: {}".format(resolution))
    mult = _G_CHANNEL_MULTIPLIERS[resolution]
    return [self._ch * c for c in mult[:-1]], [self._ch * c for c in mult[1:]]

  def apply(self, z, y, is_training):
    seed_size = 4
    if self._embed_y:
      y = ops.linear(y, self._embed_y_dim, scope="embed_y", use_sn=False, use_bias=False)
    if y is not None:
      # tf.concat([z, y], axis=1), resnet_biggan_deep.py:262-264
      y = torch.cat([ops._to_bf16(z), ops._to_bf16(y)], 1)  # pylint: disable=protected-access
      z = y
    in_channels, out_channels = self._get_in_out_channels()
    net = ops.linear(z, in_channels[0] * seed_size * seed_size, scope="fc_noise",
                     use_sn=self._spectral_norm)
    net = net.reshape(-1, seed_size, seed_size, in_channels[0])
    for block_idx in range(len(in_channels)):
      scale = "none" if block_idx % 2 == 0 else "up"
      block = self._resnet_block(name="B{}".format(block_idx + 1),
                                 in_channels=in_channels[block_idx],
                                 out_channels=out_channels[block_idx], scale=scale)
      net = block(net, z=z, y=y, is_training=is_training)
      if scale == "up" and net.shape[1] == 64:       # self-attention at 64x64
        net = ops.non_local_block(net, "non_local_block", use_sn=self._spectral_norm)
    net = ops.batch_norm(net, is_training=is_training, name="final_norm", relu=True)
    net = ops.conv2d(net, output_dim=self._image_shape[2], k_h=3, k_w=3, d_h=1, d_w=1,
                     name="final_conv", use_sn=self._spectral_norm, out_f32=True)
    return ops.output_head(net, 1)  # (tanh + 1) / 2


@gin.configurable
class Discriminator(abstract_arch.AbstractDiscriminator):
  """ResNet-based discriminator supporting resolutions 32, 64, 128, 256, 512."""

  def __init__(self, ch=128, blocks_with_attention="B1", project_y=True, **kwargs):
    super(Discriminator, self).__init__(**kwargs)
    self._ch = ch
    self._blocks_with_attention = set(blocks_with_attention.split(","))   # kept, unused (as in
    self._project_y = project_y                                           # the reference)

  def _resnet_block(self, name, in_channels, out_channels, scale):
    if scale not in ["down", "none"]:
      raise ValueError("Unknown discriminator ResNet block scaling: {}.".format(scale))
    return BigGanDeepResNetBlock(name=name, in_channels=in_channels, out_channels=out_channels,
                                 scale=scale, spectral_norm=self._spectral_norm,
                                 batch_norm=self.batch_norm, batch_norm_relu=self.batch_norm_relu)

  def _get_in_out_channels(self, colors, resolution):
    if colors not in [1, 3]:
      raise ValueError("Unsupported color channels: {}".format(colors))
    if resolution not in _D_CHANNEL_MULTIPLIERS:
      raise ValueError("Unsupported resolution: {}".format(resolution))
    mult = _D_CHANNEL_MULTIPLIERS[resolution]
    return [self._ch * c for c in mult[:-1]], [self._ch * c for c in mult[1:]]

  def apply(self, x, y, is_training):
    resnet_ops.validate_image_inputs(x)
    in_channels, out_channels = self._get_in_out_channels(colors=x.shape[-1],
                                                          resolution=x.shape[1])
    net = ops.conv2d(x, output_dim=in_channels[0], k_h=3, k_w=3, d_h=1, d_w=1,
                     name="initial_conv", use_sn=self._spectral_norm)
    for block_idx in range(len(in_channels)):
      scale = "down" if block_idx % 2 == 0 else "none"
      block = self._resnet_block(name="B{}".format(block_idx + 1),
                                 in_channels=in_channels[block_idx],
                                 out_channels=out_channels[block_idx], scale=scale)
      net = block(net, z=None, y=y, is_training=is_training)
      if scale == "none" and net.shape[1] == 64:     # self-attention at 64x64
        net = ops.non_local_block(net, "non_local_block", use_sn=self._spectral_norm)
    # relu + reduce_sum over [1, 2] + linear(C -> 1); h also feeds the projection term below
    out_logit, h = ops.pooled_linear_head(ops.relu(net), mean=False, scope="final_fc",
                                          use_sn=self._spectral_norm)
    if self._project_y:
      if y is None:
        raise ValueError("You must provide class information y to project.")
      embedded_y = ops.linear(y, out_channels[-1], scope="embedding_fc", use_bias=False,
                              use_sn=self._spectral_norm,
                              kernel_initializer=ops.glorot_normal())
      if not h.is_meta:
        out_logit = Fn.add_f32(out_logit, Fn.RowDotFn.apply(embedded_y, h))
    return ops.output_head(out_logit, 0), out_logit, h
