"""ResNet building blocks (reference: architectures/resnet_ops.py:35-219).

Same block semantics -- [BN -> ReLU -> conv] x 2 plus a 3x3 convolutional shortcut, generator
blocks upsample (zero insertion) in conv1 and the shortcut, discriminator blocks average-pool
after conv2 and the shortcut -- expressed with the fused HIP kernels: the ReLUs are input gates of
the following convolution, the zero-inserted tensor is never materialised, the shortcut is added
in conv2's epilogue, and (linearity) the two poolings of a "down" block collapse into one pooling
of the sum.
"""
import math

from compare_gan_amd.architectures import abstract_arch
from compare_gan_amd.architectures import arch_ops as ops


import os as _os
_FORK = _os.environ.get("CGAMD_FORK", "1") != "0"   # A/B switch (read once)


def unpool(value, name="unpool"):
  """Zero-insertion 2x upsampling (resnet_ops.py:35-56) as a stand-alone op: a 1x1 identity gather
  is never needed on the hot path (conv2d(upsample=True) fuses it); kept for API parity."""
  raise NotImplementedError(
      "unpool is fused into conv2d(..., upsample=True); call that instead (name=%s)" % name)


def validate_image_inputs(inputs, validate_power2=True):
  """resnet_ops.py:59-67."""
  if inputs.dim() != 4:
    raise ValueError("Input tensor must have rank 4.")
  if inputs.shape[1] != inputs.shape[2]:
    raise ValueError("Input tensor does not have equal width and height: ", inputs.shape[1:3])
  width = inputs.shape[1]
  if validate_power2 and math.log(width, 2) != int(math.log(width, 2)):
    raise ValueError("Input tensor `width` is not a power of 2: ", width)


class _Shape(object):
  """Duck-typed NHWC tensor description for ops.conv_pool_supported."""

  def __init__(self, x, channels):
    self.shape = (x.shape[0], x.shape[1], x.shape[2], channels)
    self.is_meta = False

  def dim(self):
    return 4


class ResNetBlock(object):
  """ResNet block with options for various normalizations (resnet_ops.py:70-182)."""

  def __init__(self, name, in_channels, out_channels, scale, is_gen_block, layer_norm=False,
               spectral_norm=False, batch_norm=None, batch_norm_relu=None):
    assert scale in ["up", "down", "none"]
    self._name = name
    self._in_channels = in_channels
    self._out_channels = out_channels
    self._scale = scale
    # Generators upscale in the first conv, discriminators downscale after the second conv.
    self._scale1 = scale if is_gen_block else "none"
    self._scale2 = "none" if is_gen_block else scale
    self._layer_norm = layer_norm
    self._spectral_norm = spectral_norm
    self.batch_norm = batch_norm
    self.batch_norm_relu = batch_norm_relu

  def __call__(self, inputs, z, y, is_training):
    return self.apply(inputs=inputs, z=z, y=y, is_training=is_training)

  def _norm_relu(self, inputs, z, y, is_training, bn_name, ln_name):
    """batch_norm -> [layer_norm] -> ReLU (resnet_ops.py:159-165,169-175).  Without layer norm the
    ReLU stays fused (batch-norm kernel or the consumer convolution's input gate); with it the
    batch norm runs bare, the layer norm follows and the ReLU is left pending for the convolution."""
    if not self._layer_norm:
      return self.batch_norm_relu(inputs, z=z, y=y, is_training=is_training, name=bn_name)
    output = self.batch_norm(inputs, z=z, y=y, is_training=is_training, name=bn_name)
    output = ops.layer_norm(output, is_training=is_training, scope=ln_name)
    return ops.Act(output, 0.0)

  def _get_conv(self, inputs, in_channels, out_channels, scale, suffix, kernel_size=(3, 3),
                strides=(1, 1), residual=None, pool=True):
    """One convolution of the block (resnet_ops.py:112-134).  `pool=False` lets the caller pool
    the sum of two "down" branches once."""
    if inputs.shape[-1] != in_channels:
      raise ValueError("Unexpected number of input channels.")
    if scale not in ["up", "down", "none"]:
      raise ValueError("Scale: got {}, expected 'up', 'down', or 'none'.".format(scale))
    name = "{}_{}".format("same" if scale == "none" else scale, suffix)
    if (scale == "down" and pool and strides == (1, 1) and
        ops.conv_pool_supported(inputs, out_channels, kernel_size[0], kernel_size[1])):
      # convolution + 2x2 average pooling in one kernel; `residual` is at the POOLED resolution
      return ops.conv2d(
          inputs, output_dim=out_channels, k_h=kernel_size[0], k_w=kernel_size[1], d_h=1, d_w=1,
          use_sn=self._spectral_norm, name=name, residual=residual, pool=True)
    if scale == "down" and pool and residual is not None:
      raise ValueError("a pooled residual needs the fused convolution (check conv_pool_supported)")
    outputs = ops.conv2d(
        inputs, output_dim=out_channels, k_h=kernel_size[0], k_w=kernel_size[1],
        d_h=strides[0], d_w=strides[1], use_sn=self._spectral_norm, name=name,
        upsample=(scale == "up"), residual=residual)
    if scale == "down" and pool:
      outputs = ops.avg_pool2(outputs)
    return outputs

  def apply(self, inputs, z, y, is_training):
    if inputs.shape[-1] != self._in_channels:
      raise ValueError("Unexpected number of input channels.")
    with ops.variable_scope(self._name):
      # variables are created in the reference order: shortcut, bn1, conv1, bn2, conv2
      down = self._scale == "down"
      # discriminator "down" block with both poolings fused into their convolutions' epilogues:
      # pool(conv2) + pool(shortcut), no full-resolution tensor written (resnet_ops.py:131-133)
      fuse = (down and not inputs.is_meta and
              ops.conv_pool_supported(inputs, self._out_channels, 3, 3) and
              ops.conv_pool_supported(_Shape(inputs, self._out_channels), self._out_channels,
                                      3, 3))
      if _FORK:
        inputs, inputs_main = ops.fork(inputs)
      else:
        inputs_main = inputs
      shortcut = self._get_conv(inputs, self._in_channels, self._out_channels, self._scale,
                                suffix="conv_shortcut", pool=fuse)
      output = self._norm_relu(inputs_main, z, y, is_training, "bn1", "ln1")
      output = self._get_conv(output, self._in_channels, self._out_channels, self._scale1,
                              suffix="conv1")
      output = self._norm_relu(output, z, y, is_training, "bn2", "ln2")
      # conv2 + shortcut in one epilogue; pool(conv2) + pool(shortcut) == pool(conv2 + shortcut)
      output = self._get_conv(output, self._out_channels, self._out_channels, self._scale2,
                              suffix="conv2", residual=shortcut, pool=fuse)
      if down and not fuse:
        output = ops.avg_pool2(output)
      return output


class ResNetGenerator(abstract_arch.AbstractGenerator):
  """Abstract base class for generators based on the ResNet architecture."""

  def _resnet_block(self, name, in_channels, out_channels, scale):
    if scale not in ["up", "none"]:
      raise ValueError("Unknown generator ResNet block scaling: {}.".format(scale))
    return ResNetBlock(name=name, in_channels=in_channels, out_channels=out_channels,
                       scale=scale, is_gen_block=True, spectral_norm=self._spectral_norm,
                       batch_norm=self.batch_norm, batch_norm_relu=self.batch_norm_relu)


class ResNetDiscriminator(abstract_arch.AbstractDiscriminator):
  """Abstract base class for discriminators based on the ResNet architecture."""

  def _resnet_block(self, name, in_channels, out_channels, scale):
    if scale not in ["down", "none"]:
      raise ValueError("Unknown discriminator ResNet block scaling: {}.".format(scale))
    return ResNetBlock(name=name, in_channels=in_channels, out_channels=out_channels,
                       scale=scale, is_gen_block=False, layer_norm=self._layer_norm,
                       spectral_norm=self._spectral_norm, batch_norm=self.batch_norm,
                       batch_norm_relu=self.batch_norm_relu)
