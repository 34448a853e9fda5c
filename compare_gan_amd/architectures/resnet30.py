"""ResNet30: 6 super-blocks of 5 same-resolution blocks (+ one re-sampling block between them),
128x128 (reference: architectures/resnet30.py:36-143)."""
from compare_gan_amd.architectures import arch_ops as ops
from compare_gan_amd.architectures import resnet_ops


class Generator(resnet_ops.ResNetGenerator):
  """ResNet30 generator, generates images of resolution 128x128 (resnet30.py:36-88).  No batch
  norm / ReLU in front of the final convolution (as in the reference)."""

  def apply(self, z, y, is_training):
    if z.dim() != 2:
      raise ValueError("Expected shape [batch_size, z_dim], got %s." % list(z.shape))
    ch = 64
    colors = self._image_shape[2]
    output = ops.linear(z, 4 * 4 * 8 * ch, scope="fc_noise")
    output = output.reshape(-1, 4, 4, 8 * ch)
    in_channels, out_channels = 8 * ch, 4 * ch
    for superblock in range(6):
      for i in range(5):
        block = self._resnet_block(name="B_{}_{}".format(superblock, i), in_channels=in_channels,
                                   out_channels=in_channels, scale="none")
        output = block(output, z=z, y=y, is_training=is_training)
      if superblock < 5:    # upscale 5 times
        block = self._resnet_block(name="B_{}_up".format(superblock), in_channels=in_channels,
                                   out_channels=out_channels, scale="up")
        output = block(output, z=z, y=y, is_training=is_training)
      in_channels //= 2
      out_channels //= 2
    output = ops.conv2d(output, output_dim=colors, k_h=3, k_w=3, d_h=1, d_w=1, name="final_conv",
                        out_f32=True)
    return ops.output_head(output, 0)  # sigmoid


class Discriminator(resnet_ops.ResNetDiscriminator):
  """ResNet30 discriminator, 128x128x3 and 128x128x1 inputs (resnet30.py:91-143): an unnormalised
  3x3 colour convolution to 16 channels, 6 super-blocks, flatten, fc."""

  def apply(self, x, y, is_training):
    resnet_ops.validate_image_inputs(x)
    colors = x.shape[-1]
    assert colors in [1, 3]
    ch = 64
    output = ops.conv2d(x, output_dim=ch // 4, k_h=3, k_w=3, d_h=1, d_w=1, name="color_conv")
    in_channels, out_channels = ch // 4, ch // 2
    for superblock in range(6):
      for i in range(5):
        block = self._resnet_block(name="B_{}_{}".format(superblock, i), in_channels=in_channels,
                                   out_channels=in_channels, scale="none")
        output = block(output, z=None, y=y, is_training=is_training)
      if superblock < 5:    # downscale 5 times (the reference keeps the "_up" suffix)
        block = self._resnet_block(name="B_{}_up".format(superblock), in_channels=in_channels,
                                   out_channels=out_channels, scale="down")
        output = block(output, z=None, y=y, is_training=is_training)
      in_channels *= 2
      out_channels *= 2
    output = ops.as_tensor(output).reshape(-1, 4 * 4 * 8 * ch)
    out_logit = ops.linear(output, 1, scope="disc_final_fc", use_sn=self._spectral_norm,
                           out_f32=True)
    return ops.output_head(out_logit, 0), out_logit, output
