"""ResNet for 48x48 images (STL-10), 3 generator / 5 discriminator blocks (reference:
architectures/resnet_stl.py:33-108)."""
from compare_gan_amd.architectures import arch_ops as ops
from compare_gan_amd.architectures import resnet_ops


class Generator(resnet_ops.ResNetGenerator):
  """ResNet generator, 3 blocks, supporting 48x48 resolution (resnet_stl.py:33-65)."""

  def apply(self, z, y, is_training):
    ch = 64
    colors = self._image_shape[2]
    magic = [(8, 4), (4, 2), (2, 1)]
    output = ops.linear(z, 6 * 6 * 512, scope="fc_noise")
    output = output.reshape(-1, 6, 6, 512)
    for block_idx in range(3):
      block = self._resnet_block(name="B{}".format(block_idx + 1),
                                 in_channels=ch * magic[block_idx][0],
                                 out_channels=ch * magic[block_idx][1], scale="up")
      output = block(output, z=z, y=y, is_training=is_training)
    # the reference passes scope="final_norm" (resnet_stl.py:60-61): no batch norm function
    # accepts a `scope` argument, call_with_accepted_args drops it and the variables land under the
    # function's default name -- kept, checkpoints depend on it
    output = self.batch_norm_relu(output, z=z, y=y, is_training=is_training, scope="final_norm")
    output = ops.conv2d(output, output_dim=colors, k_h=3, k_w=3, d_h=1, d_w=1, name="final_conv",
                        out_f32=True)
    return ops.output_head(output, 0)  # sigmoid


class Discriminator(resnet_ops.ResNetDiscriminator):
  """ResNet discriminator, 5 blocks (4 of them down-sampling), 48x48 (resnet_stl.py:68-108)."""

  def apply(self, x, y, is_training):
    resnet_ops.validate_image_inputs(x, validate_power2=False)
    colors = x.shape[-1]
    if colors not in [1, 3]:
      raise ValueError("Number of color channels unknown: %s" % colors)
    ch = 64
    block = self._resnet_block(name="B0", in_channels=colors, out_channels=ch, scale="down")
    output = block(x, z=None, y=y, is_training=is_training)
    magic = [(1, 2), (2, 4), (4, 8), (8, 16)]
    for block_idx in range(4):
      block = self._resnet_block(name="B{}".format(block_idx + 1),
                                 in_channels=ch * magic[block_idx][0],
                                 out_channels=ch * magic[block_idx][1],
                                 scale="down" if block_idx < 3 else "none")
      output = block(output, z=None, y=y, is_training=is_training)
    out_logit, pre_logits = ops.pooled_linear_head(ops.relu(output), mean=True, scope="disc_final_fc",
                                                   use_sn=self._spectral_norm)
    return ops.output_head(out_logit, 0), out_logit, pre_logits
