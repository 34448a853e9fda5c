"""SN-GAN CIFAR ResNet, 32x32 (reference: architectures/resnet_cifar.py:34-167; SN-GAN Table 4)."""
import torch

from compare_gan_amd import gin
from compare_gan_amd.architectures import arch_ops as ops
from compare_gan_amd.architectures import resnet_ops
from compare_gan_amd.hip import functional as Fn


def split_z_and_condition(z, y, num_blocks, hierarchical_z):
  """z0 for the seed plus per-block (z, y) lists (resnet_cifar.py:80-88, resnet_biggan.py:250-258)."""
  y_per_block = num_blocks * [y]
  if not hierarchical_z:
    return z, num_blocks * [z], y_per_block
  chunks = torch.chunk(z, num_blocks + 1, dim=1)
  if len(chunks) != num_blocks + 1 or any(c.shape[1] != chunks[0].shape[1] for c in chunks):
    raise ValueError("z_dim {} is not divisible into {} chunks.".format(z.shape[1], num_blocks + 1))
  z0, z_per_block = chunks[0].contiguous(), [c.contiguous() for c in chunks[1:]]
  if y is not None:
    y_per_block = [torch.cat([ops._to_bf16(zi), ops._to_bf16(y)], 1)  # pylint: disable=protected-access
                   for zi in z_per_block]
  return z0, z_per_block, y_per_block


@gin.configurable
class Generator(resnet_ops.ResNetGenerator):
  """ResNet generator, 3 up-blocks from a 4x4x256 seed, 32x32 output."""

  def __init__(self, hierarchical_z=False, embed_z=False, embed_y=False, **kwargs):
    super(Generator, self).__init__(**kwargs)
    self._hierarchical_z = hierarchical_z
    self._embed_z = embed_z
    self._embed_y = embed_y

  def apply(self, z, y, is_training):
    assert self._image_shape[0] == 32
    assert self._image_shape[1] == 32
    num_blocks = 3
    z_dim = z.shape[1]
    if self._embed_z:
      z = ops.linear(z, z_dim, scope="embed_z", use_sn=self._spectral_norm)
    if self._embed_y:
      y = ops.linear(y, z_dim, scope="embed_y", use_sn=self._spectral_norm)
    z0, z_per_block, y_per_block = split_z_and_condition(z, y, num_blocks, self._hierarchical_z)
    output = ops.linear(z0, 4 * 4 * 256, scope="fc_noise", use_sn=self._spectral_norm)
    output = output.reshape(-1, 4, 4, 256)
    for block_idx in range(num_blocks):
      block = self._resnet_block(name="B{}".format(block_idx + 1), in_channels=256,
                                 out_channels=256, scale="up")
      output = block(output, z=z_per_block[block_idx], y=y_per_block[block_idx],
                     is_training=is_training)
    output = self.batch_norm_relu(output, z=z, y=y, is_training=is_training, name="final_norm")
    output = ops.conv2d(output, output_dim=self._image_shape[2], k_h=3, k_w=3, d_h=1, d_w=1,
                        name="final_conv", use_sn=self._spectral_norm, out_f32=True)
    return ops.output_head(output, 0)  # sigmoid


@gin.configurable
class Discriminator(resnet_ops.ResNetDiscriminator):
  """ResNet discriminator, 4 blocks, 32x32 inputs with 1 or 3 colors."""

  def __init__(self, project_y=False, **kwargs):
    super(Discriminator, self).__init__(**kwargs)
    self._project_y = project_y

  def apply(self, x, y, is_training):
    resnet_ops.validate_image_inputs(x)
    colors = x.shape[3]
    if colors not in [1, 3]:
      raise ValueError("Number of color channels not supported: {}".format(colors))
    output = x
    for block_idx in range(4):
      block = self._resnet_block(name="B{}".format(block_idx + 1),
                                 in_channels=colors if block_idx == 0 else 128,
                                 out_channels=128, scale="down" if block_idx <= 1 else "none")
      output = block(output, z=None, y=y, is_training=is_training)
    # relu + reduce_mean over [1, 2] + linear(128 -> 1)
    out_logit, h = ops.pooled_linear_head(ops.relu(output), mean=True, scope="disc_final_fc",
                                          use_sn=self._spectral_norm)
    if self._project_y:
      if y is None:
        raise ValueError("You must provide class information y to project.")
      embedded_y = ops.linear(y, 128, use_bias=False, scope="embedding_fc",
                              use_sn=self._spectral_norm)
      if not h.is_meta:
        out_logit = Fn.add_f32(out_logit, Fn.RowDotFn.apply(embedded_y, h))
    return ops.output_head(out_logit, 0), out_logit, h
