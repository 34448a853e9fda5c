"""SN-DCGAN (reference: architectures/sndcgan.py:36-127; Miyato et al. 2018)."""
import math

from compare_gan_amd.architectures import abstract_arch
from compare_gan_amd.architectures import arch_ops as ops


def conv_out_size_same(size, stride):
  return int(math.ceil(float(size) / float(stride)))


class Generator(abstract_arch.AbstractGenerator):
  """fc -> BN(flat) -> 3 x deconv 4x4/2 -> deconv 3x3/1 -> (tanh + 1) / 2."""

  def apply(self, z, y, is_training):
    batch_size = z.shape[0]
    s_h, s_w, colors = self._image_shape
    s_h2, s_w2 = conv_out_size_same(s_h, 2), conv_out_size_same(s_w, 2)
    s_h4, s_w4 = conv_out_size_same(s_h2, 2), conv_out_size_same(s_w2, 2)
    s_h8, s_w8 = conv_out_size_same(s_h4, 2), conv_out_size_same(s_w4, 2)
    net = ops.linear(z, s_h8 * s_w8 * 512, scope="g_fc1")
    # the first batch norm normalises the FLAT fc output (sndcgan.py:59-62)
    net = ops.as_tensor(self.batch_norm_relu(net, z=z, y=y, is_training=is_training, name="g_bn1"))
    net = net.reshape(batch_size, s_h8, s_w8, 512)
    net = ops.deconv2d(net, [batch_size, s_h4, s_w4, 256], 4, 4, 2, 2, name="g_dc2")
    net = self.batch_norm_relu(net, z=z, y=y, is_training=is_training, name="g_bn2")
    net = ops.deconv2d(net, [batch_size, s_h2, s_w2, 128], 4, 4, 2, 2, name="g_dc3")
    net = self.batch_norm_relu(net, z=z, y=y, is_training=is_training, name="g_bn3")
    net = ops.deconv2d(net, [batch_size, s_h, s_w, 64], 4, 4, 2, 2, name="g_dc4")
    net = self.batch_norm_relu(net, z=z, y=y, is_training=is_training, name="g_bn4")
    net = ops.deconv2d(net, [batch_size, s_h, s_w, colors], 3, 3, 1, 1, name="g_dc5",
                       out_f32=True)
    return ops.output_head(net, 1)  # (tanh + 1) / 2


class Discriminator(abstract_arch.AbstractDiscriminator):
  """7 convolutions (3x3/1 and 4x4/2 alternating) with leaky ReLU 0.1, fc.

  The reference rescales its [0,1] input with `x * 2 - 1` (sndcgan.py:108); the staging kernel
  that converts images to bf16 applies it (input_affine), so `x` arrives already rescaled."""

  input_affine = (2.0, -1.0)

  def apply(self, x, y, is_training):
    del is_training, y
    use_sn = self._spectral_norm
    layers = [(64, 3, 1), (128, 4, 2), (128, 3, 1), (256, 4, 2), (256, 3, 1), (512, 4, 2),
              (512, 3, 1)]
    net = x
    for i, (channels, k, stride) in enumerate(layers):
      net = ops.conv2d(net, channels, k, k, stride, stride, name="d_conv%d" % (i + 1),
                       use_sn=use_sn)
      net = ops.lrelu(net, leak=0.1)
    batch_size = x.shape[0]
    flat = ops.Act(net.x.reshape(batch_size, -1), net.slope)
    out_logit = ops.linear(flat, 1, scope="d_fc1", use_sn=use_sn, out_f32=True)
    return ops.output_head(out_logit, 0), out_logit, net
