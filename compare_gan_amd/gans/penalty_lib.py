"""Discriminator penalties (reference: gans/penalty_lib.py:28-108)."""
import torch

from compare_gan_amd import gin
from compare_gan_amd import utils
from compare_gan_amd.hip import functional as Fn
from compare_gan_amd.hip import kernels as K
from compare_gan_amd.tpu import tpu_random


@gin.configurable
def no_penalty():
  return None   # tf.constant(0.0): nothing to add to d_loss


def _gradient_norm_penalty(discriminator, x_in, y, is_training):
  """mean((sqrt(1e-4 + sum_{hwc} (d logits / d x)^2) - 1)^2)   (penalty_lib.py:50-55,74-81).

  x_in: staged bf16 images requiring grad.  The inner backward runs with create_graph=True so the
  penalty is differentiable w.r.t. D's weights; every op on that path is a HIP kernel."""
  logits = discriminator(x_in, y=y, is_training=is_training, reuse=True)[1]
  ones = torch.ones_like(logits)   # d(sum logits)/d logits: a constant fill, not arithmetic
  with Fn.only_input_grads():
    gradients, = torch.autograd.grad(logits, [x_in], grad_outputs=ones, create_graph=True)
  if gradients.dtype != torch.float32:
    gradients = Fn.ToF32Fn.apply(gradients)
  return Fn.GradientPenaltyFn.apply(gradients)


@gin.configurable(whitelist=[])
def dragan_penalty(discriminator, x, y, is_training):
  """DRAGAN gradient penalty (penalty_lib.py:33-56)."""
  raise NotImplementedError(
      "dragan_penalty needs the global std of x (tf.nn.moments over all axes); not used by any "
      "example config -- listed as a 'next' row in SURVEY.md section 8f.")


@gin.configurable(whitelist=[])
def wgangp_penalty(discriminator, x, x_fake, y, is_training):
  """WGAN gradient penalty (penalty_lib.py:59-82).  x, x_fake: fp32 images in [0,1]."""
  alpha = tpu_random.uniform([x.shape[0]], name="wgangp_penalty/alpha", device=x.device)
  interpolates = K.interpolate(x.contiguous(), x_fake.detach().contiguous(), alpha)
  a, b = getattr(discriminator, "input_affine", (1.0, 0.0))
  if (a, b) != (1.0, 0.0):
    interpolates = K.axpby(interpolates, a, None, 0.0) if b == 0.0 else _affine(interpolates, a, b)
  interpolates.requires_grad_(True)
  pen = _gradient_norm_penalty(discriminator, interpolates, y, is_training)
  return pen


def _affine(x, a, b):
  return K.cast_f32_to_bf16(K.cast_bf16_to_f32(x), a, b)


@gin.configurable(whitelist=[])
def l2_penalty(discriminator):
  """L2 penalty over D's kernels (penalty_lib.py:85-102)."""
  raise NotImplementedError("l2_penalty: not used by any example config (SURVEY.md section 8f).")


@gin.configurable("penalty", whitelist=["fn"])
def get_penalty_loss(fn=no_penalty, **kwargs):
  """Returns the penalty loss (penalty_lib.py:105-108)."""
  return utils.call_with_accepted_args(fn, **kwargs)
