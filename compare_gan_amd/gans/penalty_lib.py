"""Discriminator penalties (reference: gans/penalty_lib.py:28-108)."""
import torch

from compare_gan_amd import gin
from compare_gan_amd import utils
from compare_gan_amd.architectures import arch_ops as ops
from compare_gan_amd.hip import functional as Fn
from compare_gan_amd.hip import kernels as K
from compare_gan_amd.tpu import tpu_random


@gin.configurable
def no_penalty():
  return None   # tf.constant(0.0): nothing to add to d_loss


def _gradient_norm_penalty(discriminator, x_in, y, is_training, input_scale=1.0):
  """mean((sqrt(1e-4 + sum_{hwc} (d logits / d x)^2) - 1)^2)   (penalty_lib.py:50-55,74-81).

  x_in: staged bf16 images requiring grad, ALREADY mapped by the discriminator's input affine
  a * x + b (sndcgan.py:108 keeps `x * 2 - 1` inside D, so the reference differentiates w.r.t. the
  [0,1] image): d logits / d x = a * d logits / d x_in, hence `input_scale` = a.  The inner backward
  runs with create_graph=True so the penalty is differentiable w.r.t. D's weights; every op on that
  path is a HIP kernel."""
  with ops.twice_differentiable():     # this call's backward is differentiated again below
    logits = discriminator(x_in, y=y, is_training=is_training, reuse=True)[1]
  ones = torch.ones_like(logits)   # d(sum logits)/d logits: a constant fill, not arithmetic
  with Fn.only_input_grads():
    gradients, = torch.autograd.grad(logits, [x_in], grad_outputs=ones, create_graph=True)
  if gradients.dtype != torch.float32:
    gradients = Fn.ToF32Fn.apply(gradients)
  if input_scale != 1.0:
    gradients = Fn.ScaleF32Fn.apply(gradients, float(input_scale))
  return Fn.GradientPenaltyFn.apply(gradients)


@gin.configurable(whitelist=[])
def dragan_penalty(discriminator, x, y, is_training):
  """DRAGAN gradient penalty (penalty_lib.py:33-56): x_noisy = clip(x + std(x) * (U[0,1) - 0.5),
  0, 1) with the standard deviation over ALL elements of x, then the gradient-norm penalty."""
  x = x.contiguous()
  u = tpu_random.uniform(list(x.shape), name="dragan_penalty/random_uniform/%d" %
                         tpu_random.sub_step(), device=x.device)
  a, b = getattr(discriminator, "input_affine", (1.0, 0.0))
  x_noisy = K.dragan_perturb(x, u, K.moments_f32(x), a, b)
  x_noisy.requires_grad_(True)
  return _gradient_norm_penalty(discriminator, x_noisy, y, is_training, input_scale=a)


@gin.configurable(whitelist=[])
def wgangp_penalty(discriminator, x, x_fake, y, is_training):
  """WGAN gradient penalty (penalty_lib.py:59-82).  x, x_fake: fp32 images in [0,1]."""
  alpha = tpu_random.uniform([x.shape[0]], name=alpha_name(tpu_random.sub_step()),
                             device=x.device)
  interpolates = K.interpolate(x.contiguous(), x_fake.detach().contiguous(), alpha)
  a, b = getattr(discriminator, "input_affine", (1.0, 0.0))
  if (a, b) != (1.0, 0.0):
    interpolates = K.axpby(interpolates, a, None, 0.0) if b == 0.0 else _affine(interpolates, a, b)
  interpolates.requires_grad_(True)
  return _gradient_norm_penalty(discriminator, interpolates, y, is_training, input_scale=a)


def alpha_name(sub_step):
  """Random-op name of the interpolation weights of sub-step `sub_step` (one op per sub-step, as in
  the reference's unrolled graph; sub-step 0 keeps the plain name)."""
  return "wgangp_penalty/alpha" if sub_step == 0 else "wgangp_penalty/alpha/%d" % sub_step


def _affine(x, a, b):
  return K.cast_f32_to_bf16(K.cast_bf16_to_f32(x), a, b)


@gin.configurable(whitelist=[])
def l2_penalty(discriminator):
  """Mean over D's kernels (fully connected, conv2d, deconv2d; biases excluded) of
  tf.nn.l2_loss = sum(w^2) / 2 (penalty_lib.py:85-102).  The RAW variables are penalised, as in
  the reference (`discriminator.trainable_variables`), not their spectrally normalised forms."""
  with ops.use_store(ops.current_store()):
    kernels = [v for n, v in discriminator.trainable_variables if n.endswith("/kernel")]
  if not kernels:
    raise ValueError("l2_penalty: the discriminator has no kernels")
  if kernels[0].is_meta:
    return kernels[0].new_zeros(())
  # one multi-tensor reduction over all kernels (three launches instead of three per kernel)
  return Fn.L2LossMeanFn.apply(*kernels)


@gin.configurable("penalty", whitelist=["fn"])
def get_penalty_loss(fn=no_penalty, **kwargs):
  """Returns the penalty loss (penalty_lib.py:105-108)."""
  return utils.call_with_accepted_args(fn, **kwargs)
