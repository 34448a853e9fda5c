"""Self-supervised GAN with the auxiliary rotation loss (reference: gans/ssgan.py:39-226;
http://arxiv.org/abs/1811.11212).

Same constructor surface and create_loss semantics as the reference; the arithmetic runs on the
HIP kernels of the base class plus cg_softmax_xent_eps (the rotation cross-entropy with the
reference's log(p + 1e-10)).  The rotations themselves are index permutations (flip / transpose:
data movement, like torch.cat)."""
import numpy as np
import torch

from compare_gan_amd import gin
from compare_gan_amd.architectures import arch_ops as ops
from compare_gan_amd.gans import loss_lib
from compare_gan_amd.gans import modular_gan
from compare_gan_amd.gans import penalty_lib
from compare_gan_amd.hip import functional as Fn
from compare_gan_amd.tpu import tpu_ops
from compare_gan_amd.tpu import tpu_random

NUM_ROTATIONS = 4


def rotate_images(images, rot90_scalars=(0, 1, 2, 3)):
  """The input images and their 90, 180 and 270 degree rotations (gans/utils.py:38-50): NHWC,
  the requested rotations stacked rotation-major into one batch."""
  rotated = [
      lambda x: x,                                   # 0 degrees
      lambda x: x.transpose(1, 2).flip(1),           # flip_up_down(transpose_image(x))
      lambda x: x.flip(1).flip(2),                   # flip_left_right(flip_up_down(x))
      lambda x: x.flip(1).transpose(1, 2),           # transpose_image(flip_up_down(x))
  ]
  return torch.cat([rotated[i](images) for i in rot90_scalars], dim=0).contiguous()


@gin.configurable(blacklist=["kwargs"])
class SSGAN(modular_gan.ModularGAN):
  """Self-Supervised GAN (ssgan.py:39-77)."""

  def __init__(self, self_supervision="rotation_gan", rotated_batch_size=gin.REQUIRED,
               weight_rotation_loss_d=1.0, weight_rotation_loss_g=0.2, **kwargs):
    super(SSGAN, self).__init__(**kwargs)
    self._self_supervision = self_supervision
    self._rotated_batch_size = rotated_batch_size
    self._weight_rotation_loss_d = weight_rotation_loss_d
    self._weight_rotation_loss_g = weight_rotation_loss_g
    assert not self._deprecated_split_disc_calls, \
        "Splitting discriminator calls is not supported in SSGAN."
    self.c_real_loss = self.c_fake_loss = None

  # -- the rotation head ---------------------------------------------------------------------------
  def _rotation_head(self, final, batch):
    """linear(reshape(final, [batch, -1]), 4) under "discriminator_rotation/score_classify"
    (ssgan.py:95-101): the scope name contains "discriminator", so the head trains with D."""
    use_sn = self.discriminator._spectral_norm     # pylint: disable=protected-access
    final = ops.as_tensor(final)
    with ops.variable_scope("discriminator_rotation"):
      return ops.linear(final.reshape(batch, -1), NUM_ROTATIONS, scope="score_classify",
                        use_sn=use_sn, out_f32=True)

  def discriminator_with_rotation_head(self, x, y, is_training):
    """(real_probs, real_scores, rotation_scores) (ssgan.py:79-102)."""
    real_probs, real_scores, final = self.discriminator(x=x, y=y, is_training=is_training)
    return real_probs, real_scores, self._rotation_head(final, x.shape[0])

  def _build_heads(self, x, d_out):
    self._rotation_head(d_out[2], x.shape[0])

  def _rotation_sizes(self, bs):
    num_replicas = tpu_ops.num_replicas()
    assert self._rotated_batch_size % num_replicas == 0
    rotated_bs = self._rotated_batch_size // num_replicas       # per replica
    assert rotated_bs % 4 == 0
    num_rotated_examples = rotated_bs // 4                      # each gets rotated 3 times
    assert num_rotated_examples <= bs, (num_rotated_examples, bs)
    return rotated_bs, num_rotated_examples

  # -- losses (ssgan.py:104-226) -------------------------------------------------------------------
  def create_loss(self, features, labels, params=None, is_training=True):
    del params
    images = features["images"]
    generated = features["generated"]
    if self.conditional:
      y = self._get_one_hot_labels(labels)
      sampled_y = self._get_one_hot_labels(features["sampled_labels"])
    else:
      y = sampled_y = all_y = None
    bs = images.shape[0]
    rotation = "rotation" in self._self_supervision
    a, b = getattr(self.discriminator, "input_affine", (1.0, 0.0))
    self.d_opt.join()
    if rotation:
      rotated_bs, nrot = self._rotation_sizes(bs)
      # first the upright images, then rotated_bs * 3 / 4 images at the 3 other angles
      images_rotated = rotate_images(images[bs - nrot:], rot90_scalars=(1, 2, 3))
      generated_rotated = rotate_images(generated[bs - nrot:], rot90_scalars=(1, 2, 3))
      rotate_labels = torch.from_numpy(
          np.repeat(np.arange(NUM_ROTATIONS, dtype=np.int32), nrot)).to(images.device)
      real_all = torch.cat([images, images_rotated], dim=0)
      fake_all = torch.cat([generated, generated_rotated], dim=0)
      if self.conditional:
        y_rotated = y[bs - nrot:].repeat(3, 1)
        sampled_y_rotated = y[bs - nrot:].repeat(3, 1)      # (sic: the reference tiles y twice)
        all_y = torch.cat([y, y_rotated, sampled_y, sampled_y_rotated], dim=0)
    else:
      real_all, fake_all = images, generated
      if self.conditional:
        all_y = torch.cat([y, sampled_y], dim=0)
    all_images = Fn.stage_images(real_all, fake_all, a, b)
    d_all, d_all_logits, c_all_logits = self.discriminator_with_rotation_head(
        all_images, y=all_y, is_training=is_training)
    half = d_all.shape[0] // 2
    d_real, d_fake = d_all[:half][:bs], d_all[half:][:bs]
    d_real_logits, d_fake_logits = d_all_logits[:half][:bs], d_all_logits[half:][:bs]
    c_real_logits, c_fake_logits = c_all_logits[:half], c_all_logits[half:]
    self.d_loss, _, _, self.g_loss = loss_lib.get_losses(
        d_real=d_real, d_fake=d_fake, d_real_logits=d_real_logits, d_fake_logits=d_fake_logits)
    self.penalty_loss = None
    tpu_random.set_sub_step(features.get("_sub_step", 0))
    if torch.is_grad_enabled() and not features.get("_generator_step", False):
      penalty_loss = penalty_lib.get_penalty_loss(
          x=images, x_fake=generated, y=y, is_training=is_training,
          discriminator=self.discriminator)
      if penalty_loss is not None:
        self.penalty_loss = penalty_loss
        self.d_loss = Fn.add_f32(self.d_loss.reshape(1), penalty_loss.reshape(1), 1.0,
                                 float(self._lambda)).reshape(())
    if rotation:
      # an even piece for every rotation angle: the last rotated_bs rows of each half
      c_real_loss = Fn.SoftmaxXentEpsFn.apply(
          c_real_logits[half - rotated_bs:].contiguous(), rotate_labels, 1e-10)
      c_fake_loss = Fn.SoftmaxXentEpsFn.apply(
          c_fake_logits[half - rotated_bs:].contiguous(), rotate_labels, 1e-10)
      gan_w = 0.0 if self._self_supervision == "rotation_only" else 1.0
      self.d_loss = Fn.add_f32(self.d_loss.reshape(1), c_real_loss.reshape(1), gan_w,
                               float(self._weight_rotation_loss_d)).reshape(())
      self.g_loss = Fn.add_f32(self.g_loss.reshape(1), c_fake_loss.reshape(1), gan_w,
                               float(self._weight_rotation_loss_g)).reshape(())
      self.c_real_loss, self.c_fake_loss = c_real_loss.detach(), c_fake_loss.detach()
    else:
      self.c_real_loss = self.c_fake_loss = None
