"""S3GAN: auxiliary heads for the modular GAN (reference: gans/s3gan.py:39-321;
https://arxiv.org/abs/1903.02271) -- (1) a projection layer on D's features with the (inferred)
label, (2) a predictor (classifier) head that infers labels for unlabelled real examples, (3) the
rotation self-supervision of SSGAN.

Same constructor surface, variable scopes ("discriminator_rotation/score_classify",
"discriminator_predictor/predictor_linear", "discriminator_projection/kernel" -- all matched by D's
scope prefix, so they train with D) and create_loss semantics as the reference; arithmetic on the
HIP kernels of the base class plus cg_s3gan_labels, cg_softmax_xent_weighted and
cg_softmax_xent_eps."""
import numpy as np
import torch

from compare_gan_amd import gin
from compare_gan_amd.architectures import arch_ops as ops
from compare_gan_amd.gans import loss_lib
from compare_gan_amd.gans import modular_gan
from compare_gan_amd.gans.ssgan import rotate_images
from compare_gan_amd.hip import functional as Fn
from compare_gan_amd.hip import kernels as K

NUM_ROTATIONS = 4


@gin.configurable(blacklist=["kwargs"])
class S3GAN(modular_gan.ModularGAN):
  """S3GAN which enables auxiliary heads for the modular GAN (s3gan.py:39-96)."""

  def __init__(self, self_supervision="rotation", rotated_batch_fraction=gin.REQUIRED,
               weight_rotation_loss_d=1.0, weight_rotation_loss_g=0.2, project_y=False,
               use_predictor=False, use_soft_pred=False, weight_class_loss=1.0,
               use_soft_labels=False, **kwargs):
    super(S3GAN, self).__init__(**kwargs)
    if use_predictor and not project_y:
      raise ValueError("Using predictor requires projection.")
    assert self_supervision in {"none", "rotation"}
    self._self_supervision = self_supervision
    self._rotated_batch_fraction = rotated_batch_fraction
    self._weight_rotation_loss_d = weight_rotation_loss_d
    self._weight_rotation_loss_g = weight_rotation_loss_g
    self._project_y = project_y
    self._use_predictor = use_predictor
    self._use_soft_pred = use_soft_pred
    self._weight_class_loss = weight_class_loss
    self._use_soft_labels = use_soft_labels
    assert not self._deprecated_split_disc_calls, \
        "Splitting discriminator calls is not supported in S3GAN."
    self.rot_real_loss = self.rot_fake_loss = self.class_loss_real = None

  # -- heads (s3gan.py:98-176) ---------------------------------------------------------------------
  def get_class_embedding(self, y, embedding_dim, use_sn):
    """matmul(y, kernel) with "discriminator_projection/kernel" [num_classes, embedding_dim],
    glorot-normal, spectrally normalised when D is (s3gan.py:164-176)."""
    return ops.linear(y, embedding_dim, scope="discriminator_projection", use_bias=False,
                      use_sn=use_sn, kernel_initializer=ops.glorot_normal())

  def discriminator_with_additonal_heads(self, x, y, is_training):
    """(d_probs, d_logits, rotation_logits, aux_logits, is_label_available) (s3gan.py:98-162)."""
    d_probs, d_logits, x_rep = self.discriminator(x, y=y, is_training=is_training)
    use_sn = self.discriminator._spectral_norm     # pylint: disable=protected-access
    x_rep = ops.as_tensor(x_rep)
    assert x_rep.dim() == 2, x_rep.shape
    meta = x_rep.is_meta
    rotation_logits = None
    if "rotation" in self._self_supervision:
      with ops.variable_scope("discriminator_rotation"):
        rotation_logits = ops.linear(x_rep, NUM_ROTATIONS, scope="score_classify", use_sn=use_sn,
                                     out_f32=True)
    is_label_available = None
    if not self._project_y:
      if not meta and y is not None:
        _, is_label_available = K.s3gan_labels(None, y.contiguous(), False)
      return d_probs, d_logits, rotation_logits, None, is_label_available
    aux_logits = None
    if self._use_predictor:
      with ops.variable_scope("discriminator_predictor"):
        aux_logits = ops.linear(x_rep, y.shape[1], use_bias=True, scope="predictor_linear",
                                use_sn=use_sn, out_f32=True)
    if not meta:
      # y <- stop_gradient((1 - available) * y_predicted + available * y)
      y, is_label_available = K.s3gan_labels(
          None if aux_logits is None else aux_logits.detach().contiguous(), y.contiguous(),
          self._use_soft_pred)
    class_embedding = self.get_class_embedding(y=y, embedding_dim=x_rep.shape[-1], use_sn=use_sn)
    if not meta:
      d_logits = Fn.add_f32(d_logits, Fn.RowDotFn.apply(class_embedding, x_rep))
      d_probs = ops.output_head(d_logits, 0)
    return d_probs, d_logits, rotation_logits, aux_logits, is_label_available

  def _build_heads(self, x, d_out):
    y = None
    if self.conditional:
      y = torch.empty((x.shape[0], self._dataset.num_classes), dtype=torch.bfloat16,
                      device="meta")
    elif self._project_y:
      raise ValueError("S3GAN.project_y needs a conditional GAN (labels to project).")
    self.discriminator_with_additonal_heads(x, y, is_training=True)

  def merge_with_rotation_data(self, real, fake, real_labels, fake_labels, num_rot_examples):
    """The original data concatenated with its rotated versions (s3gan.py:178-196); returns the
    real half, the fake half and the labels of both."""
    bs = real.shape[0]
    real_rotated = rotate_images(real[bs - num_rot_examples:], rot90_scalars=(1, 2, 3))
    fake_rotated = rotate_images(fake[bs - num_rot_examples:], rot90_scalars=(1, 2, 3))
    all_labels = None
    if self.conditional:
      real_rotated_labels = real_labels[bs - num_rot_examples:].repeat(3, 1)
      fake_rotated_labels = fake_labels[bs - num_rot_examples:].repeat(3, 1)
      all_labels = torch.cat([real_labels, real_rotated_labels, fake_labels, fake_rotated_labels],
                             dim=0)
    return (torch.cat([real, real_rotated], dim=0), torch.cat([fake, fake_rotated], dim=0),
            all_labels)

  # -- losses (s3gan.py:198-321) -------------------------------------------------------------------
  def create_loss(self, features, labels, params=None, is_training=True):
    del params
    real_images = features["images"]
    real_labels = fake_labels = None
    if self.conditional:
      if self._use_soft_labels:
        assert labels.dim() == 2 and labels.shape[1] == self._dataset.num_classes, (
            "Need soft labels of dimension {} but got {}".format(
                self._dataset.num_classes, tuple(labels.shape)))
        real_labels = ops._to_bf16(labels)     # pylint: disable=protected-access
      else:
        real_labels = self._get_one_hot_labels(labels)
      fake_labels = self._get_one_hot_labels(features["sampled_labels"])
    # the reference recomputes G(z) here when the forwards were not generated jointly
    # (s3gan.py:232-238): the same tensor the sub-step already holds
    fake_images = features["generated"]
    bs = real_images.shape[0]
    rotation = self._self_supervision == "rotation"
    if self._self_supervision:
      assert bs % self._rotated_batch_fraction == 0, (
          "Rotated batch fraction is invalid: %d doesn't divide %d" % (
              self._rotated_batch_fraction, bs))
      rotated_bs = bs // self._rotated_batch_fraction
      num_rot_examples = rotated_bs // NUM_ROTATIONS
      assert num_rot_examples > 0
    a, b = getattr(self.discriminator, "input_affine", (1.0, 0.0))
    self.d_opt.join()
    if rotation:
      assert num_rot_examples <= bs, (num_rot_examples, bs)
      real_all, fake_all, all_labels = self.merge_with_rotation_data(
          real_images, fake_images, real_labels, fake_labels, num_rot_examples)
    else:
      real_all, fake_all, all_labels = real_images, fake_images, None
      if self.conditional:
        all_labels = torch.cat([real_labels, fake_labels], dim=0)
    all_features = Fn.stage_images(real_all, fake_all, a, b)
    d_predictions, d_logits, rot_logits, aux_logits, is_label_available = (
        self.discriminator_with_additonal_heads(x=all_features, y=all_labels,
                                                is_training=is_training))
    expected_batch_size = 2 * bs
    if rotation:
      expected_batch_size += 2 * (NUM_ROTATIONS - 1) * num_rot_examples
    if d_logits.shape[0] != expected_batch_size:
      raise ValueError("Batch size unexpected: got %r expected %r" % (
          d_logits.shape[0], expected_batch_size))
    half = expected_batch_size // 2
    prob_real, prob_fake = d_predictions[:half][:bs], d_predictions[half:][:bs]
    logits_real, logits_fake = d_logits[:half][:bs], d_logits[half:][:bs]
    self.d_loss, _, _, self.g_loss = loss_lib.get_losses(
        d_real=prob_real, d_fake=prob_fake, d_real_logits=logits_real, d_fake_logits=logits_fake)
    self.penalty_loss = None      # (the reference's S3GAN applies no penalty term)
    self.rot_real_loss = self.rot_fake_loss = self.class_loss_real = None
    if rotation:
      labels_rotated = torch.from_numpy(np.repeat(
          np.arange(NUM_ROTATIONS, dtype=np.int32), num_rot_examples)).to(real_images.device)
      real_loss = Fn.SoftmaxXentEpsFn.apply(
          rot_logits[:half][half - rotated_bs:].contiguous(), labels_rotated, 1e-10)
      fake_loss = Fn.SoftmaxXentEpsFn.apply(
          rot_logits[half:][half - rotated_bs:].contiguous(), labels_rotated, 1e-10)
      self.d_loss = Fn.add_f32(self.d_loss.reshape(1), real_loss.reshape(1), 1.0,
                               float(self._weight_rotation_loss_d)).reshape(())
      self.g_loss = Fn.add_f32(self.g_loss.reshape(1), fake_loss.reshape(1), 1.0,
                               float(self._weight_rotation_loss_g)).reshape(())
      self.rot_real_loss, self.rot_fake_loss = real_loss.detach(), fake_loss.detach()
    if self._use_predictor:
      real_aux_logits = aux_logits[:half][:bs].contiguous()
      weights = is_label_available[:half][:bs].contiguous()
      class_loss_real = Fn.SoftmaxXentWeightedFn.apply(real_aux_logits, real_labels.contiguous(),
                                                       weights)
      self.d_loss = Fn.add_f32(self.d_loss.reshape(1), class_loss_real.reshape(1), 1.0,
                               float(self._weight_class_loss)).reshape(())
      self.class_loss_real = class_loss_real.detach()
