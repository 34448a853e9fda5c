"""ORACLE -- test infrastructure only.

CPU restatement of the training-step structure of the reference's gans/modular_gan.py:
  _split_inputs_and_generate_samples :428-469, _train_discriminator :471-485,
  _train_generator :487-510, model_fn (unrolled) :568-584, create_loss :618-670.
No reference test pins numbers for this path ("parity unpinned": modular_gan_test.py only checks
that one step runs and the step counters, :142-177); the counters are pinned in
tests/test_modular_gan_gpu.py.
"""
import torch
import torch.nn.functional as F

from oracle import architectures as A
from oracle import gan as ogan


class OracleGAN(object):

  def __init__(self, vs, architecture, g_cfg, d_cfg, image_shape, loss="non_saturating",
               penalty="no_penalty", lamba=1.0, disc_iters=1, conditional=False, num_classes=None,
               g_lr=0.0002, d_lr=None, beta1=0.9, beta2=0.999, g_use_ema=False,
               ema_decay=0.9999, ema_start_step=40000, joint_gen_for_disc=False):
    self.vs = vs
    # experimental_joint_gen_for_disc (modular_gan.py:444-463): ONE generator call on the z of all
    # discriminator sub-steps (batch norm statistics over the joint batch), split afterwards
    self.joint_gen_for_disc = joint_gen_for_disc
    self.arch = architecture
    self.g_cfg, self.d_cfg = g_cfg, d_cfg
    self.image_shape = image_shape
    self.loss, self.penalty, self.lamba = loss, penalty, lamba
    self.disc_iters = disc_iters
    self.conditional, self.num_classes = conditional, num_classes
    self.g_lr, self.d_lr = g_lr, g_lr if d_lr is None else d_lr
    self.beta1, self.beta2 = beta1, beta2
    self.g_use_ema, self.ema_decay, self.ema_start_step = g_use_ema, ema_decay, ema_start_step
    self.global_step = 0
    self.global_step_disc = 0
    self.g_opt = self.d_opt = None
    self.ema = None

  def one_hot(self, labels):
    return F.one_hot(labels.long(), self.num_classes).to(self.vs.dtype)

  def G(self, z, y, is_training=True):
    return A.GENERATORS[self.arch](self.vs, self.g_cfg, z, y, is_training, self.image_shape)

  def D(self, x, y, is_training=True):
    return A.DISCRIMINATORS[self.arch](self.vs, self.d_cfg, x, y, is_training)

  def g_vars(self):
    return [self.vs.vars[n] for n in self.vs.trainable if n.startswith("generator")]

  def d_vars(self):
    return [self.vs.vars[n] for n in self.vs.trainable if n.startswith("discriminator")]

  def d_var_names(self):
    return [n for n in self.vs.trainable if n.startswith("discriminator")]

  def create_loss(self, images, generated, labels, sampled_labels, alpha=None, with_penalty=True):
    """modular_gan.py:618-670 -> (d_loss, g_loss, d_all_logits)."""
    if self.conditional:
      y, sampled_y = self.one_hot(labels), self.one_hot(sampled_labels)
      all_y = torch.cat([y, sampled_y], 0)
    else:
      y = sampled_y = all_y = None
    all_images = torch.cat([images, generated], 0)                       # :657
    d_all, d_all_logits, _ = self.D(all_images, all_y)                   # :658-659
    b = images.shape[0]
    d_loss, _, _, g_loss = ogan.get_losses(self.loss, d_all[:b], d_all[b:], d_all_logits[:b],
                                           d_all_logits[b:])             # :663-665
    if with_penalty and self.penalty == "wgangp_penalty":
      pen = ogan.wgangp_penalty(lambda x, yy, t: self.D(x, yy, t), images, generated.detach(), y,
                                True, alpha.reshape(-1, 1, 1, 1))
      d_loss = d_loss + self.lamba * pen                                 # :670
    elif with_penalty and self.penalty == "dragan_penalty":
      # `alpha` carries the U[0,1) noise of the images' shape (penalty_lib.py:47)
      pen = ogan.dragan_penalty(lambda x, yy, t: self.D(x, yy, t), images, y, True, alpha)
      d_loss = d_loss + self.lamba * pen
    elif with_penalty and self.penalty == "l2_penalty":
      kernels = [v for n, v in zip(self.d_var_names(), self.d_vars()) if n.endswith("/kernel")]
      d_loss = d_loss + self.lamba * ogan.l2_penalty(kernels)
    return d_loss, g_loss, d_all_logits

  def _ensure_opts(self):
    if self.g_opt is None:
      self.g_opt = ogan.TFAdam(self.g_vars(), self.g_lr, self.beta1, self.beta2)
      self.d_opt = ogan.TFAdam(self.d_vars(), self.d_lr, self.beta1, self.beta2)
      if self.g_use_ema:
        self.ema = [p.detach().clone() for p in self.g_vars()]

  def train_step(self, subs):
    """subs: list of disc_iters+1 dicts {images, z, labels, sampled_labels, alpha}."""
    self._ensure_opts()
    d_losses = []
    joint = None
    if self.joint_gen_for_disc:
      # modular_gan.py:451-458: generator(z[:batch_size * disc_iters]) then tf.split
      with torch.no_grad():
        z = torch.cat([s["z"] for s in subs[:self.disc_iters]], dim=0)
        sy = None
        if self.conditional:
          sy = self.one_hot(torch.cat([s["sampled_labels"] for s in subs[:self.disc_iters]]))
        joint = torch.chunk(self.G(z, sy), self.disc_iters, dim=0)
    for i in range(self.disc_iters):
      s = subs[i]
      with torch.no_grad():
        if joint is not None:
          generated = joint[i]
        else:
          sy = self.one_hot(s["sampled_labels"]) if self.conditional else None
          generated = self.G(s["z"], sy)
      d_loss, _, _ = self.create_loss(s["images"], generated, s.get("labels"),
                                      s.get("sampled_labels"), s.get("alpha"))
      grads = torch.autograd.grad(d_loss, self.d_vars())
      self.d_opt.step(grads)
      self.global_step_disc += 1
      d_losses.append(float(d_loss.detach()))
    s = subs[-1]
    sy = self.one_hot(s["sampled_labels"]) if self.conditional else None
    generated = self.G(s["z"], sy)
    _, g_loss, _ = self.create_loss(s["images"], generated, s.get("labels"),
                                    s.get("sampled_labels"), with_penalty=False)
    grads = torch.autograd.grad(g_loss, self.g_vars())
    self.g_opt.step(grads)
    if self.g_use_ema:
      decay = self.ema_decay if self.global_step >= self.ema_start_step else 0.0
      ogan.ema_update(self.ema, self.g_vars(), decay)
    self.global_step += 1
    return d_losses, float(g_loss.detach())

  def train_step_not_unrolled(self, sub):
    """One session.run of the NOT unrolled graph (modular_gan.py:533-584, use_tpu=False): one
    generator forward shared by both updates, a D update, then -- when disc_step % disc_iters == 0
    after the increment (:566-569) -- a G update through a FRESH forward of the updated D."""
    self._ensure_opts()
    s = sub
    sy = self.one_hot(s["sampled_labels"]) if self.conditional else None
    generated = self.G(s["z"], sy)                                       # :465-467, once
    d_loss, _, _ = self.create_loss(s["images"], generated.detach(), s.get("labels"),
                                    s.get("sampled_labels"), s.get("alpha"))
    grads = torch.autograd.grad(d_loss, self.d_vars())
    self.d_opt.step(grads)
    self.global_step_disc += 1
    g_val = 0.0
    if self.global_step_disc % self.disc_iters == 0:
      _, g_loss, _ = self.create_loss(s["images"], generated, s.get("labels"),
                                      s.get("sampled_labels"), with_penalty=False)
      grads = torch.autograd.grad(g_loss, self.g_vars())
      self.g_opt.step(grads)
      if self.g_use_ema:
        decay = self.ema_decay if self.global_step >= self.ema_start_step else 0.0
        ogan.ema_update(self.ema, self.g_vars(), decay)
      self.global_step += 1
      g_val = float(g_loss.detach())
    return float(d_loss.detach()), g_val


def rotate_images(images, rot90_scalars=(0, 1, 2, 3)):
  """gans/utils.py:38-50 on NHWC tensors: out[n,i,j] = x[n,j,H-1-i] (90), x[n,H-1-i,W-1-j] (180),
  x[n,H-1-j,i] (270), written with explicit index arithmetic; rotation-major batch."""
  n, h, w, c = images.shape
  assert h == w
  idx = torch.arange(h)
  ii, jj = torch.meshgrid(idx, idx, indexing="ij")
  outs = []
  for k in rot90_scalars:
    if k == 0:
      outs.append(images)
    elif k == 1:      # flip_up_down(transpose_image(x))
      outs.append(images[:, jj, h - 1 - ii, :])
    elif k == 2:      # flip_left_right(flip_up_down(x))
      outs.append(images[:, h - 1 - ii, w - 1 - jj, :])
    else:             # transpose_image(flip_up_down(x))
      outs.append(images[:, h - 1 - jj, ii, :])
  return torch.cat(outs, dim=0)


class OracleSSGAN(OracleGAN):
  """gans/ssgan.py:39-226: the rotation head on D's features and the two rotation losses."""

  def __init__(self, *args, rotated_batch_size=4, weight_rotation_loss_d=1.0,
               weight_rotation_loss_g=0.2, self_supervision="rotation_gan", **kwargs):
    super(OracleSSGAN, self).__init__(*args, **kwargs)
    self.rotated_batch_size = rotated_batch_size
    self.w_d, self.w_g = weight_rotation_loss_d, weight_rotation_loss_g
    self.self_supervision = self_supervision

  def create_loss(self, images, generated, labels, sampled_labels, alpha=None, with_penalty=True):
    from oracle import arch_ops as ops
    assert not self.conditional
    bs = images.shape[0]
    rotated_bs = self.rotated_batch_size
    nrot = rotated_bs // 4
    images_rot = rotate_images(images[bs - nrot:], (1, 2, 3))                # ssgan.py:150-153
    generated_rot = rotate_images(generated[bs - nrot:], (1, 2, 3))
    rotate_labels = torch.arange(4).repeat_interleave(nrot)                  # :157-159
    all_images = torch.cat([images, images_rot, generated, generated_rot], 0)  # :161-162
    d_all, d_all_logits, final = self.D(all_images, None)
    c_all_logits = ops.linear(self.vs, final.reshape(all_images.shape[0], -1), 4,
                              "discriminator_rotation/score_classify", self.d_cfg.sn_cfg,
                              use_sn=self.d_cfg.spectral_norm, out_f32=True)  # :95-101
    half = d_all.shape[0] // 2
    d_loss, _, _, g_loss = ogan.get_losses(self.loss, d_all[:half][:bs], d_all[half:][:bs],
                                           d_all_logits[:half][:bs], d_all_logits[half:][:bs])
    assert self.penalty == "no_penalty"
    c_real, c_fake = c_all_logits[:half][half - rotated_bs:], c_all_logits[half:][half - rotated_bs:]
    onehot = F.one_hot(rotate_labels, 4).to(c_real.dtype)
    c_real_loss = -(onehot * torch.log(torch.softmax(c_real, -1) + 1e-10)).sum(1).mean()  # :191-199
    c_fake_loss = -(onehot * torch.log(torch.softmax(c_fake, -1) + 1e-10)).sum(1).mean()
    if self.self_supervision == "rotation_only":
      d_loss, g_loss = d_loss * 0.0, g_loss * 0.0
    self.c_real_loss, self.c_fake_loss = float(c_real_loss.detach()), float(c_fake_loss.detach())
    return d_loss + c_real_loss * self.w_d, g_loss + c_fake_loss * self.w_g, d_all_logits

  def d_vars(self):
    return [self.vs.vars[n] for n in self.d_var_names()]

  def d_var_names(self):
    # "discriminator_rotation/..." matches the scope prefix "discriminator" (abstract_arch.py:43-45)
    return [n for n in self.vs.trainable if n.startswith("discriminator")]


class OracleS3GAN(OracleGAN):
  """gans/s3gan.py:39-321: projection on the (inferred) label, predictor head, rotation head."""

  def __init__(self, *args, self_supervision="rotation", rotated_batch_fraction=2,
               weight_rotation_loss_d=1.0, weight_rotation_loss_g=0.2, project_y=False,
               use_predictor=False, use_soft_pred=False, weight_class_loss=1.0, **kwargs):
    super(OracleS3GAN, self).__init__(*args, **kwargs)
    if use_predictor and not project_y:
      raise ValueError("Using predictor requires projection.")
    self.self_supervision = self_supervision
    self.rotated_batch_fraction = rotated_batch_fraction
    self.w_rot_d, self.w_rot_g = weight_rotation_loss_d, weight_rotation_loss_g
    self.project_y, self.use_predictor = project_y, use_predictor
    self.use_soft_pred, self.w_class = use_soft_pred, weight_class_loss

  def one_hot(self, labels):
    """Label -1 = no label: an all-zero row (s3gan.py:112)."""
    lab = labels.long()
    oh = F.one_hot(lab.clamp(min=0), self.num_classes).to(self.vs.dtype)
    return oh * (lab >= 0).to(self.vs.dtype).unsqueeze(1)

  def heads(self, x, y):
    """s3gan.py:98-162."""
    from oracle import arch_ops as ops
    d_probs, d_logits, x_rep = self.D(x, y)
    sn, sn_cfg = self.d_cfg.spectral_norm, self.d_cfg.sn_cfg
    avail = (y.sum(dim=1, keepdim=True) > 0.5).to(y.dtype)                      # :118-119
    rot = None
    if "rotation" in self.self_supervision:
      rot = ops.linear(self.vs, x_rep, 4, "discriminator_rotation/score_classify", sn_cfg,
                       use_sn=sn, out_f32=True)                                  # :123-130
    if not self.project_y:
      return d_probs, d_logits, rot, None, avail
    aux = None
    if self.use_predictor:
      aux = ops.linear(self.vs, x_rep, y.shape[1], "discriminator_predictor/predictor_linear",
                       sn_cfg, use_sn=sn, out_f32=True)                          # :138-141
      if self.use_soft_pred:
        y_pred = torch.softmax(aux, dim=1)
      else:
        y_pred = F.one_hot(aux.argmax(dim=1), aux.shape[1]).to(y.dtype)
      y = ((1.0 - avail) * y_pred + avail * y).detach()                          # :147-148
    emb = ops.linear(self.vs, self.vs.q(y), x_rep.shape[-1], "discriminator_projection", sn_cfg,
                     use_sn=sn, use_bias=False, kernel_init=self.vs.glorot_normal_init())
    d_logits = d_logits + (emb * x_rep).sum(dim=1, keepdim=True)                 # :157
    return torch.sigmoid(d_logits), d_logits, rot, aux, avail

  def create_loss(self, images, generated, labels, sampled_labels, alpha=None, with_penalty=True):
    assert self.conditional
    bs = images.shape[0]
    real_y, fake_y = self.one_hot(labels), self.one_hot(sampled_labels)
    rotation = self.self_supervision == "rotation"
    rotated_bs = bs // self.rotated_batch_fraction
    nrot = rotated_bs // 4
    if rotation:                                                                 # :178-196
      real_rot = rotate_images(images[bs - nrot:], (1, 2, 3))
      fake_rot = rotate_images(generated[bs - nrot:], (1, 2, 3))
      all_x = torch.cat([images, real_rot, generated, fake_rot], 0)
      all_y = torch.cat([real_y, real_y[bs - nrot:].repeat(3, 1), fake_y,
                         fake_y[bs - nrot:].repeat(3, 1)], 0)
    else:
      all_x, all_y = torch.cat([images, generated], 0), torch.cat([real_y, fake_y], 0)
    d_pred, d_logits, rot, aux, avail = self.heads(all_x, all_y)
    half = d_logits.shape[0] // 2
    d_loss, _, _, g_loss = ogan.get_losses(self.loss, d_pred[:half][:bs], d_pred[half:][:bs],
                                           d_logits[:half][:bs], d_logits[half:][:bs])
    self.rot_real_loss = self.rot_fake_loss = self.class_loss_real = None
    if rotation:                                                                 # :283-296
      lab = torch.arange(4).repeat_interleave(nrot)
      onehot = F.one_hot(lab, 4).to(rot.dtype)
      rr, rf = rot[:half][half - rotated_bs:], rot[half:][half - rotated_bs:]
      real_loss = -(onehot * torch.log(torch.softmax(rr, -1) + 1e-10)).sum(1).mean()
      fake_loss = -(onehot * torch.log(torch.softmax(rf, -1) + 1e-10)).sum(1).mean()
      d_loss = d_loss + real_loss * self.w_rot_d
      g_loss = g_loss + fake_loss * self.w_rot_g
      self.rot_real_loss, self.rot_fake_loss = float(real_loss.detach()), float(fake_loss.detach())
    if self.use_predictor:                                                       # :306-314
      w = avail[:half][:bs].reshape(-1)
      ce = -(real_y * torch.log_softmax(aux[:half][:bs], dim=1)).sum(1)
      cnt = (w != 0).sum().clamp(min=1).to(ce.dtype)
      class_loss = (w * ce).sum() / cnt          # Reduction.SUM_BY_NONZERO_WEIGHTS
      d_loss = d_loss + self.w_class * class_loss
      self.class_loss_real = float(class_loss.detach())
    return d_loss, g_loss, d_logits

  def d_vars(self):
    return [self.vs.vars[n] for n in self.d_var_names()]

  def d_var_names(self):
    return [n for n in self.vs.trainable if n.startswith("discriminator")]
