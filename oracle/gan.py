"""ORACLE -- test infrastructure only.

Restatement of gans/loss_lib.py:53-148, gans/penalty_lib.py:28-102, the optimiser semantics the
reference delegates to TF (tf.train.AdamOptimizer, tf.train.ExponentialMovingAverage; SURVEY
App. A.5, "parity unpinned") and the training-step structure of gans/modular_gan.py:428-670.
"""
import math

import torch
import torch.nn.functional as F


# ---- losses (loss_lib.py) ----------------------------------------------------------------------
def _sce(logits, labels):
  """tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log(1+exp(-|x|))."""
  return torch.clamp(logits, min=0) - logits * labels + torch.log1p(torch.exp(-logits.abs()))


def non_saturating(d_real_logits, d_fake_logits, d_real=None, d_fake=None):
  d_loss_real = _sce(d_real_logits, torch.ones_like(d_real_logits)).mean()      # :69-71
  d_loss_fake = _sce(d_fake_logits, torch.zeros_like(d_fake_logits)).mean()     # :72-74
  g_loss = _sce(d_fake_logits, torch.ones_like(d_fake_logits)).mean()           # :76-78
  return d_loss_real + d_loss_fake, d_loss_real, d_loss_fake, g_loss


def wasserstein(d_real_logits, d_fake_logits, d_real=None, d_fake=None):
  d_loss_real = -d_real_logits.mean()                                           # :98
  d_loss_fake = d_fake_logits.mean()                                            # :99
  return d_loss_real + d_loss_fake, d_loss_real, d_loss_fake, -d_loss_fake


def least_squares(d_real, d_fake, d_real_logits=None, d_fake_logits=None):
  d_loss_real = ((d_real - 1.0) ** 2).mean()                                    # :121
  d_loss_fake = (d_fake ** 2).mean()                                            # :122
  return (0.5 * (d_loss_real + d_loss_fake), d_loss_real, d_loss_fake,
          0.5 * ((d_fake - 1.0) ** 2).mean())


def hinge(d_real_logits, d_fake_logits, d_real=None, d_fake=None):
  d_loss_real = torch.relu(1.0 - d_real_logits).mean()                          # :144
  d_loss_fake = torch.relu(1.0 + d_fake_logits).mean()                          # :145
  return d_loss_real + d_loss_fake, d_loss_real, d_loss_fake, -d_fake_logits.mean()


LOSSES = {"non_saturating": non_saturating, "wasserstein": wasserstein,
          "least_squares": least_squares, "hinge": hinge}


def get_losses(fn, d_real, d_fake, d_real_logits, d_fake_logits):
  """loss_lib.py:151-154."""
  return LOSSES[fn](d_real=d_real, d_fake=d_fake, d_real_logits=d_real_logits,
                    d_fake_logits=d_fake_logits)


# ---- penalties (penalty_lib.py) ----------------------------------------------------------------
def wgangp_penalty(discriminator, x, x_fake, y, is_training, alpha):
  """penalty_lib.py:59-82; alpha [B,1,1,1] ~ U[0,1) is supplied by the caller (tpu_random)."""
  interpolates = (x + alpha * (x_fake - x)).detach().requires_grad_(True)
  logits = discriminator(interpolates, y, is_training)[1]
  gradients = torch.autograd.grad(logits.sum(), interpolates, create_graph=True)[0]
  slopes = torch.sqrt(0.0001 + (gradients ** 2).sum(dim=(1, 2, 3)))
  return ((slopes - 1.0) ** 2).mean()


def dragan_penalty(discriminator, x, y, is_training, noise):
  """penalty_lib.py:33-56; noise ~ U[0,1) with x's shape is supplied by the caller."""
  std = torch.sqrt(x.var(unbiased=False))
  x_noisy = torch.clamp(x + std * (noise - 0.5), 0.0, 1.0).detach().requires_grad_(True)
  logits = discriminator(x_noisy, y, is_training)[1]
  gradients = torch.autograd.grad(logits.sum(), x_noisy, create_graph=True)[0]
  slopes = torch.sqrt(0.0001 + (gradients ** 2).sum(dim=(1, 2, 3)))
  return ((slopes - 1.0) ** 2).mean()


def l2_penalty(d_kernels):
  """penalty_lib.py:85-102: mean over kernels of tf.nn.l2_loss = sum(w^2)/2."""
  return torch.stack([(w ** 2).sum() / 2 for w in d_kernels]).mean()


# ---- optimiser (TF1 AdamOptimizer) ------------------------------------------------
class TFAdam(object):
  """m <- b1 m + (1-b1) g; v <- b2 v + (1-b2) g^2;
  theta <- theta - lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)   (eps OUTSIDE the corrected root)."""

  def __init__(self, params, lr, beta1=0.9, beta2=0.999, epsilon=1e-8):
    self.params = list(params)
    self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, epsilon
    self.t = 0
    self.m = [torch.zeros_like(p) for p in self.params]
    self.v = [torch.zeros_like(p) for p in self.params]

  def step(self, grads):
    self.t += 1
    lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
    with torch.no_grad():
      for p, g, m, v in zip(self.params, grads, self.m, self.v):
        m.mul_(self.b1).add_((1 - self.b1) * g)
        v.mul_(self.b2).add_((1 - self.b2) * g * g)
        p.sub_(lr_t * m / (v.sqrt() + self.eps))


def ema_update(shadow, params, decay):
  """tf.train.ExponentialMovingAverage.apply: s <- s - (1-decay)(s - p)  (modular_gan.py:498-508)."""
  with torch.no_grad():
    for s, p in zip(shadow, params):
      s.sub_((1 - decay) * (s - p))
