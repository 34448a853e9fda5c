"""GAN losses (reference: gans/loss_lib.py:28-154): same gin configurables, argument names and
shape checks; the arithmetic (loss values and d loss / d logits) is one HIP kernel."""
from compare_gan_amd import gin
from compare_gan_amd import utils
from compare_gan_amd.hip import functional as Fn
from compare_gan_amd.hip import kernels as K


def check_dimensions(d_real, d_fake, d_real_logits, d_fake_logits):
  """Checks the shapes and ranks of logits and prediction tensors (loss_lib.py:28-50)."""
  def _check_pair(a, b):
    if a != b:
      raise ValueError("Shape mismatch: %s vs %s." % (a, b))
    if len(a) != 2 or len(b) != 2:
      raise ValueError("Rank: expected 2, got %s and %s" % (len(a), len(b)))

  if (d_real is not None) and (d_fake is not None):
    _check_pair(list(d_real.shape), list(d_fake.shape))
  if (d_real_logits is not None) and (d_fake_logits is not None):
    _check_pair(list(d_real_logits.shape), list(d_fake_logits.shape))
  if (d_real is not None) and (d_real_logits is not None):
    _check_pair(list(d_real.shape), list(d_real_logits.shape))


def _run(kind, d_real_logits, d_fake_logits, d_real, d_fake):
  """Joins the two [B,1] halves (modular_gan.py:660-661 split them from one D call) and runs the
  fused loss kernel; autograd routes d loss / d logits back through the slices."""
  import torch
  check_dimensions(d_real, d_fake, d_real_logits, d_fake_logits)
  if d_real_logits.is_meta:
    z = d_real_logits.new_empty(())
    return z, z, z, z
  joined = getattr(d_real_logits, "_cg_joined", None)
  if (joined is not None and joined is getattr(d_fake_logits, "_cg_joined", None) and
      joined.shape[0] == d_real_logits.shape[0] + d_fake_logits.shape[0]):
    # the halves are views of ONE discriminator call's logits (modular_gan.py:660-661): the kernel
    # reads that tensor directly -- no concatenation forward, no zero-fill / copy / add of the two
    # slice gradients backward
    all_logits = joined
  else:
    all_logits = torch.cat([d_real_logits, d_fake_logits], dim=0)   # data movement only
  return Fn.GanLossFn.apply(all_logits, K.LOSS_KINDS[kind])


@gin.configurable(whitelist=[])
def non_saturating(d_real_logits, d_fake_logits, d_real=None, d_fake=None):
  """(d_loss, d_loss_real, d_loss_fake, g_loss) for the non-saturating loss (loss_lib.py:53-79)."""
  return _run("non_saturating", d_real_logits, d_fake_logits, d_real, d_fake)


@gin.configurable(whitelist=[])
def wasserstein(d_real_logits, d_fake_logits, d_real=None, d_fake=None):
  """Wasserstein loss (loss_lib.py:82-102)."""
  return _run("wasserstein", d_real_logits, d_fake_logits, d_real, d_fake)


@gin.configurable(whitelist=[])
def least_squares(d_real, d_fake, d_real_logits=None, d_fake_logits=None):
  """Least-squares loss on the sigmoid outputs (loss_lib.py:105-125); the kernel recomputes
  d_real/d_fake = sigmoid(logits)."""
  if d_real_logits is None or d_fake_logits is None:
    raise ValueError("least_squares needs the logits next to the probabilities.")
  return _run("least_squares", d_real_logits, d_fake_logits, d_real, d_fake)


@gin.configurable(whitelist=[])
def hinge(d_real_logits, d_fake_logits, d_real=None, d_fake=None):
  """Hinge loss (loss_lib.py:128-148)."""
  return _run("hinge", d_real_logits, d_fake_logits, d_real, d_fake)


@gin.configurable("loss", whitelist=["fn"])
def get_losses(fn=non_saturating, **kwargs):
  """Returns the losses for the discriminator and generator (loss_lib.py:151-154)."""
  return utils.call_with_accepted_args(fn, **kwargs)
