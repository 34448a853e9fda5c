"""Placeholder until the eval path lands (see compare_gan_amd/eval_gan_lib.py in later commits)."""
from compare_gan_amd import gin
from compare_gan_amd import utils
from compare_gan_amd.gans.modular_gan import random_uniform

NAN_DETECTED = 31337.0


@gin.configurable("eval_z", blacklist=["shape", "name"])
def z_generator(shape, distribution_fn=random_uniform, minval=-1.0, maxval=1.0, stddev=1.0,
                name=None, device=None):
  """Random noise distributions for evaluation (eval_gan_lib.py:43-62)."""
  return utils.call_with_accepted_args(distribution_fn, shape=shape, minval=minval, maxval=maxval,
                                       stddev=stddev, name=name, device=device)
