"""Evaluation of a trained GAN: IS + FID on Inception features.

Reference: compare_gan/eval_gan_lib.py:43-212 (evaluate_tfhub_module).  The TF-Hub module export /
import is replaced by evaluating the GAN object directly (its generator with is_training=False and
EMA weights when g_use_ema, modular_gan.py:266-285); everything else keeps the reference's
semantics: seeds fixed to 42, batch size 64, ceil(num_test_examples / 64) batches truncated to the
test-set size, optional BN accumulator fill with 204,800 samples, per-task mean / std / list over
`num_averaging_runs` fake sets."""
import numpy as np
import torch

from compare_gan_amd import eval_utils
from compare_gan_amd import gin
from compare_gan_amd import utils
from compare_gan_amd.gans.modular_gan import random_uniform
from compare_gan_amd.tpu import tpu_random

NAN_DETECTED = 31337.0
LAST_TIMING = {}   # wall-clock split of the most recent evaluate_gan call (seconds)


@gin.configurable("eval_z", blacklist=["shape", "name"])
def z_generator(shape, distribution_fn=random_uniform, minval=-1.0, maxval=1.0, stddev=1.0,
                name=None, device=None):
  """Random noise distributions for evaluation (eval_gan_lib.py:43-62)."""
  return utils.call_with_accepted_args(distribution_fn, shape=shape, minval=minval, maxval=maxval,
                                       stddev=stddev, name=name, device=device)


def _update_bn_accumulators(gan, generate_fn, batch_size, num_accu_examples):
  """Fills the accumulator statistics of batch norms configured with use_moving_averages=False
  (eval_gan_lib.py:65-92).  Returns True if there were accumulators."""
  switches = [v for n, v in gan.store.vars.items() if "accu/update_accus" in n]
  if not switches:
    return False
  gan.store.set_accu_fill(True)    # the variables and their host mirror (arch_ops.standardize_batch)
  try:
    for _ in range(num_accu_examples // batch_size):
      generate_fn()
  finally:
    gan.store.set_accu_fill(False)
  return True


def evaluate_gan(gan, eval_tasks, num_averaging_runs=1, num_accu_examples=204800,
                 num_test_examples=None):
  """Evaluates `gan` (built, weights loaded) with the given tasks -> {metric_mean/_std/_list}.

  Raises eval_utils.NanFoundError if the generator output has NaNs (eval_gan_lib.py:95-212)."""
  np.random.seed(42)
  dataset = gan._dataset  # pylint: disable=protected-access
  if num_test_examples is None:
    num_test_examples = dataset.eval_test_samples
  batch_size = 64
  num_batches = int(np.ceil(num_test_examples / batch_size))
  device = gan.device
  # the same latent variables for each evaluation: a dedicated counter-based stream, seed 42
  eval_step = torch.zeros((), dtype=torch.int64, device=device)
  saved = (tpu_random._st()["seed"], tpu_random._st()["step"])  # pylint: disable=protected-access
  tpu_random.set_random_offset(42, eval_step)
  counter = [0]

  def generate():
    counter[0] += 1
    z = z_generator(shape=[batch_size, gan._z_dim], name="eval_z/%d" % counter[0],  # pylint: disable=protected-access
                    device=device)
    labels = None
    if gan.conditional:
      labels = tpu_random.labels(batch_size, dataset.num_classes, "eval_labels/%d" % counter[0],
                                 device)
    return gan.generate(z, labels)

  import time
  timing = {"accumulators": 0.0, "sample": 0.0, "inception": 0.0, "stats": 0.0}

  def tick():
    torch.cuda.synchronize(device)
    return time.perf_counter()

  try:
    t0 = tick()
    _update_bn_accumulators(gan, generate, batch_size, num_accu_examples)
    timing["accumulators"] = tick() - t0
    if not eval_tasks:
      return None
    fake_dsets = []
    for i in range(num_averaging_runs):
      t0 = tick()
      fake_dset = eval_utils.EvalDataSample(eval_utils.sample_fake_dataset(generate, num_batches))
      t1 = tick()
      timing["sample"] += t1 - t0
      activations, logits = eval_utils.inception_transform_np(fake_dset.images, batch_size)
      timing["inception"] += tick() - t1
      fake_dset.set_inception_features(activations=activations, logits=logits)
      fake_dset.set_num_examples(num_test_examples)
      if i != 0:
        fake_dset.discard_images()
      fake_dsets.append(fake_dset)
  finally:
    tpu_random.set_random_offset(*saved)

  t0 = tick()
  real_dset = eval_utils.EvalDataSample(
      eval_utils.get_real_images(dataset=dataset, num_examples=num_test_examples, device=device))
  real_dset.activations, _ = eval_utils.inception_transform_np(real_dset.images, batch_size)
  real_dset.set_num_examples(num_test_examples)
  t1 = tick()
  timing["inception"] += t1 - t0

  result_dict = {}
  for task in eval_tasks:
    task_results_dicts = [task.run_after_session(fake_dset, real_dset) for fake_dset in fake_dsets]
    result_statistics = {}
    for key in task_results_dicts[0].keys():
      scores_for_key = np.array([d[key] for d in task_results_dicts])
      result_statistics[key + "_mean"] = np.mean(scores_for_key)
      result_statistics[key + "_std"] = np.std(scores_for_key)
      result_statistics[key + "_list"] = "_".join([str(x) for x in scores_for_key])
    result_dict.update(result_statistics)
  # trained Inception weights cannot be fetched offline: results computed on the seeded synthetic
  # extractor are tagged so they are never mistaken for numbers comparable with published FIDs
  result_dict["inception_weights_synthetic"] = float(
      eval_utils.inception_weights_are_synthetic(device))
  timing["stats"] = tick() - t1
  LAST_TIMING.clear()
  LAST_TIMING.update(timing)
  return result_dict
