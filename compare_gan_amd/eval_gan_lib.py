"""Evaluation of a trained GAN: IS + FID on Inception features.

Reference: compare_gan/eval_gan_lib.py:43-212 (evaluate_tfhub_module).  The TF-Hub module export /
import is replaced by evaluating the GAN object directly (its generator with is_training=False and
EMA weights when g_use_ema, modular_gan.py:266-285); everything else keeps the reference's
semantics: seeds fixed to 42, batch size 64, ceil(num_test_examples / 64) batches truncated to the
test-set size, optional BN accumulator fill with 204,800 samples, per-task mean / std / list over
`num_averaging_runs` fake sets."""
import numpy as np
import torch

from compare_gan_amd import eval_shard
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


def _update_bn_accumulators(gan, generate_fn, batch_size, num_accu_examples, first_index=1,
                            rank=0, world=1):
  """Fills the accumulator statistics of batch norms configured with use_moving_averages=False
  (eval_gan_lib.py:65-92) with batches first_index .. first_index + n - 1; with world > 1 the
  batches are dealt round-robin and the per-rank sums all-reduced (eval_shard.allreduce_deltas).
  Returns the number of batches consumed (0 if there are no accumulators)."""
  switches = [v for n, v in gan.store.vars.items() if "accu/update_accus" in n]
  if not switches:
    return 0
  num = num_accu_examples // batch_size
  accus = [v for n, v in sorted(gan.store.vars.items())
           if "accu/accu_mean" in n or "accu/accu_variance" in n or "accu/accu_counter" in n]
  before = [v.detach().clone() for v in accus] if world > 1 else None
  gan.store.set_accu_fill(True)    # the variables and their host mirror (arch_ops.standardize_batch)
  try:
    for i in eval_shard.shard_indices(num, rank, world):
      generate_fn(first_index + i)
  finally:
    gan.store.set_accu_fill(False)
  eval_shard.allreduce_deltas(accus, before, world)
  return num


def evaluate_gan(gan, eval_tasks, num_averaging_runs=1, num_accu_examples=204800,
                 num_test_examples=None, shard=None):
  """Evaluates `gan` (built, weights loaded) with the given tasks -> {metric_mean/_std/_list}.

  Raises eval_utils.NanFoundError if the generator output has NaNs (eval_gan_lib.py:95-212).
  With an initialised process group of W > 1 ranks (all holding the same weights; every rank must
  call this) the generator / Inception batches are sharded over the ranks and the features
  all-gathered (compare_gan_amd/eval_shard.py); every rank returns the same result.  shard=False
  evaluates on the calling rank alone."""
  np.random.seed(42)
  dataset = gan._dataset  # pylint: disable=protected-access
  if num_test_examples is None:
    num_test_examples = dataset.eval_test_samples
  batch_size = 64
  num_batches = int(np.ceil(num_test_examples / batch_size))
  device = gan.device
  rank, world = eval_shard.rank_world(shard)
  if num_batches < world:
    rank, world = 0, 1          # fewer batches than ranks: not worth a collective
  # the same latent variables for each evaluation: a dedicated counter-based stream, seed 42; batch
  # `index` is a pure function of its index (the name keys the stream), which is what lets ranks
  # take batches independently
  eval_step = torch.zeros((), dtype=torch.int64, device=device)
  saved = (tpu_random._st()["seed"], tpu_random._st()["step"])  # pylint: disable=protected-access
  tpu_random.set_random_offset(42, eval_step)
  next_index = 1

  def draw(index):
    z = z_generator(shape=[batch_size, gan._z_dim], name="eval_z/%d" % index,  # pylint: disable=protected-access
                    device=device)
    labels = None
    if gan.conditional:
      labels = tpu_random.labels(batch_size, dataset.num_classes, "eval_labels/%d" % index, device)
    return z, labels

  def generate_eager(index):
    return gan.generate(*draw(index))

  # the sampling loop proper: one hipGraph replay of the generator per batch (its result is a static
  # buffer that the sink consumes before the next replay); the accumulator fill below changes
  # host-side state per call and stays on generate()
  sampler = [None]

  def generate(index):
    if sampler[0] is None:
      sampler[0] = gan.make_sampler(batch_size)
    return sampler[0](*draw(index))

  import time
  timing = {"accumulators": 0.0, "sample": 0.0, "sample_first_batch": 0.0, "inception": 0.0,
            "gather": 0.0, "stats": 0.0, "ranks": world}

  def tick():
    torch.cuda.synchronize(device)
    return time.perf_counter()

  transform = lambda images: eval_utils.inception_transform_np(images, batch_size)
  try:
    t0 = tick()
    next_index += _update_bn_accumulators(gan, generate_eager, batch_size, num_accu_examples,
                                          next_index, rank, world)
    timing["accumulators"] = tick() - t0
    if not eval_tasks:
      return None
    fake_dsets = []
    for i in range(num_averaging_runs):
      images, activations, logits, nan_found = eval_shard.sharded_fake_features(
          generate, transform, num_batches, next_index, rank, world,
          keep_images=(world == 1 and i == 0), timing=timing, tick=tick,
          sink_cls=eval_utils.FakeImageSink)
      next_index += num_batches
      if nan_found:
        raise eval_utils.NanFoundError("Detected NaN in fake images.")
      fake_dset = eval_utils.EvalDataSample(images)
      fake_dset.set_inception_features(activations=activations, logits=logits)
      fake_dset.set_num_examples(num_test_examples)
      fake_dsets.append(fake_dset)
  finally:
    tpu_random.set_random_offset(*saved)

  t0 = tick()
  chunk = eval_shard.row_chunk(num_test_examples, world, batch_size)
  lo, hi = eval_shard.row_range(num_test_examples, rank, world, chunk)
  if hi > lo:
    real_dset = eval_utils.EvalDataSample(
        eval_utils.get_real_images(dataset=dataset, num_examples=num_test_examples, device=device,
                                   rows=(lo, hi)))
    local_act, _ = eval_utils.inception_transform_np(real_dset.images, batch_size)
  else:
    # the chunks are whole Inception batches, so a small real set leaves the last ranks without rows
    # (N = 257, 4 ranks: chunk 128, rank 3 empty): they contribute an empty block to the gather
    # instead of running the extractor on nothing while the others wait in the collective
    real_dset = eval_utils.EvalDataSample(None)
    ref = fake_dsets[0].activations
    local_act = ref.new_zeros((0,) + tuple(ref.shape[1:]))
  if world > 1:
    real_dset.discard_images()    # only this rank's rows: nothing downstream may mistake them for the set
  real_dset.activations = eval_shard.gather_rows(local_act, num_test_examples, rank, world, chunk)
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
