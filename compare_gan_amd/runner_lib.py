"""Options, schedules and the train / eval loop (reference: compare_gan/runner_lib.py:58-354).

Kept surface: gin "options" (runner_lib.py:72-111), TaskManager with its TRAIN_DONE marker,
checkpoint discovery and scores.csv (:114-232), _run_eval (:235-277) and run_with_schedule
(:280-354) with the schedules train / eval_after_train / continuous_eval.  TPUEstimator is replaced
by the GAN object's own train_step(); checkpoints are torch files named `model.ckpt-<step>.pt`
holding the reference's variable names.
"""
import csv
import os
import time

import numpy as np
import torch

from compare_gan_amd import datasets
from compare_gan_amd import gin


@gin.configurable("options")
def get_options_dict(batch_size=gin.REQUIRED, gan_class=gin.REQUIRED,
                     architecture=gin.REQUIRED, training_steps=gin.REQUIRED,
                     discriminator_normalization=None, lamba=1, disc_iters=1, z_dim=128):
  """Parse legacy options from Gin configurations into a Python dict (runner_lib.py:72-111)."""
  del discriminator_normalization
  return {
      "use_tpu": False,
      "batch_size": batch_size,
      "gan_class": gan_class,
      "architecture": architecture,
      "training_steps": training_steps,
      "lambda": lamba,  # Different spelling intended.
      "disc_iters": disc_iters,
      "z_dim": z_dim,
  }


@gin.configurable("run_config")
class RunConfig(object):
  """The fields of main.py:79-95's RunConfig that matter off-TPU."""

  def __init__(self, model_dir=None, tf_random_seed=None, single_core=False,
               iterations_per_loop=1000, save_checkpoints_steps=5000, keep_checkpoint_max=1000):
    self.model_dir = model_dir
    self.tf_random_seed = tf_random_seed
    self.single_core = single_core
    self.iterations_per_loop = iterations_per_loop
    self.save_checkpoints_steps = save_checkpoints_steps
    self.keep_checkpoint_max = keep_checkpoint_max


class TaskManager(object):
  """Interface for managing a task (runner_lib.py:114-232)."""

  _TASK_IS_DONE_MARKER = "TRAIN_DONE"
  _RESULTS_FILE = "scores.csv"

  def __init__(self, model_dir):
    self._model_dir = model_dir
    os.makedirs(model_dir, exist_ok=True)

  @property
  def model_dir(self):
    return self._model_dir

  def mark_training_done(self):
    with open(os.path.join(self.model_dir, self._TASK_IS_DONE_MARKER), "w") as f:
      f.write("")

  def is_training_done(self):
    return os.path.exists(os.path.join(self.model_dir, self._TASK_IS_DONE_MARKER))

  def add_eval_result(self, checkpoint_path, result_dict, default_value):
    """Appends one row to scores.csv: the metrics + every operative gin binding."""
    step = _step_of(checkpoint_path)
    row = {"checkpoint_path": checkpoint_path, "step": step}
    row.update(_parse_gin_config(os.path.join(
        self.model_dir, "operative_config-{}.gin".format(step))))
    row.update(result_dict)
    path = self._score_path()
    rows = []
    if os.path.exists(path):
      with open(path) as f:
        rows = list(csv.DictReader(f))
    rows.append({k: str(v) for k, v in row.items()})
    keys = sorted(set(k for r in rows for k in r))
    with open(path, "w") as f:
      w = csv.DictWriter(f, fieldnames=keys, restval=default_value)
      w.writeheader()
      w.writerows(rows)

  def _score_path(self):
    return os.path.join(self.model_dir, self._RESULTS_FILE)

  def get_checkpoints_with_results(self):
    path = self._score_path()
    if not os.path.exists(path):
      return set()
    with open(path) as f:
      return set(r["checkpoint_path"] for r in csv.DictReader(f))

  def _checkpoints(self):
    """Every checkpoint of model_dir, ascending by step: this package's `model.ckpt-<step>.pt` and
    the reference's TF-1 bundles `model.ckpt-<step>` (`.index` + `.data-*`), the native file first
    when both exist for a step."""
    found = {}
    for p in os.listdir(self.model_dir):
      if not p.startswith("model.ckpt-"):
        continue
      try:   # (a stray `model.ckpt-best.pt` / `model.ckpt-7_temp.index` is not a checkpoint of a step)
        if p.endswith(".pt"):
          found[_step_of(p)] = os.path.join(self.model_dir, p)
        elif p.endswith(".index"):
          found.setdefault(_step_of(p[:-len(".index")]),
                           os.path.join(self.model_dir, p[:-len(".index")]))
      except ValueError:
        continue
    return [found[k] for k in sorted(found)]

  WAIT = "<wait>"   # yielded instead of sleeping (yield_waits): nothing new yet, poll again later

  def unevaluated_checkpoints(self, timeout=0, eval_every_steps=None, poll_seconds=60,
                              yield_waits=False):
    """Generator for checkpoints without evaluation results (runner_lib.py:137-180): ascending by
    step; with eval_every_steps only steps > 0 divisible by it; with timeout > 0 it keeps polling
    the directory (every poll_seconds) until no new checkpoint has appeared for `timeout` seconds or
    training is marked done -- the continuous-evaluation schedule.  yield_waits: between polls the
    generator yields TaskManager.WAIT instead of sleeping itself, so that a multi-rank caller can
    keep every collective short (the other ranks must not sit in a broadcast for hours)."""
    evaluated = set(self.get_checkpoints_with_results())
    last_eval = time.time()
    while True:
      todo = [p for p in self._checkpoints() if p not in evaluated]
      if eval_every_steps:
        todo = [p for p in todo if _step_of(p) > 0 and _step_of(p) % eval_every_steps == 0]
      for path in todo:
        yield path
      if todo:
        evaluated |= set(todo)
        last_eval = time.time()
        continue
      if time.time() - last_eval > timeout or self.is_training_done():
        break
      if yield_waits:
        yield self.WAIT
      else:
        time.sleep(poll_seconds)


class TaskManagerWithCsvResults(TaskManager):
  """Task manager that writes results to a CSV file of the caller's choice
  (runner_lib.py:186-232)."""

  def __init__(self, model_dir, score_file=None):
    super(TaskManagerWithCsvResults, self).__init__(model_dir)
    self._score_file = score_file or os.path.join(model_dir, "scores.csv")

  def _score_path(self):
    return self._score_file


_CONTINUOUS_EVAL_TIMEOUT_S = 24 * 3600    # runner_lib.py:341-343
_CONTINUOUS_EVAL_POLL_S = 60              # runner_lib.py:179


def _step_of(checkpoint_path):
  return int(os.path.basename(checkpoint_path).split("-")[-1].split(".")[0])


def _parse_gin_config(config_path):
  """Parses a Gin config into a dictionary; all values are strings (runner_lib.py:58-69)."""
  config = {}
  if not os.path.exists(config_path):
    return config
  with open(config_path) as f:
    for line in f:
      line = line.split("#", 1)[0].strip()
      if "=" in line:
        k, v = line.split("=", 1)
        config[k.strip()] = v.strip()
  return config


def save_checkpoint(gan, model_dir, step):
  path = os.path.join(model_dir, "model.ckpt-{}.pt".format(step))
  # every file appears under its final name complete (temp file + os.replace): a peer process or a
  # continuous-eval job polling the directory never reads a half-written checkpoint or an empty marker
  torch.save({k: v.cpu() for k, v in gan.state_dict().items()}, path + ".tmp")
  os.replace(path + ".tmp", path)
  with open(os.path.join(model_dir, "operative_config-{}.gin".format(step)), "w") as f:
    f.write(gin.operative_config_str())
  marker = os.path.join(model_dir, "checkpoint")
  with open(marker + ".tmp", "w") as f:
    f.write(os.path.basename(path) + "\n")
  os.replace(marker + ".tmp", marker)
  return path


def latest_checkpoint(model_dir):
  marker = os.path.join(model_dir, "checkpoint")
  if not os.path.exists(marker):
    return None
  with open(marker) as f:
    name = f.read().strip()
  if not name:
    return None
  path = os.path.join(model_dir, name)
  return path if os.path.isfile(path) else None


def _latest_tf_checkpoint(model_dir):
  if not model_dir or not os.path.isdir(model_dir):
    return None
  from compare_gan_amd import tf_checkpoint
  return tf_checkpoint.latest_checkpoint(model_dir)


def _barrier():
  import torch.distributed as dist
  if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
    dist.barrier()


def _broadcast_from_rank0(value):
  """`value` of rank 0 on every rank (a small picklable object)."""
  import torch.distributed as dist
  if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
    return value
  box = [value]
  dist.broadcast_object_list(box, src=0)
  return box[0]


def _run_eval(gan, checkpoint_path, task_manager, options, num_averaging_runs, device):
  """Evaluates one checkpoint with IS + FID (runner_lib.py:235-277)."""
  from compare_gan_amd import eval_gan_lib
  from compare_gan_amd import eval_utils
  from compare_gan_amd.metrics import fid_score as fid_score_lib
  from compare_gan_amd.metrics import inception_score as inception_score_lib
  del options
  eval_tasks = [inception_score_lib.InceptionScoreTask(), fid_score_lib.FIDScoreTask()]
  if checkpoint_path.endswith(".pt"):
    gan.load_state_dict(torch.load(checkpoint_path, map_location=device))
  else:     # a TF-1 bundle prefix written by the reference (modular_gan.py:266-285 restores those)
    from compare_gan_amd import tf_checkpoint
    tf_checkpoint.import_tf_checkpoint(gan, checkpoint_path)
  from compare_gan_amd.tpu import tpu_ops
  in_context = tpu_ops._STATE["enabled"]  # pylint: disable=protected-access
  # no collectives INSIDE the networks (each rank runs different evaluation batches on the same
  # weights); the ranks share the batches and exchange features (eval_shard.py), every rank gets
  # the result and rank 0 records it
  tpu_ops.enable_cross_replica(False)
  try:
    result_dict = eval_gan_lib.evaluate_gan(gan, eval_tasks, num_averaging_runs)
  except eval_utils.NanFoundError as nan_found_error:   # raised on all ranks together
    result_dict = {}
    print("NanFoundError:", nan_found_error)
  finally:
    tpu_ops.enable_cross_replica(in_context)
  default_value = eval_gan_lib.NAN_DETECTED
  if tpu_ops.replica_id() == 0:
    task_manager.add_eval_result(checkpoint_path, result_dict, default_value)
  return result_dict


def run_with_schedule(schedule, run_config, task_manager, options, use_tpu=False,
                      num_eval_averaging_runs=1, eval_every_steps=-1, device="cuda:0",
                      log_every=100):
  """Run the schedule with the given options (runner_lib.py:280-354).

  `training_steps` counts GENERATOR steps; checkpoints are written every
  save_checkpoints_steps generator steps and training resumes from the latest one."""
  del use_tpu
  if schedule not in {"train", "eval_after_train", "continuous_eval"}:
    raise ValueError("Schedule {} not supported.".format(schedule))
  seed = run_config.tf_random_seed if run_config.tf_random_seed is not None else 0
  np.random.seed(seed)                                            # runner_lib.py:303-305
  dataset = datasets.get_dataset()
  gan = options["gan_class"](dataset=dataset, parameters=options,
                             model_dir=run_config.model_dir)
  from compare_gan_amd.tpu import tpu_ops
  tpu_ops.init_replicas(device)      # no-op unless launched with WORLD_SIZE > 1
  world = tpu_ops.num_replicas()
  if options["batch_size"] % world:
    raise ValueError("batch_size {} is not divisible by {} replicas".format(
        options["batch_size"], world))
  bsz = options["batch_size"] // world              # runner_lib.py:84-85
  gan.build(batch_size=bsz, device=device, seed=seed)
  if schedule in {"train", "eval_after_train"}:
    start = 0
    # rank 0 decides where to resume and tells the others: ranks looking at the directory on their own
    # could disagree while rank 0 writes checkpoint 0
    ckpt = _broadcast_from_rank0(
        latest_checkpoint(run_config.model_dir) if tpu_ops.replica_id() == 0 else None)
    if ckpt is not None:                              # README.md:93-94 resume
      gan.load_state_dict(torch.load(ckpt, map_location=device))
      start = int(gan.global_step.item())
    else:
      # a model_dir written by the REFERENCE (TF-1 tensor bundles `model.ckpt-<step>.index` +
      # `.data-*`, same variable names: SURVEY App. D): continue from its newest checkpoint
      tf_ckpt = _broadcast_from_rank0(_latest_tf_checkpoint(run_config.model_dir)
                                      if tpu_ops.replica_id() == 0 else None)
      if tf_ckpt is not None:
        from compare_gan_amd import tf_checkpoint
        tf_checkpoint.import_tf_checkpoint(gan, tf_ckpt)
        start = int(gan.global_step.item())
    num_sub = options.get("disc_iters", 1) + 1
    batches = dataset.train_batches(bsz * num_sub, seed=dataset._seed + tpu_ops.replica_id())  # pylint: disable=protected-access
    if start == 0 and tpu_ops.replica_id() == 0:
      save_checkpoint(gan, run_config.model_dir, 0)
    _barrier()
    t0, last = time.time(), start
    for step in range(start, options["training_steps"]):
      images, labels = next(batches)
      out = gan.train_step(torch.from_numpy(images).to(gan.device),
                           torch.from_numpy(labels).to(gan.device))
      done = step + 1
      if log_every and done % log_every == 0:
        torch.cuda.synchronize()
        dt = time.time() - t0
        print("%.1f%% @%d, %.2f steps/s, %.1f img/s, d_loss %.4f g_loss %.4f" % (
            100.0 * done / options["training_steps"], done, (done - last) / dt,
            (done - last) * bsz * num_sub * world / dt, float(out["d_losses"][0]),
            float(out["g_loss"])))
        t0, last = time.time(), done
      if tpu_ops.replica_id() == 0 and (done % run_config.save_checkpoints_steps == 0 or
                                        done == options["training_steps"]):
        save_checkpoint(gan, run_config.model_dir, done)
    if tpu_ops.replica_id() == 0:
      task_manager.mark_training_done()
  if schedule in {"eval_after_train", "continuous_eval"}:
    _barrier()  # the last checkpoint and the training-done marker are on disk
    # rank 0 looks at the directory (continuous_eval: keeps polling it, up to 24 hours between
    # checkpoints, runner_lib.py:340-343) and hands every rank the next checkpoint; every rank
    # evaluates its share of that checkpoint's batches (eval_shard.py)
    every = eval_every_steps if eval_every_steps and eval_every_steps > 0 else None
    pending = None
    if tpu_ops.replica_id() == 0:
      pending = iter(task_manager.unevaluated_checkpoints(
          timeout=_CONTINUOUS_EVAL_TIMEOUT_S if schedule == "continuous_eval" else 0,
          eval_every_steps=every, poll_seconds=_CONTINUOUS_EVAL_POLL_S, yield_waits=True))
    while True:
      # every broadcast is short: while rank 0 has nothing new it hands out a WAIT token and ALL ranks
      # sleep one poll interval outside any collective (a rank parked in a broadcast for longer than
      # the process group's timeout -- 10 minutes on RCCL -- would be killed by its watchdog)
      checkpoint_path = _broadcast_from_rank0(next(pending, None) if pending is not None else None)
      if checkpoint_path is None:
        break
      if checkpoint_path == TaskManager.WAIT:
        time.sleep(_CONTINUOUS_EVAL_POLL_S)
        continue
      _run_eval(gan, checkpoint_path, task_manager, options, num_eval_averaging_runs, device)
  _barrier()    # nobody tears the process group down while another rank still works
  return gan
