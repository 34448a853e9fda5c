"""Launcher of the train / eval schedules (reference: compare_gan/main.py:43-133, flags kept).

One process per MI355X.  Single GPU:
    python -m compare_gan_amd.main --model_dir /tmp/m --gin_config example_configs/x.gin
Data parallel over the GPUs of one node (RCCL over xGMI; the gin batch size is the GLOBAL batch,
runner_lib.py:84-85):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m compare_gan_amd.main --model_dir /tmp/m --gin_config example_configs/x.gin
The process reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, joins the
process group and switches batch norm to cross-replica statistics (tpu_ops.init_replicas).
"""
import argparse
import os

import torch


def parse_args(argv=None):
  p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawTextHelpFormatter)
  p.add_argument("--model_dir", required=True, help="Where to store files.")
  p.add_argument("--schedule", default="train",
                 help="Schedule to run. Options: train, eval_after_train, continuous_eval.")
  p.add_argument("--gin_config", action="append", default=[], help="Paths to the config files.")
  p.add_argument("--gin_bindings", action="append", default=[], help="Gin parameter bindings.")
  p.add_argument("--score_filename", default="scores.csv")
  p.add_argument("--num_eval_averaging_runs", type=int, default=3)
  p.add_argument("--eval_every_steps", type=int, default=5000)
  p.add_argument("--log_every", type=int, default=100)
  # the reference's data flags (datasets.py:46-63) and --use_tpu (main.py:66), so that its command
  # lines run unchanged
  p.add_argument("--tfds_data_dir", default=None,
                 help="TFDS data directory (record shards) or <dir>/<dataset>/{train,test}.npz; "
                      "default: CGAMD_DATA_DIR, else synthetic data.")
  p.add_argument("--data_fake_dataset", type=_flag_bool, nargs="?", const=True, default=None,
                 help="True: synthetic data even if a data directory is given (the default here "
                      "when no directory is set: there is no network to fetch datasets).")
  p.add_argument("--data_shuffle_buffer_size", type=int, default=10000)
  p.add_argument("--data_reading_num_threads", type=int, default=64, help="accepted, unused")
  p.add_argument("--use_tpu", type=_flag_bool, nargs="?", const=True, default=None,
                 help="accepted, ignored: the accelerator path (unrolled steps) is always taken")
  # absl's negative forms of the boolean flags (--nouse_tpu, --nodata_fake_dataset)
  p.add_argument("--nodata_fake_dataset", dest="data_fake_dataset", action="store_const", const=False,
                 help=argparse.SUPPRESS)
  p.add_argument("--nouse_tpu", dest="use_tpu", action="store_const", const=False,
                 help=argparse.SUPPRESS)
  return p.parse_args(argv)


def _flag_bool(text):
  """absl's boolean flag syntax: --flag, --flag=true / false / 1 / 0."""
  if text.lower() in ("1", "true", "t", "yes", "y"):
    return True
  if text.lower() in ("0", "false", "f", "no", "n"):
    return False
  raise argparse.ArgumentTypeError("not a boolean: %r" % text)


def configure_data(args):
  """--tfds_data_dir / --data_fake_dataset / --data_shuffle_buffer_size -> datasets.use_data_dir."""
  from compare_gan_amd import datasets
  path = args.tfds_data_dir or os.environ.get("CGAMD_DATA_DIR") or None
  if args.data_fake_dataset:
    path = None
  elif args.data_fake_dataset is False and path is None:
    # the reference would read the real dataset here; silently training on noise instead would be a
    # different experiment
    raise SystemExit("--data_fake_dataset=false needs a data directory (--tfds_data_dir or "
                     "CGAMD_DATA_DIR): there is no network to fetch datasets from")
  datasets.use_data_dir(path, shuffle_buffer_size=args.data_shuffle_buffer_size)
  return path


def main(argv=None):
  args = parse_args(argv)
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  torch.cuda.set_device(local_rank)
  device = torch.device("cuda", local_rank)
  from compare_gan_amd import gin
  from compare_gan_amd import runner_lib
  from compare_gan_amd.tpu import tpu_ops
  tpu_ops.init_replicas(device)
  gin.parse_config_files_and_bindings(args.gin_config, args.gin_bindings)
  configure_data(args)
  run_config = runner_lib.RunConfig(model_dir=args.model_dir)
  task_manager = runner_lib.TaskManagerWithCsvResults(
      model_dir=args.model_dir, score_file=os.path.join(args.model_dir, args.score_filename))
  options = runner_lib.get_options_dict()
  try:
    runner_lib.run_with_schedule(
        schedule=args.schedule, run_config=run_config, task_manager=task_manager, options=options,
        num_eval_averaging_runs=args.num_eval_averaging_runs,
        eval_every_steps=args.eval_every_steps, device=device, log_every=args.log_every)
  finally:
    import torch.distributed as dist
    if dist.is_initialized():
      dist.destroy_process_group()


if __name__ == "__main__":
  main()
