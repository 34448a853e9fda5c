"""Deterministic stateless random numbers for ops inside the training step.

Reference: compare_gan/tpu/tpu_random.py:54-154 -- every random op gets a stable id derived from
its name and is re-keyed per training step and per replica, so a step's randomness is reproducible
and differs across steps and cores (tpu_random_test.py:87-168).  Here: Philox4x32-10 in a HIP
kernel keyed by (seed, op id), counter = (element, replica, device-resident step), which keeps the
whole step hipGraph-capturable (the step is read on the device).
"""
import hashlib

from compare_gan_amd.hip import kernels as K
from compare_gan_amd.tpu import tpu_ops

_STATE = {"seed": 0, "step": None, "sub_step": 0}


def _st():
  """The module state; an in-process replica thread (tpu_ops.InProcessReplicas) has its own."""
  ts = tpu_ops.thread_state()
  if ts is None:
    return _STATE
  return ts.setdefault("tpu_random", {"seed": 0, "step": None, "sub_step": 0})


def _op_id(name):
  """tpu_random.py:81-86: sha512(name) mod (2^31 - 1)."""
  return int(hashlib.sha512(name.encode("utf-8")).hexdigest(), 16) % (2 ** 31 - 1)


def set_random_offset(seed, step_tensor):
  """The analogue of set_random_offset_from_features (tpu_random.py:54-78): binds the step counter
  (device int64 tensor) and the run seed used by subsequent calls."""
  _st()["seed"] = int(seed)
  _st()["step"] = step_tensor
  _st()["sub_step"] = 0     # a (re)bound stream starts at the first sub-step


def set_sub_step(index):
  """Index of the sub-step being built inside one unrolled training step: random ops without a
  per-sub-step name of their own (the penalties' draws) append it, so that every sub-step has its
  own op id -- the reference's unrolled graph holds one random op per sub-step
  (modular_gan.py:568-584 + tpu_random.py:81-86)."""
  _st()["sub_step"] = int(index)


def sub_step():
  return _st()["sub_step"]


def uniform(shape, name, minval=0.0, maxval=1.0, device=None):
  st = _st()
  step = st["step"]
  return K.random(0, minval, maxval, st["seed"], _op_id(name), tpu_ops.random_stream_id(), step,
                  tuple(shape), device if device is not None else step.device)


def normal(shape, name, mean=0.0, stddev=1.0, device=None):
  st = _st()
  step = st["step"]
  return K.random(1, mean, stddev, st["seed"], _op_id(name), tpu_ops.random_stream_id(), step,
                  tuple(shape), device if device is not None else step.device)


def labels(n, num_classes, name, device=None):
  st = _st()
  step = st["step"]
  return K.random_labels(num_classes, st["seed"], _op_id(name), tpu_ops.random_stream_id(), step, n,
                         device if device is not None else step.device)
