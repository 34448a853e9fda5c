"""Cross-replica collectives of the data-parallel path.

Reference: compare_gan/tpu/tpu_ops.py:29-125 (cross_replica_concat / cross_replica_mean /
cross_replica_moments on the TPU's `cross_replica_sum`).  Here one process drives one MI355X and
the collective is an RCCL all-reduce over xGMI through torch.distributed (backend "nccl" on the
GPU, "gloo" in the CPU tests); the arithmetic around it (mean^2, scaling) stays in HIP kernels.
"""
import os
import threading

import torch
import torch.distributed as dist

from compare_gan_amd import gin

_STATE = {"enabled": False, "groups": {}, "log": None}
_LOCAL = threading.local()   # in-process replica sets (InProcessReplicas) are per thread


def record_collectives(log):
  """Diagnostic: from now on every collective this process issues appends
  (kind, numel, dtype, group_size, issuing thread: "main" | "worker") to `log` (a list; None stops
  the recording); returns the previous log.  RCCL needs every rank to issue the collectives of one
  communicator in the same order, and two threads issue them here (the main thread: batch-norm
  forwards; autograd's device thread: batch-norm backwards and the gradient buckets -- see
  modular_gan._OptimizerState.arm): the tests compare the recorded sequences of the ranks
  (tests/test_data_parallel_gloo.py, tests/test_data_parallel_gpu.py)."""
  prev = _STATE["log"]
  _STATE["log"] = log
  return prev


def _note(kind, tensor, group_size=None):
  log = _STATE["log"]
  if log is not None:
    log.append((kind, int(tensor.numel()), str(tensor.dtype).replace("torch.", ""),
                int(group_size or 0),
                "main" if threading.current_thread() is threading.main_thread() else "worker"))


class InProcessReplicas(object):
  """A replica set inside ONE process: `world` threads, each driving one replica of the model on
  the same device, with the all-reduce done in memory.  It exists so that the product's
  data-parallel path (bucket, cross-replica sum, 1/world scaling, per-replica random streams) can
  run on a single-GPU box, where two RCCL ranks cannot share the device.

  The threads never run concurrently: a baton passes from replica to replica at every collective
  (replica r runs up to its k-th collective, hands over to r+1, ...; the last to arrive computes
  the sum in rank order, so every replica receives bit-identical values) -- module-level state
  that is set and consumed between two collectives therefore cannot race, and the state that
  lives across collectives (variable store in use, random-number step binding) is kept per
  thread (thread_state()).

  Usage, per thread:  replicas.attach(rank); build + step the model; replicas.finish(rank).
  Only collectives issued from the attached thread are supported: a cross-replica batch norm's
  backward all-reduce runs on autograd's per-device thread, which knows no replica."""

  def __init__(self, world):
    self.world = int(world)
    self._cv = threading.Condition()
    self._turn = 0
    self._done = [False] * self.world
    self._slots = [None] * self.world
    self._arrived = 0
    self._sum = None
    self._error = None

  # -- baton ------------------------------------------------------------------------------------
  def _wait_turn(self, rank):
    while self._turn != rank and self._error is None:
      self._cv.wait(timeout=1.0)
    if self._error is not None:
      raise RuntimeError("another in-process replica failed: %s" % (self._error,))

  def _pass_on(self, rank):
    for k in range(1, self.world + 1):
      nxt = (rank + k) % self.world
      if not self._done[nxt]:
        self._turn = nxt
        break
    self._cv.notify_all()

  def attach(self, rank, random_stream=None):
    """Binds the calling thread to replica `rank` and waits for its first turn.  random_stream:
    the replica id its random ops are keyed with (default: rank)."""
    _LOCAL.replicas = (self, int(rank))
    _LOCAL.state = {"random_stream": int(rank if random_stream is None else random_stream)}
    with self._cv:
      self._wait_turn(rank)

  def finish(self, rank, error=None):
    """The replica's thread is done (or failed): hands the baton on for good."""
    if torch.cuda.is_available():
      torch.cuda.synchronize()
    with self._cv:
      if error is not None and self._error is None:
        self._error = error
      self._done[rank] = True
      self._pass_on(rank)
    _LOCAL.replicas = None
    _LOCAL.state = None

  def all_reduce_sum_(self, rank, tensor):
    if tensor.is_cuda:
      torch.cuda.synchronize(tensor.device)     # the peers read it from their own streams
    with self._cv:
      self._slots[rank] = tensor
      self._arrived += 1
      if self._arrived == self.world:
        total = self._slots[0].clone()
        for t in self._slots[1:]:
          total += t          # rank order
        self._sum = total
        self._arrived = 0
      else:
        self._pass_on(rank)
        self._wait_turn(rank)
      tensor.copy_(self._sum)
      if tensor.is_cuda:
        torch.cuda.synchronize(tensor.device)
    return tensor


def thread_state():
  """Per-thread dict of an attached in-process replica (None otherwise): modules keep the state
  that lives across collectives there (see InProcessReplicas)."""
  return getattr(_LOCAL, "state", None)


def random_stream_id():
  """The replica coordinate of tpu_random's counters."""
  st = thread_state()
  return st["random_stream"] if st is not None else replica_id()


def _in_process():
  return getattr(_LOCAL, "replicas", None)


def num_replicas():
  local = _in_process()
  if local is not None:
    return local[0].world
  return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def replica_id():
  local = _in_process()
  if local is not None:
    return local[1]
  return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def force_data_parallel():
  """Debug switch (CGAMD_FORCE_DP=1): run the bucket + all-reduce path even on one replica, so the
  collective path can be exercised (and captured) on a single-GPU box."""
  return os.environ.get("CGAMD_FORCE_DP", "") == "1" and dist.is_available() and dist.is_initialized()


def data_parallel():
  """True when gradients / batch statistics have to cross replicas."""
  return num_replicas() > 1 or force_data_parallel()


def enable_cross_replica(enabled=True):
  """The analogue of 'running inside a TPU context' (arch_ops.py:258-263)."""
  _STATE["enabled"] = bool(enabled)


def in_replica_context():
  return _STATE["enabled"] and data_parallel()


def init_replicas(device=None, backend=None):
  """What a launcher calls once per process (one process per GPU, started by
  `python -m torch.distributed.run`): joins the process group named by RANK / WORLD_SIZE /
  MASTER_ADDR / MASTER_PORT -- backend "nccl" (= RCCL over xGMI) for a GPU device, "gloo" for the
  CPU tests -- and switches batch norm to cross-replica statistics, the reference's behaviour
  whenever it runs data parallel (arch_ops.py:258-263: use_cross_replica_mean defaults to "in a
  TPU context").  Returns (replica_id, num_replicas).  With WORLD_SIZE unset or 1 it does nothing
  unless CGAMD_FORCE_DP=1 asks for a one-rank group."""
  world = int(os.environ.get("WORLD_SIZE", "1"))
  force = os.environ.get("CGAMD_FORCE_DP", "") == "1"
  if world > 1 or force:
    if not dist.is_initialized():
      os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
      os.environ.setdefault("MASTER_PORT", "29517")
      os.environ.setdefault("RANK", "0")
      os.environ.setdefault("WORLD_SIZE", "1")
      backend = backend or os.environ.get("CGAMD_DIST_BACKEND") or None
      if backend is None:
        backend = "nccl" if device is not None and torch.device(device).type == "cuda" else "gloo"
      kwargs = {}
      if backend == "nccl":
        kwargs["device_id"] = torch.device(device)
      dist.init_process_group(backend=backend, **kwargs)
    enable_cross_replica(True)
  return replica_id(), num_replicas()


def _group(group_size):
  """Consecutive replica groups of `group_size` (tpu_ops.py:75-91 group_assignment)."""
  n = num_replicas()
  if group_size is None or group_size == 0 or group_size >= n:
    return None, n
  if n % group_size:
    raise ValueError("num_replicas {} is not divisible by group_size {}".format(n, group_size))
  key = int(group_size)
  if key not in _STATE["groups"]:
    mine = None
    for start in range(0, n, group_size):
      ranks = list(range(start, start + group_size))
      g = dist.new_group(ranks)  # every rank must create every group
      if replica_id() in ranks:
        mine = g
    _STATE["groups"][key] = mine
  return _STATE["groups"][key], group_size


def cross_replica_sum_(tensor, group_size=None):
  """In-place all-reduce SUM (the one primitive the reference builds everything from)."""
  if not data_parallel() or group_size == 1:
    return tensor, 1
  _note("all_reduce_sum", tensor, group_size)
  local = _in_process()
  if local is not None:
    if group_size not in (None, 0) and group_size < local[0].world:
      raise ValueError("in-process replicas do not form sub-groups")
    local[0].all_reduce_sum_(local[1], tensor)
    return tensor, local[0].world
  group, n = _group(group_size)
  if tensor.is_cuda and dist.get_backend(group) == "gloo":
    # debugging backend (CGAMD_DIST_BACKEND=gloo): replicas that cannot form an RCCL group --
    # e.g. two processes sharing one GPU in tests/test_data_parallel_gpu.py -- sum through the host
    host = tensor.detach().cpu()
    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
    tensor.copy_(host)
    return tensor, n
  dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
  return tensor, n


def cross_replica_mean(inputs, group_size=None):
  """Mean over replicas (tpu_ops.py:75-91).  group_size=1 returns the input unchanged."""
  if not data_parallel() or group_size == 1:
    return inputs
  out = inputs.clone()
  _, n = cross_replica_sum_(out, group_size)
  if out.is_cuda and out.dtype == torch.float32:
    from compare_gan_amd.hip import kernels as K
    return K.axpby_f32(out.contiguous(), 1.0 / n).reshape(inputs.shape)
  return out / n   # CPU (gloo) tests of the host logic only


def cross_replica_concat(value, replica_id_, num_replicas_):
  """All-gather along axis 0 (tpu_ops.py:29-72; unused by the hot path, kept for parity)."""
  del replica_id_
  if num_replicas() == 1:
    return value
  parts = [torch.empty_like(value) for _ in range(num_replicas_)]
  _note("all_gather", value)
  dist.all_gather(parts, value.contiguous())
  return torch.cat(parts, dim=0)


@gin.configurable(blacklist=["inputs", "axis"])
def cross_replica_moments(inputs, axis, parallel=True, group_size=None):
  """Mean and variance over the global batch (tpu_ops.py:94-125), host-level reference form.

  parallel=True: var = E[x^2] - E[x]^2 so that both reductions travel in one message."""
  mean = cross_replica_mean(inputs.mean(dim=axis), group_size)
  if parallel:
    msq = cross_replica_mean((inputs * inputs).mean(dim=axis), group_size)
    return mean, msq - mean * mean
  return mean, cross_replica_mean(((inputs - mean) ** 2).mean(dim=axis), group_size)


class SyncMoments(object):
  """Fused [2C] all-reduces used by standardize_batch on the GPU: local (mean, var) -> global."""

  def __init__(self, group_size=None):
    self.group_size = group_size

  def forward_sync(self, mean, var):
    from compare_gan_amd.hip import kernels as K
    packed = torch.cat([mean, var]).contiguous()   # data movement only
    c = mean.numel()
    K.bn_moments_convert(packed[:c], packed[c:], to_variance=False)
    _, n = cross_replica_sum_(packed, self.group_size)
    K.bn_moments_convert(packed[:c], packed[c:], to_variance=True, scale=1.0 / n)
    return packed[:c], packed[c:]

  def backward_sync(self, m12):
    from compare_gan_amd.hip import kernels as K
    _, n = cross_replica_sum_(m12, self.group_size)
    return K.axpby_f32(m12, 1.0 / n)
