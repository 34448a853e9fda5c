"""Cross-replica collectives of the data-parallel path.

Reference: compare_gan/tpu/tpu_ops.py:29-125 (cross_replica_concat / cross_replica_mean /
cross_replica_moments on the TPU's `cross_replica_sum`).  Here one process drives one MI355X and
the collective is an RCCL all-reduce over xGMI through torch.distributed (backend "nccl" on the
GPU, "gloo" in the CPU tests); the arithmetic around it (mean^2, scaling) stays in HIP kernels.
"""
import torch
import torch.distributed as dist

from compare_gan_amd import gin

_STATE = {"enabled": False, "groups": {}}


def num_replicas():
  return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def replica_id():
  return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def force_data_parallel():
  """Debug switch (CGAMD_FORCE_DP=1): run the bucket + all-reduce path even on one replica, so the
  collective path can be exercised (and captured) on a single-GPU box."""
  import os
  return os.environ.get("CGAMD_FORCE_DP", "") == "1" and dist.is_available() and dist.is_initialized()


def enable_cross_replica(enabled=True):
  """The analogue of 'running inside a TPU context' (arch_ops.py:258-263)."""
  _STATE["enabled"] = bool(enabled)


def in_replica_context():
  return _STATE["enabled"] and num_replicas() > 1


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
  if (num_replicas() == 1 and not force_data_parallel()) or group_size == 1:
    return tensor, 1
  group, n = _group(group_size)
  dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
  return tensor, n


def cross_replica_mean(inputs, group_size=None):
  """Mean over replicas (tpu_ops.py:75-91).  group_size=1 returns the input unchanged."""
  if num_replicas() == 1 or group_size == 1:
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
