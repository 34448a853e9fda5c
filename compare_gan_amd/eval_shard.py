"""Evaluation sharded over the data-parallel ranks.

The reference evaluates on one host (eval_gan_lib.py:95-212: 3200 generator batches to fill the
batch-norm accumulators, then ceil(N / 64) batches per fake set through the generator and Inception).
After data-parallel training every rank holds the same generator, and every evaluation batch is a
pure function of its index (counter-based latents: tpu_random keys on the name "eval_z/<index>"), so
the batches are dealt round-robin to the ranks and only FEATURES travel:

  * rank r runs batches r, r + W, r + 2W, ... of a phase (generator -> Inception);
  * one all-gather of the [batches, 64, 2048 + 1008] feature block (10k samples: 122 MB, one RCCL
    ring pass over xGMI) puts the complete set, in single-rank order, on every rank -- the metric
    tasks then see exactly the arrays a single-rank evaluation produces;
  * the accumulator fill adds per-rank partial sums: the deltas of all accumulators are all-reduced
    as ONE flat bucket.

Everything here is device-agnostic torch + torch.distributed, so the protocol is covered by a
world-size-2 gloo test on CPU (tests/test_eval_shard.py); on the GPU the same code runs over RCCL."""
import math

import torch
import torch.distributed as dist


def rank_world(shard=None):
  """(rank, world) of the evaluation: `shard` False -> single rank; None -> the default process
  group if one is initialised; or an explicit (rank, world) pair."""
  if shard is False:
    return 0, 1
  if isinstance(shard, (tuple, list)):
    return int(shard[0]), int(shard[1])
  if dist.is_available() and dist.is_initialized():
    return dist.get_rank(), dist.get_world_size()
  return 0, 1


def shard_indices(num_batches, rank, world):
  """Round-robin deal: the batch indices (0-based within the phase) rank `rank` runs."""
  return list(range(rank, num_batches, world))


def gather_in_order(local, num_batches, rank, world, group=None):
  """local: the tensors [B, ...] of batches rank, rank + world, ... -> all `num_batches` batches
  concatenated in index order, on every rank.  One all-gather; ranks with one batch fewer pad."""
  if world == 1:
    return torch.cat(local, dim=0)
  per = int(math.ceil(num_batches / world))
  assert len(local) == len(shard_indices(num_batches, rank, world))
  if local:
    proto = local[0]
  else:
    raise ValueError("every rank needs at least one batch (num_batches %d < world %d)" %
                     (num_batches, world))
  block = torch.stack(list(local) + [torch.zeros_like(proto)] * (per - len(local)), dim=0)
  parts = [torch.empty_like(block) for _ in range(world)]
  dist.all_gather(parts, block.contiguous(), group=group)
  out = torch.stack(parts, dim=1)            # [per, world, B, ...]: batch index j * world + r
  out = out.reshape((per * world * proto.shape[0],) + tuple(proto.shape[1:]))
  return out[:num_batches * proto.shape[0]]


def gather_rows(local_rows, total_rows, rank, world, chunk, group=None):
  """Contiguous row chunks (rank r holds rows [r * chunk, min((r + 1) * chunk, total))) -> all rows."""
  if world == 1:
    return local_rows
  pad = chunk - local_rows.shape[0]
  if pad:
    local_rows = torch.cat([local_rows, local_rows.new_zeros((pad,) + tuple(local_rows.shape[1:]))])
  parts = [torch.empty_like(local_rows) for _ in range(world)]
  dist.all_gather(parts, local_rows.contiguous(), group=group)
  return torch.cat(parts, dim=0)[:total_rows]


def row_range(total_rows, rank, world, chunk):
  """[lo, hi) of the rows rank `rank` holds; lo == hi when the chunks run out before the ranks do."""
  if world == 1:
    return 0, total_rows
  lo = min(rank * chunk, total_rows)
  return lo, min(lo + chunk, total_rows)


def row_chunk(total_rows, world, multiple):
  """Rows per rank: ceil(total / world) rounded up to a multiple of the Inception batch."""
  per = int(math.ceil(total_rows / world))
  return int(math.ceil(per / multiple)) * multiple


def allreduce_deltas(tensors, before, world, group=None):
  """tensors[i] = before[i] + (this rank's additions) -> before[i] + the additions of ALL ranks.
  One flat fp32 bucket, one all-reduce."""
  if world == 1 or not tensors:
    return
  flat = torch.cat([(t.detach().float() - b.float()).reshape(-1) for t, b in zip(tensors, before)])
  dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
  off = 0
  for t, b in zip(tensors, before):
    n = t.numel()
    t.detach().copy_((b.float() + flat[off:off + n].reshape(t.shape)).to(t.dtype))
    off += n


def any_rank(flag, device, world, group=None):
  """True on every rank if `flag` is true on any (NaN detection must abort all ranks together)."""
  if world == 1:
    return bool(flag)
  t = torch.tensor([1.0 if flag else 0.0], device=device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
  return bool(t.item() > 0)


class TorchImageSink(object):
  """Collects the batches of one rank: concatenation + NaN test with torch ops (any device)."""

  def __init__(self, num_batches):
    del num_batches
    self._imgs = []

  def add(self, images):
    self._imgs.append(images)

  def finish(self):
    local = torch.cat(self._imgs, dim=0)
    return local, bool(torch.isnan(local).any())


def sharded_fake_features(generate_batch, transform, num_batches, first_index, rank, world,
                          keep_images=True, group=None, timing=None, tick=None, sink_cls=None):
  """One fake evaluation set.  generate_batch(index) -> [B, H, W, C] images of global batch
  `index`; transform(images) -> (activations, logits).  sink_cls(num_local_batches) collects this
  rank's batches (add) and returns (all of them [n * B, H, W, C], NaN seen) from finish():
  TorchImageSink keeps them as they come (images already in [0, 255]); the product passes
  eval_utils.FakeImageSink, whose one HIP pass per batch scales [0, 1] -> [0, 255] straight into the
  set's buffer and counts NaNs (eval_utils.py:144-162).  Returns (images or None, activations,
  logits, nan_found) with all `num_batches` batches in order on every rank."""
  def lap(key, since):
    if timing is None or tick is None:
      return None
    now = tick()
    timing[key] = timing.get(key, 0.0) + now - since
    return now

  t = tick() if tick is not None else None
  mine = shard_indices(num_batches, rank, world)
  sink = (sink_cls or TorchImageSink)(len(mine))
  for j, i in enumerate(mine):
    sink.add(generate_batch(first_index + i))
    if j == 0 and timing is not None and tick is not None:
      # the first batch on its own: a one-off cost (code objects paged in on a fresh box, allocator
      # growth) shows up here instead of hiding in the phase total
      timing["sample_first_batch"] = tick() - t
  local, local_nan = sink.finish()
  nan_found = any_rank(local_nan, local.device, world, group)
  t = lap("sample", t)
  if nan_found:
    return None, None, None, True
  act, logits = transform(local)
  t = lap("inception", t)
  bsz = local.shape[0] // max(1, len(mine))
  imgs = list(local.reshape((len(mine), bsz) + tuple(local.shape[1:]))) if world > 1 else None
  if world == 1:
    return (local if keep_images else None), act, logits, False
  act = gather_in_order(list(act.reshape((len(mine), bsz) + tuple(act.shape[1:]))), num_batches,
                        rank, world, group)
  logits = gather_in_order(list(logits.reshape((len(mine), bsz) + tuple(logits.shape[1:]))),
                           num_batches, rank, world, group)
  images = gather_in_order(imgs, num_batches, rank, world, group) if keep_images else None
  lap("gather", t)
  return images, act, logits, False
