"""The N > 1 path on CPU: world_size-2 `gloo` processes exercising the host-side collectives of
compare_gan_amd/tpu/tpu_ops.py (the RCCL path on the GPU box runs the same code with backend
"nccl") and the data-parallel arithmetic they must reproduce.

Reference pins / properties:
  * cross-replica batch norm == full-batch batch norm    tpu/tpu_ops_test.py, arch_ops_tpu_test.py:112-133
  * cross_replica_concat ordering                          tpu/tpu_ops.py:29-72
  * CrossShardOptimizer: mean of per-replica gradients ==  modular_gan.py:606-616
    gradient of the global-batch loss (equal shards)
  * per-replica random streams are distinct, per-step reproducible   tpu/tpu_random_test.py:146-168
"""
import os
import socket
import sys
import traceback

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 2


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _worker(rank, port, fn_name, errq):
  try:
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    globals()[fn_name](rank)
    dist.barrier()
    dist.destroy_process_group()
  except Exception:  # pylint: disable=broad-except
    errq.put("rank %d:\n%s" % (rank, traceback.format_exc()))


def _run(fn_name):
  ctx = mp.get_context("spawn")
  errq = ctx.SimpleQueue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, port, fn_name, errq)) for r in range(WORLD)]
  for p in procs:
    p.start()
  for p in procs:
    p.join(180)
  errs = []
  while not errq.empty():
    errs.append(errq.get())
  for p in procs:
    if p.is_alive():
      p.terminate()
      errs.append("worker timed out")
    elif p.exitcode != 0:
      errs.append("worker exit code %s" % p.exitcode)
  assert not errs, "\n".join(errs)


# ---- bodies (run inside each rank) ---------------------------------------------------------------
def _body_collectives(rank):
  from compare_gan_amd.tpu import tpu_ops
  assert tpu_ops.num_replicas() == WORLD and tpu_ops.replica_id() == rank
  g = torch.Generator().manual_seed(11)
  full = torch.randn(8, 4, 4, 6, generator=g, dtype=torch.float64)
  mine = full[rank * 4:(rank + 1) * 4]
  # mean
  m = tpu_ops.cross_replica_mean(mine.mean(dim=(0, 1, 2)))
  assert torch.allclose(m, full.mean(dim=(0, 1, 2)), atol=1e-12)
  # group_size = 1 is the identity (tpu_ops.py:80-81)
  assert torch.equal(tpu_ops.cross_replica_mean(mine, group_size=1), mine)
  # moments, both formulations (tpu_ops.py:109-125)
  for parallel in (True, False):
    mean, var = tpu_ops.cross_replica_moments(mine, axis=(0, 1, 2), parallel=parallel)
    assert torch.allclose(mean, full.mean(dim=(0, 1, 2)), atol=1e-12)
    assert torch.allclose(var, full.var(dim=(0, 1, 2), unbiased=False), atol=1e-10)
  # concat keeps replica order
  cat = tpu_ops.cross_replica_concat(mine, rank, WORLD)
  assert torch.equal(cat, full)
  # in-place sum
  t = torch.full((3,), float(rank + 1), dtype=torch.float32)
  out, n = tpu_ops.cross_replica_sum_(t)
  assert n == WORLD and out.tolist() == [3.0, 3.0, 3.0]
  # replica context switch (arch_ops.py:258-263)
  assert not tpu_ops.in_replica_context()
  tpu_ops.enable_cross_replica(True)
  assert tpu_ops.in_replica_context()
  tpu_ops.enable_cross_replica(False)


def _body_sync_bn_matches_full_batch(rank):
  """arch_ops_tpu_test.py:112-133: BN over shards with cross-replica moments equals BN over the
  concatenated batch -- forward AND the gradient w.r.t. the input."""
  from compare_gan_amd.tpu import tpu_ops
  from oracle import arch_ops as oops

  class _AllReduceMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
      return tpu_ops.cross_replica_mean(t)

    @staticmethod
    def backward(ctx, g):   # adjoint of the mean over replicas
      return tpu_ops.cross_replica_mean(g)

  def sync(mean, mean_sq):
    return _AllReduceMean.apply(mean), _AllReduceMean.apply(mean_sq)

  g = torch.Generator().manual_seed(5)
  full = torch.randn(8, 3, 3, 5, generator=g, dtype=torch.float64)
  w = torch.randn(8, 3, 3, 5, generator=g, dtype=torch.float64)
  # reference: single-process full batch
  xf = full.clone().requires_grad_(True)
  vs = oops.VarStore()
  yf = oops.standardize_batch(vs, xf, True, "", oops.BNConfig(0.9, 1e-5))
  (yf * w).sum().backward()
  # sharded
  xs = full[rank * 4:(rank + 1) * 4].clone().requires_grad_(True)
  vs2 = oops.VarStore()
  ys = oops.standardize_batch(vs2, xs, True, "", oops.BNConfig(0.9, 1e-5, cross_replica=sync))
  # each replica's loss is the sum over its shard; the global loss is the sum over replicas, and
  # the adjoint of mean-over-replicas is again mean-over-replicas, so x.grad is the global gradient
  (ys * w[rank * 4:(rank + 1) * 4]).sum().backward()
  assert torch.allclose(ys, yf[rank * 4:(rank + 1) * 4], atol=1e-10)
  assert torch.allclose(xs.grad, xf.grad[rank * 4:(rank + 1) * 4], atol=1e-9)
  assert torch.allclose(vs2.vars["moving_mean"], vs.vars["moving_mean"], atol=1e-12)
  assert torch.allclose(vs2.vars["moving_variance"], vs.vars["moving_variance"], atol=1e-12)


def _body_gradient_mean_equals_global_batch(rank):
  """One D sub-step of resnet_cifar10.gin (no BN in D): the all-reduced mean of the per-replica
  gradients equals the gradient of the loss over the global batch."""
  from compare_gan_amd.tpu import tpu_ops
  from oracle import arch_ops as oops
  from tests import gan_util as U
  bsz = 2
  g = torch.Generator().manual_seed(9)
  images = torch.rand(WORLD * bsz, 32, 32, 3, generator=g, dtype=torch.float64)
  fakes = torch.rand(WORLD * bsz, 32, 32, 3, generator=g, dtype=torch.float64)

  def d_grads(img, fk):
    vs = oops.VarStore(seed=1)
    ora = U.build_oracle("resnet_cifar10.gin", vs)
    d_loss, _, _ = ora.create_loss(img, fk, None, None)
    return torch.autograd.grad(d_loss, ora.d_vars()), float(d_loss)

  grads_full, loss_full = d_grads(images, fakes)
  sl = slice(rank * bsz, (rank + 1) * bsz)
  grads_mine, loss_mine = d_grads(images[sl], fakes[sl])
  flat = torch.cat([t.reshape(-1) for t in grads_mine]).to(torch.float32)   # bucket
  tpu_ops.cross_replica_sum_(flat)
  flat = flat.double() / WORLD
  ref = torch.cat([t.reshape(-1) for t in grads_full])
  assert float((flat - ref).norm() / ref.norm()) < 1e-6
  lm = tpu_ops.cross_replica_mean(torch.tensor([loss_mine], dtype=torch.float64))
  assert abs(float(lm) - loss_full) < 1e-10


def _body_per_replica_streams(rank):
  from compare_gan_amd.tpu import tpu_ops
  from oracle import rng as orng
  from tests import gan_util as U
  z = orng.uniform(64, -1, 1, 3, U.op_id("z/0"), tpu_ops.replica_id(), 0)
  zs = [torch.zeros(64, dtype=torch.float64) for _ in range(WORLD)]
  dist.all_gather(zs, torch.from_numpy(np.asarray(z, dtype=np.float64)))
  assert not torch.equal(zs[0], zs[1])
  again = orng.uniform(64, -1, 1, 3, U.op_id("z/0"), rank, 0)
  assert np.array_equal(z, again)
  # the synthetic input pipeline shards by replica (runner_lib: seed + replica id)
  from compare_gan_amd import datasets
  ds = datasets.DATASETS["cifar10"](seed=547)
  a = next(ds.train_batches(4, seed=547 + rank))[0]
  parts = [torch.zeros(a.shape, dtype=torch.float32) for _ in range(WORLD)]
  dist.all_gather(parts, torch.from_numpy(a))
  assert not torch.equal(parts[0], parts[1])


def _body_reference_goldens(rank):
  """The literal vectors the reference's own two-core tests hold, on two gloo ranks:
  tpu/tpu_ops_test.py:44-65 (cross_replica_concat), :73-101 (cross_replica_mean, group_size None /
  0 / 2), :103-128 (group_size 1 = identity), and architectures/arch_ops_tpu_test.py:30-52,112-133
  (batch norm over 4 images split across two cores with cross-replica moments == the golden array)."""
  import numpy as np
  from compare_gan_amd.tpu import tpu_ops
  from oracle import arch_ops as oops
  # batch_parallel splits axis 0 over the cores: core r sees row r
  x = torch.tensor([[3, 4], [1, 5]], dtype=torch.float32)[rank:rank + 1]
  cat = tpu_ops.cross_replica_concat(x, rank, WORLD)
  assert cat.tolist() == [[3.0, 4.0], [1.0, 5.0]]          # (the test concatenates both cores' copies)
  inputs = torch.tensor([[0.55, 0.70, -1.29, 0.502], [0.57, 0.90, 1.290, 0.202]], dtype=torch.float32)
  want = torch.tensor([0.56, 0.8, 0.0, 0.352], dtype=torch.float32)
  for group_size in (None, 0, 2):
    got = tpu_ops.cross_replica_mean(inputs[rank:rank + 1], group_size=group_size)
    assert torch.allclose(got[0], want, rtol=1e-6, atol=1e-6), (group_size, got)
  got = tpu_ops.cross_replica_mean(inputs[rank:rank + 1], group_size=1)
  assert torch.equal(got, inputs[rank:rank + 1])
  # arch_ops_tpu_test.py: 4 images of 2x1x3, two per core
  x1 = [[[5, 7, 2]], [[5, 8, 8]]]
  x2 = [[[1, 2, 0]], [[4, 0, 4]]]
  x3 = [[[6, 2, 6]], [[5, 0, 5]]]
  x4 = [[[2, 4, 2]], [[6, 4, 1]]]
  images = torch.tensor([x1, x2, x3, x4], dtype=torch.float64)
  expected = torch.tensor(
      [[[[0.4375205, 1.30336881, -0.58830315]], [[0.4375205, 1.66291881, 1.76490951]]],
       [[[-1.89592218, -0.49438119, -1.37270737]], [[-0.14584017, -1.21348119, 0.19610107]]],
       [[[1.02088118, -0.49438119, 0.98050523]], [[0.4375205, -1.21348119, 0.58830321]]],
       [[[-1.31256151, 0.22471881, -0.58830315]], [[1.02088118, 0.22471881, -0.98050523]]]],
      dtype=torch.float64)

  def sync(mean, mean_sq):
    return tpu_ops.cross_replica_mean(mean), tpu_ops.cross_replica_mean(mean_sq)

  vs = oops.VarStore()
  mine = images[rank * 2:(rank + 1) * 2]
  # (tf.layers' default epsilon in that test is 1e-3: arch_ops_tpu_test.py:117-121 calls
  # standardize_batch with its default epsilon)
  y = oops.standardize_batch(vs, mine, True, "", oops.BNConfig(0.999, 1e-3, cross_replica=sync))
  assert torch.allclose(y, expected[rank * 2:(rank + 1) * 2], rtol=1e-5, atol=1e-5), (rank, y)
  del np


# ---- tests -----------------------------------------------------------------------------------------
def _body_collective_sequence(rank):
  """tpu_ops.record_collectives: the recorded (kind, numel, dtype, group, thread) sequence of a rank,
  with collectives issued from the main thread AND from autograd's worker thread (a custom Function's
  backward, as the cross-replica batch norm's is), equals the other rank's."""
  from compare_gan_amd.tpu import tpu_ops

  class SumInBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
      return x.clone()

    @staticmethod
    def backward(ctx, g):
      g = g.clone()
      tpu_ops.cross_replica_sum_(g)
      return g

  log = []
  assert tpu_ops.record_collectives(log) is None
  x = torch.randn(4, 5, dtype=torch.float64, requires_grad=True)
  tpu_ops.cross_replica_moments(x.detach(), axis=(0,), parallel=True)      # 2 all-reduces, main thread
  y = SumInBackward.apply(x * (rank + 1.0)).sum()
  y.backward()                                                              # 1 all-reduce in backward
  tpu_ops.cross_replica_concat(torch.zeros(3), rank, WORLD)                 # all-gather
  tpu_ops.cross_replica_mean(torch.zeros(7), group_size=1)                  # identity: nothing issued
  assert tpu_ops.record_collectives(None) is log
  tpu_ops.cross_replica_mean(torch.zeros(7))                                # not recorded any more
  assert [c[:2] for c in log] == [("all_reduce_sum", 5), ("all_reduce_sum", 5),
                                  ("all_reduce_sum", 20), ("all_gather", 3)], log
  assert log[0][2] == "float64" and log[0][4] == "main"
  both = [None] * WORLD
  dist.all_gather_object(both, log)
  assert both[0] == both[1], both
  assert torch.allclose(x.grad, torch.full((4, 5), 2.0 * (rank + 1), dtype=torch.float64))   # ones summed over 2 ranks


def test_collective_sequence_world2():
  _run("_body_collective_sequence")


def test_reference_goldens_world2():
  _run("_body_reference_goldens")


def test_collectives_world2():
  _run("_body_collectives")


def test_sync_bn_matches_full_batch_world2():
  _run("_body_sync_bn_matches_full_batch")


def test_gradient_mean_equals_global_batch_world2():
  _run("_body_gradient_mean_equals_global_batch")


def test_per_replica_streams_world2():
  _run("_body_per_replica_streams")


def _body_launcher_init(rank):
  """What bench.py / runner_lib / compare_gan_amd.main call per process: the group is already
  up here (init_replicas must not re-initialise it), cross-replica batch norm is switched on."""
  from compare_gan_amd.tpu import tpu_ops
  os.environ["WORLD_SIZE"] = str(WORLD)
  os.environ["RANK"] = str(rank)
  tpu_ops.enable_cross_replica(False)
  assert not tpu_ops.in_replica_context()
  assert tpu_ops.init_replicas(None) == (rank, WORLD)
  assert tpu_ops.data_parallel() and tpu_ops.in_replica_context()
  assert tpu_ops.random_stream_id() == rank
  tpu_ops.enable_cross_replica(False)


def test_launcher_init_world2():
  _run("_body_launcher_init")


def test_init_replicas_single_process_is_a_no_op():
  from compare_gan_amd.tpu import tpu_ops
  env = {k: os.environ.pop(k, None) for k in ("WORLD_SIZE", "CGAMD_FORCE_DP")}
  try:
    assert tpu_ops.init_replicas(None) == (0, 1)
    assert not dist.is_initialized() and not tpu_ops.data_parallel()
    assert not tpu_ops.in_replica_context()
  finally:
    for k, v in env.items():
      if v is not None:
        os.environ[k] = v


def test_in_process_replicas_all_reduce():
  """tpu_ops.InProcessReplicas (the single-GPU stand-in for two RCCL ranks, used by
  tests/test_data_parallel_gpu.py): baton order, rank-ordered sums, several collectives per
  thread, per-thread state, and error propagation."""
  import threading
  from compare_gan_amd.tpu import tpu_ops
  world = 3
  replicas = tpu_ops.InProcessReplicas(world)
  out, errs, running = {}, [], []

  def body(rank):
    try:
      replicas.attach(rank)
      assert tpu_ops.num_replicas() == world and tpu_ops.replica_id() == rank
      assert tpu_ops.data_parallel() and tpu_ops.random_stream_id() == rank
      tpu_ops.thread_state()["mine"] = rank
      vals = []
      for k in range(4):
        running.append(rank)
        t = torch.full((5,), float((rank + 1) * 10 ** k), dtype=torch.float64)
        res, n = tpu_ops.cross_replica_sum_(t)
        assert n == world and tpu_ops.thread_state()["mine"] == rank
        vals.append(float(res[0]))
      out[rank] = vals
      replicas.finish(rank)
    except BaseException:  # pylint: disable=broad-except
      errs.append(traceback.format_exc())
      replicas.finish(rank, error="rank %d" % rank)

  threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(60)
  assert not errs, "\n".join(errs)
  for r in range(world):
    assert out[r] == [6.0 * 10 ** k for k in range(4)]
  assert tpu_ops.num_replicas() == 1 and tpu_ops.thread_state() is None

  # a failing replica releases the others instead of leaving them waiting for the baton
  replicas = tpu_ops.InProcessReplicas(2)
  seen = []

  def failing(rank):
    try:
      replicas.attach(rank)
      if rank == 1:
        raise ValueError("boom")
      tpu_ops.cross_replica_sum_(torch.zeros(2))
      replicas.finish(rank)
    except BaseException as e:  # pylint: disable=broad-except
      seen.append(type(e).__name__)
      replicas.finish(rank, error=repr(e))

  threads = [threading.Thread(target=failing, args=(r,)) for r in range(2)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(60)
  assert not any(t.is_alive() for t in threads)
  assert sorted(seen) == ["RuntimeError", "ValueError"]


def test_plan_buckets_contiguous_tail_first_and_balanced():
  """modular_gan.plan_buckets: the gradient buckets that leave during the backward pass."""
  from compare_gan_amd.gans.modular_gan import plan_buckets
  rng = np.random.default_rng(3)
  for trial in range(50):
    n = int(rng.integers(1, 60))
    numels = [int(x) for x in rng.integers(1, 5_000_000, size=n)]
    nb = int(rng.integers(1, 6))
    min_bytes = int(rng.integers(1, 32)) << 20
    ranges = plan_buckets(numels, nb, min_bytes)
    total = 4 * sum(numels)
    assert 1 <= len(ranges) <= max(1, min(nb, total // min_bytes, n))
    assert ranges[0][1] == n and ranges[-1][0] == 0          # launch order: the tail first
    for (lo, hi), (lo2, hi2) in zip(ranges[:-1], ranges[1:]):
      assert hi2 == lo and lo2 < hi2                           # contiguous, no gaps, no overlap
    assert sorted(i for lo, hi in ranges for i in range(lo, hi)) == list(range(n))
    if len(ranges) > 1:
      target = total / len(ranges)
      biggest = 4 * max(numels)
      for lo, hi in ranges[:-1]:
        got = 4 * sum(numels[lo:hi])
        assert target <= got < target + biggest              # closes as soon as the target is met
  # a network below 2 x min_bytes is one bucket (= the single flat all-reduce of round 2)
  assert plan_buckets([1000, 2000, 3000], 4, 16 << 20) == [(0, 3)]
  # BigGAN-sized example: 70 M parameters in two ~140 MB halves
  numels = [4_000_000] * 17 + [1_000_000] * 2
  r = plan_buckets(numels, 2, 16 << 20)
  assert len(r) == 2 and abs(4 * sum(numels[r[0][0]:]) - 4 * sum(numels[:r[1][1]])) <= 2 * 16_000_000
