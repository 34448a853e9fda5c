"""The product's data-parallel path on ONE GPU (modular_gan._OptimizerState.apply_gradients,
tpu_ops.cross_replica_sum_, SyncMoments) -- reference: modular_gan.py:606-616 (CrossShardOptimizer:
mean of the replicas' gradients), arch_ops.py:258-263 (cross-replica batch norm on by default when
data parallel), tpu_random.py:54-78 (per-replica random streams).

Two RCCL ranks cannot share a device, so the N > 1 arithmetic is exercised two ways:
  * CGAMD_FORCE_DP=1: a ONE-rank RCCL group; bucket, ncclAllReduce (captured in the hipGraph),
    1/world scaling and the communication-stream overlap all run, and must reproduce the
    non-data-parallel weights bit for bit (tests/dp_force_worker.py, run in a subprocess because
    the process group is process-global);
  * tpu_ops.InProcessReplicas: two replicas of the product in two threads with an in-memory
    all-reduce -- identical inputs must reproduce the single-replica weights bit for bit
    ((g + g) / 2 == g), different shards must leave both replicas with identical weights that
    differ from a single replica's.
The world_size-2 `gloo` tests of the host logic are in test_data_parallel_gloo.py."""
import os
import subprocess
import sys
import threading
import traceback

import numpy as np
import pytest
import torch

from tests import gan_util as U

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("mode", ["local", "sync"])
def test_force_dp_one_rank_group_matches_single_replica(mode):
    """mode local: per-replica batch norm, every variable bit-identical to the single replica;
    mode sync: cross-replica batch norm (SyncMoments) on the one-rank group, updates agree."""
    env = dict(os.environ)
    env.pop("CGAMD_FORCE_DP", None)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = "29533" if mode == "local" else "29535"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dp_force_worker.py"), mode],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode("utf-8", "replace")
    brief = "\n".join(l for l in out.splitlines() if not l.startswith("frame #"))
    assert p.returncode == 0 and "DP_FORCE_OK" in out, brief[-4000:]


def _batches(dataset, n, seed, steps):
    it = dataset.train_batches(n, seed=seed)
    return [next(it) for _ in range(steps)]


def _run_steps(gan, batches, dev):
    for images, labels in batches:
        gan.train_step(torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev))
    torch.cuda.synchronize()
    return {k: v.detach().clone() for k, v in gan.store.vars.items()}


def _replica_thread(replicas, rank, random_stream, make_gan, batches, dev, out, errs):
    try:
        replicas.attach(rank, random_stream=random_stream)
        gan = make_gan()
        out[rank] = _run_steps(gan, batches, dev)
        replicas.finish(rank)
    except BaseException:  # pylint: disable=broad-except
        errs.append("replica %d:\n%s" % (rank, traceback.format_exc()))
        replicas.finish(rank, error="replica %d raised" % rank)


@pytest.mark.parametrize("config,bs", [("resnet_cifar10.gin", 8)])
def test_two_in_process_replicas(dev, config, bs):
    from compare_gan_amd import datasets, runner_lib
    from compare_gan_amd.tpu import tpu_ops
    steps = 2
    # per-replica batch statistics: the cross-replica batch norm's backward all-reduce is issued
    # from autograd's device thread, which InProcessReplicas does not serve (SyncMoments is
    # covered by the one-rank RCCL test and the gloo tests)
    bindings = ("standardize_batch.use_cross_replica_mean = False",)
    gan0, options, dataset = U.build_product(config, bs, dev, seed=3, bindings=bindings)
    nsub = options["disc_iters"] + 1
    shard = [_batches(dataset, bs * nsub, 100 + r, steps) for r in range(2)]
    single = _run_steps(gan0, shard[0], dev)

    def make_gan():
        gan = options["gan_class"](dataset=datasets.get_dataset(), parameters=options,
                                   model_dir="/tmp/cg_test")
        gan.build(batch_size=bs, device=dev, seed=3)
        return gan

    def run_pair(batches_per_rank, streams):
        replicas = tpu_ops.InProcessReplicas(2)
        out, errs = {}, []
        threads = [threading.Thread(target=_replica_thread, args=(
            replicas, r, streams[r], make_gan, batches_per_rank[r], dev, out, errs))
                   for r in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(600)
        assert not errs, "\n".join(errs)
        assert not any(t.is_alive() for t in threads)
        return out

    # (a) both replicas see replica 0's shard and random stream: mean of two equal gradients
    same = run_pair([shard[0], shard[0]], [0, 0])
    for name, want in single.items():
        assert torch.equal(same[0][name], want), name
        assert torch.equal(same[1][name], want), name
    # (b) different shards, different random streams: replicas stay in lockstep, and the update
    #     is no longer replica 0's own
    diff = run_pair(shard, [0, 1])
    moved = 0
    for name in gan0.store.trainable:
        assert torch.equal(diff[0][name], diff[1][name]), name
        moved += int(not torch.equal(diff[0][name], single[name]))
    assert moved >= len(gan0.store.trainable) // 2
    assert tpu_ops.num_replicas() == 1 and tpu_ops.thread_state() is None
