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
  * two PROCESSES sharing the GPU, summing through the host (`gloo`, CGAMD_DIST_BACKEND): the
    complete product path including cross-replica batch norm forward and backward, checked
    against ONE replica training on the concatenated global batch with the concatenated random
    draws (tests/dp_two_process_worker.py).
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


@pytest.mark.parametrize("mode", ["local", "buckets", "buckets_r5", "buckets_nu", "sync"])
def test_force_dp_one_rank_group_matches_single_replica(mode):
    """mode local: per-replica batch norm, every variable bit-identical to the single replica;
    mode sync: cross-replica batch norm (SyncMoments) on the one-rank group, updates agree."""
    env = dict(os.environ)
    env.pop("CGAMD_FORCE_DP", None)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = {"local": "29533", "buckets": "29537", "buckets_r5": "29541", "buckets_nu": "29539",
                          "sync": "29535"}[mode]
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dp_force_worker.py"), mode],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode("utf-8", "replace")
    brief = "\n".join(l for l in out.splitlines() if not l.startswith("frame #"))
    assert p.returncode == 0 and "DP_FORCE_OK" in out, brief[-4000:]


def test_bench_entry_point_with_two_ranks():
    """`bench.py --gpus 2` exactly as the driver launches it for the scaling
    curve (python -m torch.distributed.run --nproc-per-node 2 ...), dry-run on ONE GPU: both ranks on
    cuda:0, host-side all-reduce (gloo; two RCCL ranks cannot share a device), eager launches.  What
    this executes before an 8-GPU node ever does: RANK / LOCAL_RANK plumbing, per-rank data shards,
    cross-replica batch norm + bucketed gradient all-reduce, the barrier-bracketed timed region with
    MAX over ranks, the threaded communicator teardown, and rank 0 alone printing ONE JSON line whose
    `value` counts the images of BOTH ranks."""
    import json
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CGAMD_FORCE_DP"):
        env.pop(k, None)
    env.update({"CGAMD_BENCH_DEVICE": "0", "CGAMD_DIST_BACKEND": "gloo", "CGAMD_DP_GRAPH": "0",
                "PYTHONPATH": ROOT + os.pathsep + env.get("PYTHONPATH", "")})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29561", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--preheat-s", "0", "--no-legs", "--no-fid",
           "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    out, err = p.stdout.decode("utf-8", "replace"), p.stderr.decode("utf-8", "replace")
    assert p.returncode == 0, (out[-2000:], err[-3000:])
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]                      # rank 0 only, once
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["hip_graph"] is False
    # whole-job throughput: both ranks' images over the slowest rank's time
    assert abs(d["value"] - 2 * 64 * 6 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-3 * d["value"]
    assert out.strip().splitlines()[-1] == lines[0]          # the JSON line is the LAST line


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


def test_two_processes_match_one_replica_on_the_global_batch(dev, tmp_path):
    """modular_gan.py:606-616 + arch_ops.py:258-263: two replicas with batch B each, gradient mean
    and cross-replica batch norm, ARE one replica with batch 2B -- provided that replica sees the
    concatenation of the shards and of the replicas' random draws (tpu_random.py: every replica
    has its own stream).  Tolerance: the reductions run in a different order (per-replica partial
    sums) and bf16 activations re-round; Adam runs with epsilon = 1 (see the worker) so that the
    update is linear in the gradient: cosine of the two-step update over all variables >= 0.999
    and norm ratio within 0.5 % (a missing 1/world would show here); per variable the distance
    from the reference may not exceed 3x the distance between two reference runs that only
    differ in the order of the shards (+ 2e-3); the replicas themselves agree bit for bit."""
    import socket
    from compare_gan_amd.tpu import tpu_ops, tpu_random
    from tests import dp_two_process_worker as W
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    out_path = str(tmp_path / "vars.pt")
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_two_process_worker.py"),
                               str(r), str(world), port, out_path], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]

    # meanwhile: the single replica on the global batch, its random draws the concatenation of
    # the two replicas' streams -- twice, with the shards in either order: the difference between
    # those two runs (same mathematics, other summation order) is the rounding-noise floor the
    # data-parallel result is held against
    stream = [0]
    saved = (tpu_ops.random_stream_id, tpu_random.uniform, tpu_random.normal, tpu_random.labels)

    def run_reference(order):
        gan, options, dataset = U.build_product(W.CONFIG, world * W.BS, dev, seed=W.SEED,
                                                bindings=W.BINDINGS)
        init = {k: v.detach().cpu().clone() for k, v in gan.store.vars.items()}
        nsub = options["disc_iters"] + 1

        def halves(fn):
            def draw(shape, name, *a, **kw):
                parts = []
                for r in order:
                    stream[0] = r
                    parts.append(fn((shape[0] // world,) + tuple(shape[1:]), name, *a, **kw))
                return torch.cat(parts, dim=0)
            return draw

        def labels(n, num_classes, name, device=None):
            parts = []
            for r in order:
                stream[0] = r
                parts.append(saved[3](n // world, num_classes, name, device=device))
            return torch.cat(parts, dim=0)

        try:
            tpu_ops.random_stream_id = lambda: stream[0]
            tpu_random.uniform, tpu_random.normal = halves(saved[1]), halves(saved[2])
            tpu_random.labels = labels
            for per_rank in W.global_batches(dataset, world, nsub):
                imgs, labs = [], []
                for i in range(nsub):       # sub-step i of the global batch = the shards' sub-step i
                    for r in order:
                        imgs.append(per_rank[r][0][i * W.BS:(i + 1) * W.BS])
                        labs.append(per_rank[r][1][i * W.BS:(i + 1) * W.BS])
                gan.train_step(torch.from_numpy(np.concatenate(imgs)).to(dev),
                               torch.from_numpy(np.concatenate(labs)).to(dev))
            torch.cuda.synchronize()
        finally:
            (tpu_ops.random_stream_id, tpu_random.uniform, tpu_random.normal,
             tpu_random.labels) = saved
        final = {k: v.detach().cpu().clone() for k, v in gan.store.vars.items()}
        return init, final, list(gan.store.trainable)

    init, ref_a, trainable = run_reference((0, 1))
    _, ref_b, _ = run_reference((1, 0))
    outs = [p.communicate(timeout=600)[0].decode("utf-8", "replace") for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0 and "DP_WORKER_OK" in outs[r], outs[r][-3000:]
    reps = [torch.load("%s.%d" % (out_path, r)) for r in range(world)]
    for name in trainable:
        assert torch.equal(reps[0][name], reps[1][name]), name

    def compare(other):
        """Update cosine / norm ratio against ref_a: {name: (cos, ratio)} and the global pair."""
        per, num, den_a, den_b = {}, 0.0, 0.0, 0.0
        for name in trainable:
            da = (ref_a[name] - init[name]).double().flatten()
            db = (other[name] - init[name]).double().flatten()
            num += float(torch.dot(da, db))
            den_a += float(da.norm() ** 2)
            den_b += float(db.norm() ** 2)
            if float(da.norm()) > 0 and float(db.norm()) > 0:
                per[name] = (float(torch.dot(da, db) / (da.norm() * db.norm())),
                             float(db.norm() / da.norm()))
        return per, num / (den_a * den_b) ** 0.5, (den_b / den_a) ** 0.5

    noise, noise_cos, _ = compare(ref_b)
    dp, dp_cos, dp_ratio = compare(reps[0])
    worse = []
    for name, (cos, ratio) in dp.items():
        floor = 1.0 - noise[name][0]
        if (1.0 - cos) > 3.0 * floor + 2e-3 or abs(ratio - 1.0) > 3.0 * abs(noise[name][1] - 1.0) + 0.02:
            worse.append((name, round(cos, 4), round(noise[name][0], 4), round(ratio, 4)))
    print("two-process data parallel vs global batch: update cosine %.5f (shard-order noise %.5f), "
          "norm ratio %.5f, beyond noise: %s" % (dp_cos, noise_cos, dp_ratio, worse))
    assert dp_cos >= 0.999 and abs(dp_ratio - 1.0) <= 0.005, (dp_cos, dp_ratio)
    assert not worse, worse


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_ranks_issue_their_collectives_in_the_same_order(dev, tmp_path, overlap):
    """RCCL requires every rank to issue the collectives of a communicator in the same order; here two
    threads issue them (main thread: the cross-replica batch norms of the forward passes; autograd's
    device thread: their backward all-reduces and the gradient buckets, modular_gan._OptimizerState.arm
    -- reference: modular_gan.py:606-616 CrossShardOptimizer, tpu/tpu_ops.py:94-125).  Two processes run
    two data-parallel steps of resnet_cifar10.gin each (both on cuda:0, sums through the host) and
    record (kind, numel, dtype, group, issuing thread) of every collective: the sequences of the two
    ranks must be identical, step by step, with the gradients leaving in one all-reduce per network
    after its backward pass (overlap 0) and in buckets during it (overlap 1, 1 MiB buckets so that
    the small networks split); the step must contain both threads' collectives, and with the overlap
    on the first gradient bucket must leave BEFORE the backward pass's last batch-norm all-reduce."""
    import socket
    from tests import dp_two_process_worker as W
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    out_path = str(tmp_path / "vars.pt")
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.update({"CGAMD_DP_OVERLAP": overlap, "CGAMD_DP_BUCKET_MIN_MB": "1", "CGAMD_DP_BUCKETS": "2"})
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_two_process_worker.py"),
                               str(r), str(world), port, out_path], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode("utf-8", "replace") for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0 and "DP_WORKER_OK" in outs[r], outs[r][-3000:]
    seqs = [torch.load("%s.%d.collectives" % (out_path, r)) for r in range(world)]
    assert len(seqs[0]) == W.STEPS
    for step in range(W.STEPS):
        a, b = seqs[0][step], seqs[1][step]
        assert len(a) == len(b) and len(a) > 0
        for i, (ca, cb) in enumerate(zip(a, b)):
            assert tuple(ca) == tuple(cb), (step, i, ca, cb)
        threads = {c[4] for c in a}
        assert threads == {"main", "worker"}, threads
        # gradient all-reduces are the large fp32 ones (batch-norm messages are [2C], C <= 256)
        grads = [i for i, c in enumerate(a) if c[1] > 4096]
        bn_bwd = [i for i, c in enumerate(a) if c[4] == "worker" and c[1] <= 4096]
        assert grads and bn_bwd
        if overlap == "1":
            assert any(g < max(bn_bwd) for g in grads), (grads, max(bn_bwd))
        else:
            # 5 D sub-steps + 1 G sub-step: one all-reduce each
            assert len(grads) == 6, len(grads)
    assert seqs[0][0] == seqs[0][1]     # and the sequence is the same from step to step
    print("collectives per step:", len(seqs[0][0]), "gradient all-reduces:",
          len([c for c in seqs[0][0] if c[1] > 4096]), "overlap", overlap)
