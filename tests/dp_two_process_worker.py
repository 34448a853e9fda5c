"""Worker of test_data_parallel_gpu.test_two_processes_match_one_replica_on_the_global_batch.

argv: rank world port out_path.  Both ranks drive cuda:0 (a single-GPU box cannot host two RCCL
ranks), the process group uses the host-staged `gloo` debugging backend of
tpu_ops.cross_replica_sum_; everything else is the product's data-parallel path: per-replica z /
label streams, cross-replica batch norm forward AND backward (collectives issued from autograd's
device thread), one gradient bucket per network, 1/world scaling.  Two eager training steps on this
replica's shard of the fixed global batch; the variables are written to out_path.<rank>, the
sequence of collectives this rank issued in each step (tpu_ops.record_collectives) to
out_path.<rank>.collectives.  CGAMD_DP_OVERLAP / CGAMD_DP_BUCKET_MIN_MB in the environment choose
between one all-reduce per network after its backward pass and buckets leaving during it.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import gan_util as U  # noqa: E402

CONFIG, BS, STEPS, SEED = "resnet_cifar10.gin", 8, 2, 3
# Adam with epsilon = 1: update = lr * m / (sqrt(v) + 1) ~ lr * m, LINEAR in the gradient -- with
# the default 1e-8 the first steps are sign-like and turn last-bit differences of small gradient
# entries into full-size update differences (measured: update cosine 0.956 after 2 steps, against
# 0.999+ in the linear regime), which would hide a real discrepancy behind a loose tolerance
BINDINGS = ("tf.train.AdamOptimizer.epsilon = 1.0",)


def global_batches(dataset, world, nsub):
    """[step][rank] -> (images [nsub*BS, ...], labels) -- the same on every caller."""
    out = []
    for r in range(world):
        it = dataset.train_batches(BS * nsub, seed=300 + r)
        out.append([next(it) for _ in range(STEPS)])
    return [[out[r][s] for r in range(world)] for s in range(STEPS)]


def main():
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": port, "CGAMD_DIST_BACKEND": "gloo"})
    os.environ.pop("CGAMD_FORCE_DP", None)
    from compare_gan_amd.tpu import tpu_ops
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    assert tpu_ops.init_replicas(dev) == (rank, world)
    assert tpu_ops.in_replica_context()
    gan, options, dataset = U.build_product(CONFIG, BS, dev, seed=SEED, bindings=BINDINGS)
    nsub = options["disc_iters"] + 1
    sequences = []
    for per_rank in global_batches(dataset, world, nsub):
        images, labels = per_rank[rank]
        log = []
        tpu_ops.record_collectives(log)
        gan.train_step(torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev))
        tpu_ops.record_collectives(None)
        sequences.append(log)
    torch.cuda.synchronize()
    assert gan.d_opt.flat is not None
    torch.save(sequences, "%s.%d.collectives" % (out_path, rank))
    torch.save({k: v.detach().cpu() for k, v in gan.store.vars.items()}, "%s.%d" % (out_path, rank))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
    print("DP_WORKER_OK", rank)


if __name__ == "__main__":
    main()
