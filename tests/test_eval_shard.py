"""compare_gan_amd/eval_shard.py over a real 2-rank gloo process group on CPU: the sharded
evaluation protocol (round-robin batches, feature all-gather, accumulator delta all-reduce, NaN
consensus) reproduces the single-rank arrays EXACTLY, hence every metric computed from them
(FID / IS / KID read only these arrays) is identical to the single-rank evaluation."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from compare_gan_amd import eval_shard

B, H, F, L = 8, 4, 32, 10


def _generate(index):
    """Stand-in for generator -> [0, 255] images: a pure function of the batch index."""
    g = torch.Generator().manual_seed(1000 + index)
    return torch.rand((B, H, H, 3), generator=g, dtype=torch.float32) * 255.0


_W_ACT = torch.linspace(-1, 1, H * H * 3 * F, dtype=torch.float64).reshape(H * H * 3, F)
_W_LOG = torch.cos(torch.arange(F * L, dtype=torch.float64)).reshape(F, L)


def _transform(images):
    """Stand-in for Inception: per-row, so batch composition cannot change a row's features."""
    act = torch.tanh(images.double().reshape(images.shape[0], -1) / 255.0 @ _W_ACT)
    return act.float(), (act @ _W_LOG).float()


def _fid(a, b):
    """Plain float64 Frechet distance (the product's solver is a HIP kernel; here only the fact
    that identical arrays give identical numbers matters)."""
    a, b = a.double().numpy(), b.double().numpy()
    m1, m2 = a.mean(0), b.mean(0)
    s1, s2 = np.cov(a, rowvar=False), np.cov(b, rowvar=False)
    ev = np.linalg.eigvals(s1 @ s2)
    return float(((m1 - m2) ** 2).sum() + np.trace(s1) + np.trace(s2) - 2 * np.sqrt(np.abs(ev.real)).sum())


def _evaluate(rank, world, num_batches, accu_batches, nan_at=None):
    """The evaluate_gan flow on stand-ins: accumulator fill, one fake set, the real set."""
    accus = [torch.zeros(F), torch.zeros(F), torch.full((), 1e-12)]
    before = [a.clone() for a in accus]
    for i in eval_shard.shard_indices(accu_batches, rank, world):
        act, _ = _transform(_generate(1 + i))
        accus[0] += act.mean(0)
        accus[1] += act.var(0, unbiased=False)
        accus[2] += 1.0
    eval_shard.allreduce_deltas(accus, before, world)
    first = 1 + accu_batches

    def gen(index):
        x = _generate(index)
        if nan_at is not None and index == first + nan_at:
            x[0, 0, 0, 0] = float("nan")
        return x
    timing = {}
    images, act, logits, nan_found = eval_shard.sharded_fake_features(
        gen, _transform, num_batches, first, rank, world, keep_images=True, timing=timing,
        tick=lambda: 0.0)
    total = num_batches * B - 3                      # a test set that is not a multiple of the batch
    g = torch.Generator().manual_seed(7)
    real = torch.rand((total, H, H, 3), generator=g) * 255.0
    chunk = eval_shard.row_chunk(total, world, B)
    lo, hi = eval_shard.row_range(total, rank, world, chunk)
    local = _transform(real[lo:hi])[0] if hi > lo else act.new_zeros((0, F))   # (evaluate_gan's rule)
    real_act = eval_shard.gather_rows(local, total, rank, world, chunk)
    return accus, images, act, logits, nan_found, real_act


def _worker(rank, world, port, num_batches, accu_batches, nan_at, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert eval_shard.rank_world(None) == (rank, world)
        assert eval_shard.rank_world(False) == (0, 1)
        res = _evaluate(rank, world, num_batches, accu_batches, nan_at)
        torch.save(res, os.path.join(out, "rank%d.pt" % rank))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("num_batches", [7, 8])       # odd: rank 1 holds one batch fewer (padding)
def test_two_rank_sharded_evaluation_equals_single_rank(tmp_path, num_batches):
    accu_batches = 5
    mp.spawn(_worker, args=(2, _free_port(), num_batches, accu_batches, None, str(tmp_path)),
             nprocs=2, join=True)
    want = _evaluate(0, 1, num_batches, accu_batches)
    ranks = [torch.load(str(tmp_path / ("rank%d.pt" % r))) for r in range(2)]
    for accus, images, act, logits, nan_found, real_act in ranks:
        assert not nan_found
        assert torch.equal(images, want[1])            # every batch, in single-rank order
        assert torch.equal(act, want[2]) and torch.equal(logits, want[3])
        assert torch.equal(real_act, want[5])
        for a, w in zip(accus, want[0]):               # sums in another order: fp32 rounding only
            assert float((a - w).abs().max()) <= 1e-6 * (1.0 + float(w.abs().max()))
        n = num_batches * B - 3
        assert abs(_fid(act[:n], real_act) - _fid(want[2][:n], want[5])) <= 1e-9
    assert act.shape == (num_batches * B, F) and real_act.shape == (num_batches * B - 3, F)


def test_real_set_smaller_than_the_rank_chunks(tmp_path):
    """the real set is split into whole Inception batches per rank, so a small set leaves
    the LAST ranks without rows (17 rows, batch 8, 4 ranks: chunk 8 -> ranks hold 8 / 8 / 1 / 0 rows).
    The empty rank contributes an empty block to the all-gather instead of hanging the others."""
    world = 4
    assert eval_shard.row_range(17, 3, 4, eval_shard.row_chunk(17, 4, 8)) == (17, 17)
    assert eval_shard.row_range(17, 2, 4, 8) == (16, 17)
    assert eval_shard.row_range(257, 3, 4, eval_shard.row_chunk(257, 4, 64)) == (257, 257)
    # 5 batches on 4 ranks: 37 real rows, chunk 16 -> ranks hold 16 / 16 / 5 / 0 rows
    assert eval_shard.row_range(37, 3, 4, eval_shard.row_chunk(37, 4, B)) == (37, 37)
    mp.spawn(_worker, args=(world, _free_port(), 5, 2, None, str(tmp_path)), nprocs=world, join=True)
    want = _evaluate(0, 1, 5, 2)
    for r in range(world):
        accus, images, act, logits, nan_found, real_act = torch.load(str(tmp_path / ("rank%d.pt" % r)))
        assert torch.equal(act, want[2]) and torch.equal(real_act, want[5])
        assert real_act.shape == (5 * B - 3, F)


def test_nan_on_one_rank_stops_all_ranks(tmp_path):
    # batch 2 of the fake set belongs to rank 0 only; rank 1 must learn about the NaN
    mp.spawn(_worker, args=(2, _free_port(), 6, 2, 2, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(str(tmp_path / ("rank%d.pt" % r)))
        assert res[4] is True and res[1] is None and res[2] is None


def test_shard_indices_cover_every_batch_once():
    for n in (1, 5, 157, 3200):
        for world in (1, 2, 3, 8):
            got = sorted(i for r in range(world) for i in eval_shard.shard_indices(n, r, world))
            assert got == list(range(n))
    assert eval_shard.row_chunk(10000, 8, 64) == 1280 and eval_shard.row_chunk(10000, 1, 64) == 10048
