"""The reference's training-step smoke matrix (gans/modular_gan_test.py:40-181) on the MI355X:
one step at batch 2 on cifar10 for every architecture of TEST_ARCHITECTURES (hinge, no penalty),
every loss and every penalty on resnet_cifar_arch, and the step counters for disc_iters 1..3 under
the unrolled and the not-unrolled step.  Arithmetic parity of each piece is the subject of the
other GPU test files; this one asserts that every combination RUNS on the HIP path with finite
losses and moves the weights."""
import numpy as np
import pytest
import torch

from tests import gan_util as U

pytestmark = pytest.mark.gpu

TEST_ARCHITECTURES = ["infogan_arch", "dcgan_arch", "resnet_cifar_arch", "sndcgan_arch",
                      "resnet5_arch"]
TEST_LOSSES = ["non_saturating", "wasserstein", "least_squares", "hinge"]
TEST_PENALTIES = ["no_penalty", "dragan_penalty", "wgangp_penalty", "l2_penalty"]


def _single_training_step(dev, architecture, loss_fn, penalty_fn, extra=()):
    bind = ['options.architecture = "%s"' % architecture, "loss.fn = @%s" % loss_fn,
            "penalty.fn = @%s" % penalty_fn, "options.lamba = 1", "options.disc_iters = 1"]
    gan, options, ds = U.build_product("resnet_cifar10.gin", 2, dev, seed=1,
                                       bindings=bind + list(extra))
    nsub = options["disc_iters"] + 1
    before = {n: v.detach().clone() for n, v in gan.store.trainable_variables()}
    images, labels = next(ds.train_batches(2 * nsub, seed=3))
    out = gan.train_step(torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev))
    assert np.isfinite(float(out["g_loss"])), (architecture, loss_fn, penalty_fn)
    assert all(np.isfinite(float(d)) for d in out["d_losses"])
    moved = [n for n, v in gan.store.trainable_variables()
             if n.endswith("/kernel") and not torch.equal(v, before[n])]
    kernels = [n for n, _ in gan.store.trainable_variables() if n.endswith("/kernel")]
    assert len(moved) == len(kernels), sorted(set(kernels) - set(moved))[:4]
    assert int(gan.global_step.item()) == 1
    assert int(gan.global_step_disc.item()) == options["disc_iters"]
    return gan


@pytest.mark.parametrize("architecture", TEST_ARCHITECTURES)
def test_single_training_step_architectures(dev, architecture):
    _single_training_step(dev, architecture, "hinge", "no_penalty")


@pytest.mark.parametrize("loss_fn", TEST_LOSSES)
def test_single_training_step_losses(dev, loss_fn):
    _single_training_step(dev, "resnet_cifar_arch", loss_fn, "no_penalty")


@pytest.mark.parametrize("penalty_fn", TEST_PENALTIES)
def test_single_training_step_penalties(dev, penalty_fn):
    _single_training_step(dev, "resnet_cifar_arch", "hinge", penalty_fn)


@pytest.mark.parametrize("unrolled", [True, False], ids=["unrolled", "not-unrolled"])
@pytest.mark.parametrize("disc_iters", [1, 2, 3])
def test_disc_iters_is_used_correctly(dev, disc_iters, unrolled):
    """modular_gan_test.py:141-177: three generator steps later global_step_disc == 3 * disc_iters
    in both graph forms (the not-unrolled graph needs disc_iters calls per generator step)."""
    bind = ["options.disc_iters = %d" % disc_iters, "ModularGAN.g_use_ema = True"]
    gan, options, ds = U.build_product("resnet_cifar10.gin", 2, dev, seed=1, bindings=bind)
    assert gan.unroll_graph(use_tpu=unrolled) == unrolled
    calls = 0
    if unrolled:
        it = ds.train_batches(2 * (disc_iters + 1), seed=3)
        for _ in range(3):
            images, labels = next(it)
            gan.train_step(torch.from_numpy(images).to(dev), torch.from_numpy(labels).to(dev))
            calls += 1
    else:
        it = ds.train_batches(2, seed=3)
        while int(gan.global_step.item()) < 3:
            images, labels = next(it)
            gan.train_step_not_unrolled(torch.from_numpy(images).to(dev),
                                        torch.from_numpy(labels).to(dev))
            calls += 1
            assert calls <= 3 * disc_iters
        assert calls == 3 * disc_iters
    assert int(gan.global_step.item()) == 3
    assert int(gan.global_step_disc.item()) == 3 * disc_iters
    # EMA shadows exist for exactly the generator's trainable variables
    # (modular_gan_test.py:124-137)
    sd = gan.state_dict()
    ema = sorted(k for k in sd if k.endswith("/ExponentialMovingAverage"))
    assert ema == sorted(n + "/ExponentialMovingAverage"
                         for n, _ in gan.store.trainable_variables("generator"))
