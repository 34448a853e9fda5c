"""Eval path on the MI355X: Inception features, FID, inception
score and evaluate_gan against the CPU oracle.

Tolerances: FID / IS statistics run in fp64 on both sides -> relative 1e-6 on FID given identical
activations (the reference's own pin is 89.091 +- 1e-4, metrics/fid_score_test.py:31-40);
Inception activations cross 94 bf16-stored convolutions -> cosine >= 0.999 and rel-L2 <= 0.05
against the fp64 oracle with emulated bf16 storage."""
import numpy as np
import pytest
import torch

from oracle import fid as ofid
from tests import gan_util as U

pytestmark = pytest.mark.gpu


def test_fid_matches_reference_golden(dev):
    from compare_gan_amd.metrics import fid_score
    real = np.ones((100, 2), dtype=np.float32)
    real[:50, 0] = 2
    gen = np.ones((100, 2), dtype=np.float32) * 9
    gen[50:, 0] = 2
    got = fid_score.compute_fid_from_activations(real, gen)   # the reference test's argument order
    assert abs(got - 89.091) <= 1e-4, got


@pytest.mark.parametrize("n,d", [(300, 64), (64, 96), (2000, 256)])
def test_fid_matches_oracle(dev, n, d):
    """Includes the rank-deficient case n < d (covariance with zero eigenvalues)."""
    from compare_gan_amd.metrics import fid_score
    rng = np.random.RandomState(n + d)
    a = (rng.randn(n, d) * rng.rand(d) * 2 + rng.randn(d)).astype(np.float32)
    b = (rng.randn(n, d) * rng.rand(d) * 3 + 0.5).astype(np.float32)
    ref = ofid.frechet_distance(a, b)
    got = fid_score.frechet_distance(a, b, device=dev)
    assert abs(got - ref) <= 1e-6 * abs(ref) + 1e-8, (got, ref)
    same = fid_score.frechet_distance(a, a, device=dev)
    assert abs(same) <= 1e-6 * float(np.trace(np.cov(a.T))), same


def test_inception_score_matches_oracle(dev):
    from compare_gan_amd.metrics import inception_score
    rng = np.random.RandomState(3)
    logits = (rng.randn(512, 1008) * 2).astype(np.float32)
    got = inception_score.classifier_score_from_logits(logits, dev)
    ref = ofid.classifier_score_from_logits(logits)
    assert abs(got - ref) <= 1e-9 * ref


def test_inception_features_match_oracle(dev):
    from compare_gan_amd import inception
    from oracle import inception as oinc
    weights = inception.make_weights(seed=7)
    net = inception.InceptionV3(dev, weights=weights)
    rng = np.random.RandomState(5)
    images = (rng.rand(2, 32, 32, 3) * 255).astype(np.float32)
    pool3, logits = net.features(torch.from_numpy(images).to(dev))
    assert tuple(pool3.shape) == (2, 2048) and tuple(logits.shape) == (2, 1008)
    p_ref, l_ref = oinc.features(inception.SPEC, weights, images, emulate_bf16=True)
    assert U.cosine(pool3, p_ref) >= 0.999 and U.rel_l2(pool3, p_ref) <= 0.05, (
        U.cosine(pool3, p_ref), U.rel_l2(pool3, p_ref))
    assert U.cosine(logits, l_ref) >= 0.999 and U.rel_l2(logits, l_ref) <= 0.05
    # batched transform == per-batch features, ragged last batch
    images3 = (rng.rand(5, 32, 32, 3) * 255).astype(np.float32)
    f_all, _ = net.transform(images3, batch_size=2)
    f_one, _ = net.features(torch.from_numpy(images3[4:5]).to(dev))
    assert tuple(f_all.shape) == (5, 2048)
    assert U.rel_l2(f_all[4:5], f_one) <= 1e-6


def test_evaluate_gan_small(dev):
    """evaluate_gan end to end on a freshly initialised ResNet-CIFAR GAN with 128 test examples:
    keys and aggregation of eval_gan_lib.py:196-212, determinism of the evaluation noise."""
    from compare_gan_amd import eval_gan_lib
    from compare_gan_amd.metrics import fid_score, inception_score
    gan, options, dataset = U.build_product("resnet_cifar10.gin", 8, dev, seed=3)
    tasks = [inception_score.InceptionScoreTask(), fid_score.FIDScoreTask()]
    r1 = eval_gan_lib.evaluate_gan(gan, tasks, num_averaging_runs=2, num_test_examples=128)
    for key in ("inception_score", "fid_score"):
        for suffix in ("_mean", "_std", "_list"):
            assert key + suffix in r1
    assert r1["inception_score_mean"] >= 1.0 - 1e-6 and np.isfinite(r1["fid_score_mean"])
    assert len(r1["fid_score_list"].split("_")) == 2
    r2 = eval_gan_lib.evaluate_gan(gan, tasks, num_averaging_runs=2, num_test_examples=128)
    assert r1["fid_score_list"] == r2["fid_score_list"]      # same latents for each evaluation
    assert eval_gan_lib.evaluate_gan(gan, [], 1, num_test_examples=64) is None


def test_nan_detection(dev):
    from compare_gan_amd import eval_utils
    def bad():
        return torch.full((4, 8, 8, 3), float("nan"), device=dev)
    with pytest.raises(eval_utils.NanFoundError):
        eval_utils.sample_fake_dataset(bad, 2)
