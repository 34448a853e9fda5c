"""Host-side input pipeline (compare_gan/datasets.py): the synthetic source and the on-disk array
source with the reference's per-dataset transforms restated in numpy."""
import os

import numpy as np
import pytest

from compare_gan_amd import datasets, gin


@pytest.fixture
def data_dir(tmp_path):
    rng = np.random.RandomState(0)
    def put(name, split, n, h, w, c):
        d = tmp_path / name
        d.mkdir(exist_ok=True)
        shape = (n, h, w, c) if c else (n, h, w)
        np.savez(str(d / (split + ".npz")), image=rng.randint(0, 256, size=shape).astype(np.uint8),
                 label=(np.arange(n) % 10).astype(np.int64))
    put("cifar10", "train", 50, 32, 32, 3)
    put("cifar10", "test", 20, 32, 32, 3)
    put("mnist", "train", 30, 28, 28, 0)
    put("celeb_a", "train", 6, 218, 178, 3)
    put("lsun-bedroom", "train", 4, 100, 150, 3)
    put("imagenet_64", "train", 5, 90, 120, 3)
    put("imagenet_64", "test", 5, 90, 120, 3)
    datasets.use_data_dir(str(tmp_path), shuffle_buffer_size=16)
    yield str(tmp_path)
    datasets.use_data_dir(None)
    gin.clear_config()


def test_fake_dataset_is_the_default():
    """datasets.py:136-145: 100 uniform images, labels all one, seeded."""
    ds = datasets.get_dataset("cifar10", seed=547)
    x, y = next(ds.train_batches(8))
    assert x.shape == (8, 32, 32, 3) and x.dtype == np.float32 and 0.0 <= x.min() and x.max() < 1.0
    assert (y == 1).all()
    x2, _ = next(datasets.get_dataset("cifar10", seed=547).train_batches(8))
    assert np.array_equal(x, x2)


def test_array_source_batches(data_dir):
    """repeat -> shuffle buffer -> batch(drop_remainder) (datasets.py:256-281): every batch is
    full, values are image / 255, all examples appear, the stream is seeded."""
    ds = datasets.get_dataset("cifar10", seed=3)
    raw = np.load(os.path.join(data_dir, "cifar10", "train.npz"))
    it = ds.train_batches(8)
    seen = []
    for _ in range(25):                       # 200 examples = 4 epochs of 50
        x, y = next(it)
        assert x.shape == (8, 32, 32, 3) and x.dtype == np.float32 and y.dtype == np.int32
        seen.append((x, y))
    flat = np.concatenate([x for x, _ in seen]).reshape(200, -1)
    ref = raw["image"].reshape(50, -1).astype(np.float32) / 255.0
    # every emitted example is one of the stored ones, and (buffer 16 << 200) all of them show up
    idx = [int(np.argmin(np.abs(ref - row).sum(axis=1))) for row in flat]
    assert all(np.array_equal(ref[i], row) for i, row in zip(idx, flat))
    assert set(idx) == set(range(50))
    assert idx[:50] != list(range(50))        # shuffled
    labels = np.concatenate([y for _, y in seen])
    assert np.array_equal(labels, raw["label"][idx].astype(np.int32))
    again = next(datasets.get_dataset("cifar10", seed=3).train_batches(8))
    assert np.array_equal(again[0], seen[0][0])
    # eval split: unshuffled prefix (datasets.py:283-307)
    ev = ds.eval_images(10)
    test = np.load(os.path.join(data_dir, "cifar10", "test.npz"))["image"][:10]
    assert np.array_equal(ev, test.astype(np.float32) / 255.0)
    with pytest.raises(ValueError):
        ds.eval_images(21)


def test_per_dataset_transforms(data_dir):
    x, y = next(datasets.get_dataset("mnist").train_batches(4))
    assert x.shape == (4, 28, 28, 1)                                   # grey images get a channel
    x, y = next(datasets.get_dataset("celeb_a").train_batches(2))      # datasets.py:388-396
    assert x.shape == (2, 64, 64, 3) and (y == 0).all() and 0.0 <= x.min() and x.max() <= 1.0
    x, y = next(datasets.get_dataset("lsun-bedroom").train_batches(2)) # datasets.py:414-421
    assert x.shape == (2, 128, 128, 3) and (y == 0).all()
    # 100 rows padded to 128: 14 zero rows above and below, 150 columns cropped to the middle 128
    assert (x[:, :14] == 0).all() and (x[:, -14:] == 0).all() and x[:, 14:-14].max() > 0
    x, _ = next(datasets.get_dataset("imagenet_64").train_batches(2))  # distorted crop + resize
    assert x.shape == (2, 64, 64, 3)
    gin.parse_config('eval_imagenet_transform.crop_method = "middle"')
    ev = datasets.get_dataset("imagenet_64").eval_images(2)
    raw = np.load(os.path.join(data_dir, "imagenet_64", "test.npz"))["image"][0]
    mid = raw[:, 15:105].astype(np.float32) / 255.0                    # 90x120 -> middle 90x90
    assert np.allclose(ev[0], datasets.resize_bilinear_tf1(mid, 64, 64), atol=1e-6)
    with pytest.raises(ValueError):
        datasets.transform_imagenet_image(mid, (64, 64, 3), "bogus")


def test_crop_and_resize_primitives():
    img = np.arange(4 * 6, dtype=np.float32).reshape(4, 6, 1)
    # TF1 legacy bilinear (no half-pixel centres): src = dst * in / out
    out = datasets.resize_bilinear_tf1(img, 2, 3)
    assert np.allclose(out[:, :, 0], img[::2, ::2, 0])
    up = datasets.resize_bilinear_tf1(img, 8, 6)
    assert np.allclose(up[0], img[0]) and np.allclose(up[1], 0.5 * (img[0] + img[1]))
    assert np.allclose(up[7], img[3])                                   # clamped at the border
    c = datasets.crop_or_pad(img, 2, 8)
    assert c.shape == (2, 8, 1) and np.array_equal(c[:, 1:7], img[1:3]) and (c[:, 0] == 0).all()


def test_missing_arrays_are_an_error(tmp_path):
    datasets.use_data_dir(str(tmp_path))
    try:
        with pytest.raises(ValueError):
            next(datasets.get_dataset("cifar10").train_batches(4))
    finally:
        datasets.use_data_dir(None)
