"""Dataset descriptors and the synthetic ("fake") input pipeline.

Reference: compare_gan/datasets.py:66-648.  The reference reads TFDS; here (no network, no TFDS)
only what the hot path needs is kept: the name -> (resolution, colors, num_classes,
eval_test_samples) table (datasets.py:332-640) and the `--data_fake_dataset` pipeline
(datasets.py:136-145: 100 uniform-random images, labels all 1, seeded by the dataset seed), which
is also what BASELINE.json's synthetic-batch measurements use.  Real-data loading is a "next" row
(SURVEY.md section 8f).
"""
import numpy as np

from compare_gan_amd import gin


class ImageDatasetV2(object):
  """Interface for image datasets (datasets.py:66-131)."""

  def __init__(self, name, resolution, colors, num_classes, eval_test_samples, seed,
               fake_dataset=True):
    self._name = name
    self._resolution = resolution
    self._colors = colors
    self._num_classes = num_classes
    self._eval_test_sample = eval_test_samples
    self._seed = seed
    self._fake = fake_dataset

  @property
  def name(self):
    return self._name

  @property
  def num_classes(self):
    return self._num_classes

  @property
  def eval_test_samples(self):
    """Number of examples in the "test" split (100 for the fake dataset, datasets.py:125-129)."""
    return self._eval_test_sample

  @property
  def image_shape(self):
    return (self._resolution, self._resolution, self._colors)

  # -- synthetic pipeline -------------------------------------------------------------------------
  def _make_fake_dataset(self, split):
    """100 uniform-random images in [0,1), labels all ones (datasets.py:136-145)."""
    del split
    rng = np.random.RandomState(self._seed)
    images = rng.uniform(size=(100,) + self.image_shape).astype(np.float32)
    labels = np.ones((100,), dtype=np.int32)
    return images, labels

  def train_batches(self, batch_size, seed=None):
    """Infinite iterator of (images [B,H,W,C] fp32, labels [B] int32): repeat + shuffle + batch."""
    if not self._fake:
      raise NotImplementedError("Only the synthetic pipeline is available offline.")
    images, labels = self._make_fake_dataset("train")
    rng = np.random.RandomState(self._seed if seed is None else seed)
    n = images.shape[0]
    while True:
      idx = rng.randint(0, n, size=batch_size)
      yield images[idx], labels[idx]

  def eval_images(self, num_examples):
    """[num_examples,H,W,C] fp32 in [0,1] from the eval split (synthetic, seeded)."""
    rng = np.random.RandomState(self._seed + 1)
    return rng.uniform(size=(num_examples,) + self.image_shape).astype(np.float32)


def _mk(name, resolution, colors, num_classes, eval_test_samples):
  def ctor(seed):
    return ImageDatasetV2(name, resolution, colors, num_classes, eval_test_samples, seed)
  return ctor


DATASETS = {
    "celeb_a": _mk("celeb_a", 64, 3, None, 10000),
    "cifar10": _mk("cifar10", 32, 3, 10, 10000),
    "fashion-mnist": _mk("fashion_mnist", 28, 1, 10, 10000),
    "lsun-bedroom": _mk("lsun-bedroom", 128, 3, None, 30000),
    "mnist": _mk("mnist", 28, 1, 10, 10000),
    "celeb_a_hq_128": _mk("celeb_a_hq_128", 128, 3, None, 3000),
    "imagenet_64": _mk("imagenet_64", 64, 3, 1000, 50000),
    "imagenet_128": _mk("imagenet_128", 128, 3, 1000, 50000),
    "imagenet_256": _mk("imagenet_256", 256, 3, 1000, 50000),
    "imagenet_512": _mk("imagenet_512", 512, 3, 1000, 50000),
}


@gin.configurable("dataset")
def get_dataset(name, seed=547):
  """Instantiates a data set and sets the random seed (datasets.py:643-648)."""
  if name not in DATASETS:
    raise ValueError("Dataset %s is not available." % name)
  return DATASETS[name](seed=seed)
