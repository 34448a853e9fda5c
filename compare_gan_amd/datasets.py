"""Dataset descriptors, the synthetic ("fake") input pipeline and an on-disk array source.

Reference: compare_gan/datasets.py:66-648.  The reference reads TFDS TFRecords; there is no TFDS
(nor TensorFlow, nor network) here, so two sources exist:

* the `--data_fake_dataset` pipeline (datasets.py:136-145: 100 uniform-random images, labels all
  1, seeded by the dataset seed) -- what BASELINE.json's synthetic-batch measurements use and the
  default;
* `use_data_dir(path)` (or CGAMD_DATA_DIR): `<path>/<dataset name>/<split>.npz` with `image`
  (uint8 [N,h,w,c], decoded but NOT yet cropped / resized) and `label` (int [N]) -- the decoded
  content of the TFDS records.  From there on the reference's pipeline is restated in numpy:
  `_parse_fn` (datasets.py:225-227,388-396,414-421: /255, CelebA 160-crop + 64-resize, LSUN
  128 crop-or-pad, constant labels), the ImageNet crops (datasets.py:424-497: "middle" / "random" /
  "none" + TF1 bilinear resize; "distorted" draws its box from a seeded numpy stream instead of
  tf.image.sample_distorted_bounding_box), then repeat -> shuffle buffer -> batch with
  drop_remainder (datasets.py:256-281) and the unshuffled eval split (datasets.py:283-307).
  The eval arrays are always `<path>/<dataset name>/test.npz` -- for ImageNet that file holds the
  VALIDATION records (datasets.py:514), for LSUN the 1 % tail of the training shards (:413-418).
  When `<path>` is a TFDS data dir instead (the reference's --tfds_data_dir: record shards under
  <path>/<tfds name>/...), compare_gan_amd/tfrecord.py decodes the TFRecord framing, the
  tf.train.Example messages and the PNG / JPEG images (Pillow) and feeds the same pipeline.
"""
import os

import numpy as np

from compare_gan_amd import gin

_SOURCE = {"dir": os.environ.get("CGAMD_DATA_DIR") or None, "shuffle_buffer": 10000}


def use_data_dir(path, shuffle_buffer_size=10000):
  """Switches every dataset created afterwards from the synthetic pipeline to the arrays under
  `path` (None: back to synthetic) -- the counterpart of --data_fake_dataset=False with
  --tfds_data_dir and --data_shuffle_buffer_size (datasets.py:40-60)."""
  _SOURCE["dir"] = path
  _SOURCE["shuffle_buffer"] = int(shuffle_buffer_size)


def resize_bilinear_tf1(image, out_h, out_w):
  """tf.image.resize_images(image, [out_h, out_w]) of TF1 (bilinear, align_corners=False, no
  half-pixel centres): src = dst * in / out.  image [h,w,c] float."""
  h, w = image.shape[0], image.shape[1]
  ys = np.arange(out_h, dtype=np.float64) * (float(h) / out_h)
  xs = np.arange(out_w, dtype=np.float64) * (float(w) / out_w)
  y0 = np.minimum(np.floor(ys).astype(np.int64), h - 1)
  x0 = np.minimum(np.floor(xs).astype(np.int64), w - 1)
  y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
  fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
  img = image.astype(np.float64)
  top = img[y0][:, x0] * (1 - fx) + img[y0][:, x1] * fx
  bot = img[y1][:, x0] * (1 - fx) + img[y1][:, x1] * fx
  return (top * (1 - fy) + bot * fy).astype(np.float32)


def crop_or_pad(image, th, tw):
  """tf.image.resize_image_with_crop_or_pad: central crop and / or zero padding to [th, tw]."""
  h, w = image.shape[0], image.shape[1]
  if h > th:
    off = (h - th) // 2
    image = image[off:off + th]
  if w > tw:
    off = (w - tw) // 2
    image = image[:, off:off + tw]
  h, w = image.shape[0], image.shape[1]
  if h < th or w < tw:
    out = np.zeros((th, tw) + image.shape[2:], dtype=image.dtype)
    oy, ox = (th - h) // 2, (tw - w) // 2
    out[oy:oy + h, ox:ox + w] = image
    image = out
  return image


def transform_imagenet_image(image, target_image_shape, crop_method, rng=None):
  """datasets.py:424-470 (_transform_imagnet_image) on one decoded image [h,w,3] in [0,1]."""
  h, w = image.shape[0], image.shape[1]
  if crop_method == "distorted":
    # square box (aspect_ratio_range [1,1]) covering 50-100 % of the image area; the reference
    # samples it with tf.image.sample_distorted_bounding_box
    area = rng.uniform(0.5, 1.0) * h * w
    size = int(min(np.sqrt(area), h, w))
    by, bx = rng.randint(0, h - size + 1), rng.randint(0, w - size + 1)
    image = image[by:by + size, bx:bx + size]
  elif crop_method == "random":
    size = min(h, w)
    u = rng.uniform(0.0, 1.0, size=2)
    by, bx = int((h - size) * u[0]), int((w - size) * u[1])
    image = image[by:by + size, bx:bx + size]
  elif crop_method == "middle":
    size = min(h, w)
    by, bx = int((h - size) / 2.0), int((w - size) / 2.0)
    image = image[by:by + size, bx:bx + size]
  elif crop_method != "none":
    raise ValueError("Unsupported crop method: {}".format(crop_method))
  return resize_bilinear_tf1(image, target_image_shape[0], target_image_shape[1])


@gin.configurable("train_imagenet_transform", whitelist=["crop_method"])
def _train_imagenet_transform(image, target_image_shape, rng, crop_method="distorted"):
  return transform_imagenet_image(image, target_image_shape, crop_method, rng)


@gin.configurable("eval_imagenet_transform", whitelist=["crop_method"])
def _eval_imagenet_transform(image, target_image_shape, rng, crop_method="middle"):
  return transform_imagenet_image(image, target_image_shape, crop_method, rng)


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
    self._fake = fake_dataset and _SOURCE["dir"] is None
    self._data_dir = _SOURCE["dir"]
    self._shuffle_buffer = _SOURCE["shuffle_buffer"]

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

  # -- on-disk arrays (see the module docstring) ---------------------------------------------------
  def _load_arrays(self, split, max_examples=None):
    path = os.path.join(self._data_dir, self._name, split + ".npz")
    if not os.path.exists(path):
      # a TFDS data dir (the reference's --tfds_data_dir): decode the records themselves (the
      # evaluation only decodes the examples it takes)
      from compare_gan_amd import tfrecord
      if tfrecord.has_split(self._data_dir, self._name, split == "train"):
        return tfrecord.load_split(self._data_dir, self._name, split == "train",
                                   max_examples=max_examples)
      raise ValueError("Dataset %s: no %s (expected arrays `image` uint8 [N,h,w,c] and `label`)" % (
          self._name, path))
    with np.load(path, allow_pickle=False) as f:
      images, labels = f["image"], f["label"]
    if len(images) != len(labels):
      raise ValueError("%s: %d images but %d labels" % (path, len(images), len(labels)))
    return images, np.asarray(labels).astype(np.int32)

  def _parse(self, image, label, training, rng):
    """_parse_fn + the split's transform of this dataset -> ([H,W,C] fp32 in [0,1], int32)."""
    image = np.asarray(image)
    if image.ndim == 2:
      image = image[:, :, None]
    if self._name == "celeb_a":                                  # datasets.py:388-396
      image = crop_or_pad(image, 160, 160)
      image = resize_bilinear_tf1(image.astype(np.float32), 64, 64) / 255.0
      label = 0
    elif self._name == "lsun-bedroom":                           # datasets.py:414-421
      image = crop_or_pad(image, 128, 128).astype(np.float32) / 255.0
      label = 0
    elif self._name.startswith("imagenet_"):                     # datasets.py:500-532
      fn = _train_imagenet_transform if training else _eval_imagenet_transform
      image = fn(image.astype(np.float32) / 255.0, self.image_shape, rng)
    else:                                                        # datasets.py:225-227
      image = image.astype(np.float32) / 255.0
    if tuple(image.shape) != self.image_shape:
      raise ValueError("%s: example of shape %s, expected %s" % (
          self._name, tuple(image.shape), self.image_shape))
    return image.astype(np.float32), np.int32(label)

  def _real_train_batches(self, batch_size, seed):
    """load -> repeat -> transform -> shuffle(buffer, seed) -> batch(drop_remainder)
    (datasets.py:256-281); the shuffle is tf.data's buffer algorithm (fill the buffer, then
    emit a uniformly chosen slot and refill it) on a numpy stream."""
    images, labels = self._load_arrays("train")
    rng = np.random.RandomState(seed)
    n = len(images)

    def examples():
      while True:                                                # ds.repeat()
        for i in range(n):
          yield self._parse(images[i], labels[i], True, rng)

    stream = examples()
    size = max(1, min(self._shuffle_buffer, 1 << 20))
    buf = [next(stream) for _ in range(size)]
    while True:
      xs, ys = [], []
      for _ in range(batch_size):
        j = rng.randint(0, len(buf))
        x, y = buf[j]
        buf[j] = next(stream)
        xs.append(x)
        ys.append(y)
      yield np.stack(xs), np.asarray(ys, dtype=np.int32)

  def train_batches(self, batch_size, seed=None):
    """Infinite iterator of (images [B,H,W,C] fp32, labels [B] int32): repeat + shuffle + batch."""
    if not self._fake:
      for batch in self._real_train_batches(batch_size, self._seed if seed is None else seed):
        yield batch
      return
    images, labels = self._make_fake_dataset("train")
    rng = np.random.RandomState(self._seed if seed is None else seed)
    n = images.shape[0]
    while True:
      idx = rng.randint(0, n, size=batch_size)
      yield images[idx], labels[idx]

  def eval_images(self, num_examples):
    """[num_examples,H,W,C] fp32 in [0,1] from the eval split: the first examples of the
    unshuffled "test" arrays (datasets.py:283-307), or seeded synthetic images."""
    if not self._fake:
      images, labels = self._load_arrays("test", max_examples=num_examples)
      if len(images) < num_examples:
        raise ValueError("%s: %d eval examples requested, %d on disk" % (
            self._name, num_examples, len(images)))
      rng = np.random.RandomState(self._seed)
      return np.stack([self._parse(images[i], labels[i], False, rng)[0]
                       for i in range(num_examples)])
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
