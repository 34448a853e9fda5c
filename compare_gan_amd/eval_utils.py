"""Data-related evaluation utilities (reference: compare_gan/eval_utils.py:41-206).

EvalDataSample, NanFoundError, get_real_images, sample_fake_dataset and the batched Inception
transform.  Image sets stay on the GPU as fp32 tensors in [0, 255] (the reference round-trips
them through NumPy: SURVEY.md section 8a row a14)."""
import os
import warnings

import numpy as np
import torch

from compare_gan_amd import gin
from compare_gan_amd import inception as inception_lib

_INCEPTION = {}


class NanFoundError(Exception):
  """Exception thrown, when the Nans are present in the output (eval_utils.py:52-53)."""


class EvalDataSample(object):
  """Images (in [0, 255]) and Inception features of one evaluation set (eval_utils.py:56-84)."""

  def __init__(self, images):
    self.images = images
    self.activations = None
    self.logits = None

  def discard_images(self):
    del self.images
    self.images = None

  def set_inception_features(self, activations, logits):
    self.activations = activations
    self.logits = logits

  def set_num_examples(self, num_examples):
    if self.images is not None:
      assert self.images.shape[0] >= num_examples
      self.images = self.images[:num_examples]
    if self.activations is not None:
      assert self.activations.shape[0] >= num_examples
      self.activations = self.activations[:num_examples]
    if self.logits is not None:
      assert self.logits.shape[0] >= num_examples
      self.logits = self.logits[:num_examples]


def _to_three_channels(images):
  if images.shape[-1] == 1:
    return images.repeat(1, 1, 1, 3) if torch.is_tensor(images) else np.tile(images, [1, 1, 1, 3])
  return images


def get_real_images(dataset, num_examples, split=None, failure_on_insufficient_examples=True,
                    device="cuda:0", rows=None):
  """num_examples real images with values in [0, 255] (eval_utils.py:87-141): the first examples
  of the dataset's eval split -- the on-disk arrays (datasets.use_data_dir) or the synthetic
  source.  Only the default eval split exists here."""
  if split not in (None, "test"):
    raise ValueError("Only the default eval split is available, got %r" % (split,))
  try:
    arrays = dataset.eval_images(num_examples)
  except ValueError:
    if failure_on_insufficient_examples:
      raise
    raise NotImplementedError("partial eval splits are not supported")
  if rows is not None:        # a rank's share of a sharded evaluation (eval_shard.py)
    arrays = arrays[rows[0]:rows[1]]
  images = torch.from_numpy(np.ascontiguousarray(arrays)).to(device)
  return to_eval_images(images)


def to_eval_images(images):
  """[B, H, W, C] in [0, 1] -> three channels in [0, 255]."""
  from compare_gan_amd.hip import kernels as K
  return _to_three_channels(K.scale_f32(images.contiguous(), None, 255.0))


class FakeImageSink(object):
  """One rank's share of a fake evaluation set (eval_shard.sharded_fake_features): every generator
  batch [B, H, W, C] fp32 in [0, 1] goes through ONE HIP pass (cg_scale_count_nan_f32) that writes
  255 * x into its slot of the set's buffer and counts its NaNs -- no concatenation copy, no
  separate isnan / any / scale passes over the 123 MB of a 10k CIFAR set."""

  def __init__(self, num_batches):
    self._n, self._j, self._out, self._nan = int(num_batches), 0, None, None

  def add(self, images):
    from compare_gan_amd.hip import kernels as K
    images = images.contiguous()
    if self._out is None:
      self._out = torch.empty((self._n * images.shape[0],) + tuple(images.shape[1:]),
                              dtype=torch.float32, device=images.device)
      self._nan = torch.zeros(1, dtype=torch.int32, device=images.device)
    b = images.shape[0]
    K.scale_count_nan(images, 255.0, self._out[self._j * b:(self._j + 1) * b], self._nan)
    self._j += 1

  def finish(self):
    if self._j != self._n:
      raise RuntimeError("FakeImageSink: %d of %d batches were added" % (self._j, self._n))
    return _to_three_channels(self._out), bool(int(self._nan.item()) > 0)


def sample_fake_dataset(generate_fn, num_batches):
  """Concatenates `num_batches` generator batches, x255; raises NanFoundError on NaNs
  (eval_utils.py:144-162).  generate_fn() -> [B, H, W, C] fp32 device tensor in [0, 1]."""
  samples = []
  for _ in range(num_batches):
    x = generate_fn()
    samples.append(x)
  fake_images = torch.cat(samples, dim=0)
  if bool(torch.isnan(fake_images).any()):
    raise NanFoundError("Detected NaN in fake images.")
  from compare_gan_amd.hip import kernels as K
  return _to_three_channels(K.scale_f32(fake_images.contiguous(), None, 255.0))


def _load_inception_weight_file(path):
  """{name: tensor} from a .npz / .safetensors / torch file in inception.py's layouts
  (HWIO conv kernels, biases, logits/kernel [2048, 1008]), or from the frozen 2015 Inception
  GraphDef itself (`.pb`: eval_utils.py:41-49 of the reference; compare_gan_amd/graphdef.py decodes
  it and folds its batch norms into the kernels)."""
  if path.endswith(".pb"):
    from compare_gan_amd import graphdef
    return {k: torch.from_numpy(v) for k, v in graphdef.inception_weights_from_graphdef(path).items()}
  if path.endswith(".npz"):
    with np.load(path) as z:
      return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
  if path.endswith(".safetensors"):
    from safetensors.torch import load_file
    return load_file(path)
  return torch.load(path, map_location="cpu")


@gin.configurable("inception_weights")
def inception_weights_path(path=None):
  """File with the trained weights of the 2015 Inception graph (eval_utils.py:41-49 downloads
  the frozen graph; offline it has to be supplied).  gin: inception_weights.path, or the
  CGAMD_INCEPTION_WEIGHTS environment variable."""
  return path or os.environ.get("CGAMD_INCEPTION_WEIGHTS") or None


def get_inception(device):
  key = str(device)
  if key not in _INCEPTION:
    path = inception_weights_path()
    if path:
      _INCEPTION[key] = inception_lib.InceptionV3(device, _load_inception_weight_file(path))
      _INCEPTION[key].synthetic_weights = False
    else:
      warnings.warn(
          "Inception weights are SYNTHETIC (seeded He-normal draws): no trained weight file was "
          "given (gin `inception_weights.path` / CGAMD_INCEPTION_WEIGHTS).  FID / IS / KID values "
          "are self-consistent but NOT comparable with published numbers; results carry "
          "inception_weights_synthetic = 1.", RuntimeWarning, stacklevel=2)
      _INCEPTION[key] = inception_lib.InceptionV3(device)
      _INCEPTION[key].synthetic_weights = True
  return _INCEPTION[key]


def inception_weights_are_synthetic(device="cuda:0"):
  return bool(getattr(get_inception(device), "synthetic_weights", True))


def inception_transform(inputs):
  """[B, H, W, 3] in [0, 255] -> (pool_3 [B, 2048], logits [B, 1008]) (eval_utils.py:165-175)."""
  if float(inputs.min()) < 0.0 or float(inputs.max()) > 255.0:
    raise ValueError("inception_transform expects images in [0, 255]")
  return get_inception(inputs.device).features(inputs)


def inception_transform_np(inputs, batch_size, device="cuda:0"):
  """Batched Inception features and logits for an image set (eval_utils.py:178-206)."""
  dev = inputs.device if torch.is_tensor(inputs) and inputs.is_cuda else torch.device(device)
  return get_inception(dev).transform(inputs, batch_size)
