"""Interface of a GAN trainer (reference: gans/abstract_gan.py:29-92).

The reference's AbstractGAN is a TPUEstimator adapter (as_estimator / input_fn / model_fn /
as_module_spec).  On MI355X there is no Estimator: a GAN object owns its variables and exposes the
same responsibilities as direct calls -- build() (graph construction), train_step() (model_fn in
TRAIN mode), generate() (the exported "gen" module) and state_dict() (the checkpoint)."""
import abc


class AbstractGAN(abc.ABC):
  """Interface for GAN models."""

  def __init__(self, dataset, parameters, model_dir):
    self._dataset = dataset
    self._parameters = parameters
    self._model_dir = model_dir

  @abc.abstractmethod
  def build(self, batch_size, device, seed=0):
    """Creates variables / optimiser state for sub-step batches of `batch_size` on `device`."""

  @abc.abstractmethod
  def train_step(self, images, labels):
    """Runs disc_iters D updates and one G update (abstract_gan.py:61-82 model_fn, TRAIN)."""

  @abc.abstractmethod
  def generate(self, z, labels=None, use_ema=None):
    """Inference-mode generator (abstract_gan.py:49-59 as_module_spec 'gen' signature)."""
