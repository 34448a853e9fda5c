"""Generator / discriminator plugin interfaces (reference: architectures/abstract_arch.py:29-146).

Same contract: G(z, y, is_training) -> images [B,H,W,C] fp32 in [0,1]; D(x, y, is_training) ->
(prob [B,1], logits [B,1], features); gin configurables "G" and "D"; `self.batch_norm(...)`
forwards only the keyword arguments the bound batch_norm_fn accepts.  Variables are created in the
active arch_ops.VariableStore under scope `self.name` on the first call and reused afterwards.
"""
import abc

from compare_gan_amd import gin
from compare_gan_amd import utils
from compare_gan_amd.architectures import arch_ops as ops


class _Module(abc.ABC):
  """Base class for architectures."""

  def __init__(self, name):
    self._name = name

  @property
  def name(self):
    return self._name

  @property
  def trainable_variables(self):
    """(name, tensor) pairs whose name contains the module name (abstract_arch.py:43-45)."""
    return ops.current_store().trainable_variables(self._name)

  def _norm(self, inputs, relu, kwargs):
    """abstract_arch.py:76-83 plus the fused ReLU: returns a tensor, or a pending Act."""
    if self._batch_norm_fn is None:
      return ops.Act(inputs, 0.0) if relu else inputs
    args = dict(kwargs)
    args["inputs"] = inputs
    if "use_sn" not in args:
      args["use_sn"] = self._spectral_norm
    if relu and utils._has_arg(self._batch_norm_fn, "relu"):   # pylint: disable=protected-access
      args["relu"] = True
      return utils.call_with_accepted_args(self._batch_norm_fn, **args)
    out = utils.call_with_accepted_args(self._batch_norm_fn, **args)
    return ops.Act(out, 0.0) if relu else out

  def batch_norm(self, inputs, **kwargs):
    return self._norm(inputs, False, kwargs)

  def batch_norm_relu(self, inputs, **kwargs):
    """tf.nn.relu(self.batch_norm(inputs, ...)) as one kernel (or a pending ReLU without BN)."""
    return self._norm(inputs, True, kwargs)


@gin.configurable("G", blacklist=["name", "image_shape"])
class AbstractGenerator(_Module):
  """Interface for generator architectures."""

  def __init__(self, name="generator", image_shape=None, batch_norm_fn=None,
               spectral_norm=False):
    super(AbstractGenerator, self).__init__(name=name)
    self._image_shape = image_shape
    self._batch_norm_fn = batch_norm_fn
    self._spectral_norm = spectral_norm

  def __call__(self, z, y, is_training, reuse=None):
    del reuse  # AUTO_REUSE semantics: the store creates on first use and reuses afterwards
    import torch
    ops.prepare_module(self.name)
    # no-gradient training-mode forward (discriminator sub-steps): convolutions feed the batch norms
    # that follow them with the partial sums of their outputs (arch_ops._conv_call)
    old = ops._EMIT_BN_STATS[0]   # pylint: disable=protected-access
    ops._EMIT_BN_STATS[0] = bool(is_training and not torch.is_grad_enabled() and   # pylint: disable=protected-access
                                 self._batch_norm_fn is not None and not z.is_meta)
    try:
      with ops.variable_scope(self.name):
        return ops.as_tensor(self.apply(z=z, y=y, is_training=is_training))
    finally:
      ops._EMIT_BN_STATS[0] = old   # pylint: disable=protected-access

  @abc.abstractmethod
  def apply(self, z, y, is_training):
    """z [B, z_dim] fp32, y [B, num_classes] one-hot (or None) -> images [B] + image_shape."""


@gin.configurable("D", blacklist=["name"])
class AbstractDiscriminator(_Module):
  """Interface for discriminator architectures."""

  def __init__(self, name="discriminator", batch_norm_fn=None, layer_norm=False,
               spectral_norm=False):
    super(AbstractDiscriminator, self).__init__(name=name)
    self._batch_norm_fn = batch_norm_fn
    self._layer_norm = layer_norm
    self._spectral_norm = spectral_norm

  def __call__(self, x, y, is_training, reuse=None):
    del reuse
    ops.prepare_module(self.name)
    with ops.variable_scope(self.name):
      return self.apply(x=x, y=y, is_training=is_training)

  @abc.abstractmethod
  def apply(self, x, y, is_training):
    """x [B,H,W,C] (bf16 staged images), y one-hot or None -> (prob, logits, features)."""
