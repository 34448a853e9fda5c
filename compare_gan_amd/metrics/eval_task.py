"""Abstract evaluation task (reference: compare_gan/metrics/eval_task.py:35-76)."""
import abc


class EvalTask(abc.ABC):
  """Class that describes a single evaluation task (e.g. inception score, FID)."""

  _LABEL = None

  def metric_list(self):
    """Names of the metrics this task generates (eval_task.py:45-54)."""
    return frozenset([self._LABEL])

  @abc.abstractmethod
  def run_after_session(self, fake_dset, real_dset):
    """Runs the task on `EvalDataSample`s of fake / real images (values in 0..255, 3 channels)
    with their Inception features.  Returns {metric name: float}."""
