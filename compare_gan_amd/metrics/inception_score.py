"""Inception score (reference: compare_gan/metrics/inception_score.py:34-48).

tfgan.eval.classifier_score_from_logits (un-vendored) = exp(mean_i KL(p(y|x_i) || p(y))) in float64;
computed by the fp64 HIP kernel cg_inception_score_f64."""
import numpy as np
import torch

from compare_gan_amd.hip import kernels as K
from compare_gan_amd.metrics import eval_task


def classifier_score_from_logits(logits, device="cuda:0"):
  if not torch.is_tensor(logits):
    logits = torch.from_numpy(np.ascontiguousarray(logits, dtype=np.float32))
  logits = logits.to(device=device, dtype=torch.float32).contiguous()
  return float(K.inception_score_f64(logits).cpu())


class InceptionScoreTask(eval_task.EvalTask):
  """Task that computes inception score for the generated images."""

  _LABEL = "inception_score"

  def run_after_session(self, fake_dset, real_dset):
    del real_dset
    dev = fake_dset.logits.device if torch.is_tensor(fake_dset.logits) and \
        fake_dset.logits.is_cuda else "cuda:0"
    return {self._LABEL: classifier_score_from_logits(fake_dset.logits, dev)}
