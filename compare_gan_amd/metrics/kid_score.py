"""Kernel Inception Distance (reference: compare_gan/metrics/kid_score.py:31-149).

Unbiased block estimator with the cubic polynomial kernel k(x, y) = (x.y / d + 1)^3 on Inception
pool_3 activations: the activations are split into ceil(max(n_real, n_gen) / 1024) blocks, each
block contributes
   -2 mean(k_rg) + (sum(k_rr) - tr(k_rr)) / (m (m - 1)) + (sum(k_gg) - tr(k_gg)) / (n (n - 1))
(with n := m, the real block's size, exactly as kid_score.py:121-136 computes it) and the score is
the mean over blocks.  The three Gram matrices of a block are fp64 GEMMs on the device
(cg_gemm_f64), the kernel polynomial and its sum / trace one reduction kernel
(cg_poly3_kernel_sums_f64); the per-block scalar assembly happens on the host.  The reference
computes in the activations' dtype (fp32, kid_score.py:88-95); fp64 here only tightens it.
"""
import math

import numpy as np
import torch

from compare_gan_amd.hip import kernels as K
from compare_gan_amd.metrics import eval_task


def _bins(n_real, n_gen, max_batch_size):
  """Block sizes of kid_score.py:97-104 (including its use of bins_r[0] for both arrays)."""
  n_bins = int(math.ceil(max(n_real, n_gen) / max_batch_size))
  bins_r = np.full(n_bins, int(math.ceil(n_real / n_bins)))
  bins_g = np.full(n_bins, int(math.ceil(n_gen / n_bins)))
  bins_r[:(n_bins * bins_r[0]) - n_real] -= 1
  bins_g[:(n_bins * bins_r[0]) - n_gen] -= 1
  assert bins_r.min() >= 2
  assert bins_g.min() >= 2
  return np.r_[0, np.cumsum(bins_r)], np.r_[0, np.cumsum(bins_g)]


def kid(fake_activations, real_activations, max_batch_size=1024, dtype=None, return_stderr=False,
        device="cuda:0"):
  """Unbiased estimator of the Kernel Inception Distance (kid_score.py:44-149) -> float
  (and the standard error of the block estimates if return_stderr; nan for fewer than 5 blocks)."""
  del dtype
  def dev(t):
    t = t if torch.is_tensor(t) else torch.from_numpy(np.ascontiguousarray(t))
    return t.to(device=device, dtype=torch.float64).contiguous()
  real, fake = dev(real_activations), dev(fake_activations)
  if real.dim() != 2 or fake.dim() != 2:
    raise ValueError("activations must have rank 2")
  n_real, dim = real.shape
  n_gen, dim2 = fake.shape
  assert dim2 == dim
  inds_r, inds_g = _bins(n_real, n_gen, max_batch_size)
  ests = []
  for i in range(len(inds_r) - 1):
    r = real[int(inds_r[i]):int(inds_r[i + 1])].contiguous()
    g = fake[int(inds_g[i]):int(inds_g[i + 1])].contiguous()
    m = float(r.shape[0])
    n = m   # kid_score.py:126: `n = tf.cast(r_e - r_s, dtype)`
    s_rr = K.poly3_kernel_sums_f64(K.gemm_f64(r, r, tb=True), dim).cpu().numpy()
    s_rg = K.poly3_kernel_sums_f64(K.gemm_f64(r, g, tb=True), dim).cpu().numpy()
    s_gg = K.poly3_kernel_sums_f64(K.gemm_f64(g, g, tb=True), dim).cpu().numpy()
    mean_rg = s_rg[0] / (r.shape[0] * g.shape[0])
    ests.append(-2.0 * mean_rg + (s_rr[0] - s_rr[1]) / (m * (m - 1)) +
                (s_gg[0] - s_gg[1]) / (n * (n - 1)))
  ests = np.asarray(ests, dtype=np.float64)
  if return_stderr:
    if len(ests) < 5:
      return float(ests.mean()), float("nan")
    return float(ests.mean()), float(np.sqrt(ests.var() / len(ests)))
  return float(ests.mean())


class KIDScoreTask(eval_task.EvalTask):
  """Evaluation task for the KID score (kid_score.py:31-40)."""

  _LABEL = "kid_score"

  def run_after_session(self, fake_dset, real_dset):
    acts = fake_dset.activations
    device = acts.device if torch.is_tensor(acts) and acts.is_cuda else torch.device("cuda:0")
    return {self._LABEL: kid(fake_dset.activations, real_dset.activations, device=device)}
