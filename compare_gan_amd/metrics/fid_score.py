"""Frechet Inception Distance (reference: compare_gan/metrics/fid_score.py:39-75).

The reference delegates the arithmetic to tensorflow_gan's
frechet_classifier_distance_from_activations (un-vendored; restated in SURVEY.md section 8c):
float64 throughout, m = mean, sigma = Xc^T Xc / (n - 1),
  fid = tr(sigma) + tr(sigma_v) - 2 tr sqrt(sqrt(sigma) sigma_v sqrt(sigma)) + |m - m_v|^2
with the symmetric square root U diag(where(s < 1e-10, s, sqrt(s))) V^T taken from an SVD.  Here the
O(n d^2) and O(d^3) parts run as fp64 HIP kernels (cg_mean_cov_f64, cg_syevj_f64, cg_gemm_f64,
cg_rowscale_f64); for a symmetric matrix the SVD is its eigen-decomposition with
s = |lambda|, V = sign(lambda) U, so the same function of the spectrum is applied to the Jacobi
eigenvalues (cg_spectral_sqrt_f64); the O(d) scalar assembly is one more kernel (cg_fid_combine_f64).
"""
import numpy as np
import torch

from compare_gan_amd.hip import kernels as K
from compare_gan_amd.metrics import eval_task

# Special value returned when FID code returned exception (fid_score.py:34).
FID_CODE_FAILED = 4242.0
_EPS = 1e-10
# Jacobi sweeps run until a whole sweep finds every pair of rows orthogonal to _TOL (the device-side
# flag turns the remaining launches into no-ops), at most _SWEEPS: ~10 for the full-rank 2048 x 2048
# covariances of FID-10k, ~30 for rank-deficient ones (fewer samples than features).  1e-12 sits
# above the rounding noise of a 2048-term fp64 inner product (~1e-14 relative) and moves the
# eigenvalues by O(tol^2).
_TOL = 1e-12
_SWEEPS = 60


def _activations_on_device(acts, device):
  if torch.is_tensor(acts):
    return acts.to(device=device, dtype=torch.float32).contiguous()
  return torch.from_numpy(np.ascontiguousarray(acts, dtype=np.float32)).to(device)


def frechet_distance(real_activations, generated_activations, device="cuda:0"):
  """tfgan.eval.frechet_classifier_distance_from_activations(real, generated) -> float."""
  real = _activations_on_device(real_activations, device)
  gen = _activations_on_device(generated_activations, device)
  if real.dim() != 2 or gen.dim() != 2 or real.shape[1] != gen.shape[1]:
    raise ValueError("activations must be [n, d] with equal d, got %s and %s" % (
        tuple(real.shape), tuple(gen.shape)))
  m, sigma = K.mean_cov_f64(real)
  m_v, sigma_v = K.mean_cov_f64(gen)
  # sqrt(sigma) = V^T diag(f(w)) V  (rows of V are eigenvectors); f and every scalar stay on the
  # device: one host read at the very end
  w, v = K.syevj_f64(sigma.clone(), max_sweeps=_SWEEPS, tol=_TOL)
  f, _ = K.spectral_sqrt_f64(w, _EPS)
  sqrt_sigma = K.gemm_f64(v, K.rowscale_f64(v, f), ta=True)
  inner = K.gemm_f64(K.gemm_f64(sqrt_sigma, sigma_v), sqrt_sigma)
  w2, _ = K.syevj_f64(inner, max_sweeps=_SWEEPS, tol=_TOL)
  _, sqrt_trace = K.spectral_sqrt_f64(w2, _EPS, want_values=False)
  return float(K.fid_combine_f64(sigma, sigma_v, m, m_v, sqrt_trace).item())


class FIDScoreTask(eval_task.EvalTask):
  """Evaluation task for the FID score (fid_score.py:39-56)."""

  _LABEL = "fid_score"

  def run_after_session(self, fake_dset, real_dset):
    fid = frechet_distance(real_dset.activations, fake_dset.activations,
                           device=_device_of(fake_dset.activations))
    return {self._LABEL: fid}


def _device_of(t):
  return t.device if torch.is_tensor(t) and t.is_cuda else torch.device("cuda:0")


def compute_fid_from_activations(fake_activations, real_activations):
  """Returns the FID based on activations (fid_score.py:58-75)."""
  assert tuple(fake_activations.shape) == tuple(real_activations.shape)
  return frechet_distance(real_activations, fake_activations,
                          device=_device_of(fake_activations))
