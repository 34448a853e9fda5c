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
import os

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


# Which symmetric-square-root path frechet_distance takes: "auto" = the GEMM-only Newton-Schulz
# iteration when both covariances come from more samples than features and it certifies itself (see
# _sqrt_newton_schulz), the Jacobi eigen-solver otherwise; "jacobi" / "newton" force one (tests).
_SOLVER = os.environ.get("CGAMD_FID_SOLVER", "auto")
# the second matrix square root only contributes its TRACE: no eigenvectors (A/B switch, 1 = on)
_VALUES_ONLY = int(os.environ.get("CGAMD_FID_VALUES_ONLY", "1"))
# the FIRST root from the rotated rows themselves instead of an accumulated eigenvector matrix (1 = on)
_ROOT_FROM_G = int(os.environ.get("CGAMD_FID_ROOT_FROM_G", "1"))
_NS_MAX_ITER = 64
# The trace of the second square root from eigenvalues computed by tridiagonalisation + bisection
# (cg_sytrd_eigvals_f64: ~0.05 s at d = 2048 against ~0.33 s of values-only Jacobi sweeps), accepted only
# under a certificate.  The method is backward stable in the ABSOLUTE sense: its values are the exact
# eigenvalues of A + E.  |E|_F is taken as _TRIDIAG_C * sqrt(d) * u * |A|_F (the sqrt(d) growth of the
# probabilistic rounding-error model, Higham & Mary 2019) and |E|_2 as a quarter of that from d = 64 on
# (a rounding-error matrix has |E|_2 ~ 2 |E|_F / sqrt(d)): at d = 2048 that is 181 u resp. 45 u of |A|_F
# against 17 u resp. 6 u measured, and tests/test_kernels_gpu.py::test_tridiagonal_eigenvalues asserts
# the measured errors of ten matrix kinds (d = 2 ... 2048) at a quarter resp. half of the assumed ones.
# cg_spectral_sqrt_bound_f64 turns the two into a bound on the error of sum f(|w_i|) (Weyl, Hoffman-
# Wielandt + Cauchy-Schwarz; an eigenvalue within |E|_2 of tfgan's 1e-10 cut-off, where f jumps from
# 1e-10 to 1e-5, is charged that jump).  Twice the bound (the trace enters the distance twice) must
# stay below _TRIDIAG_ACCEPT * (tr sigma + tr sigma_v), the natural scale of the distance; 1e-6 is the
# band the parity tests hold the distance to.  What fails it: many (near-)zero eigenvalues together
# with a norm large enough that their rounding level reaches the cut-off (fewer samples than features
# at feature scales >> 1) -- those take the Jacobi solve, whose eigenvalues are accurate relative to
# themselves.  CGAMD_FID_TRIDIAG=0 switches the path off.
_TRIDIAG = int(os.environ.get("CGAMD_FID_TRIDIAG", "1"))
_TRIDIAG_MIN_D = 128
_TRIDIAG_MAX_D = 4096
_TRIDIAG_C = 4.0
_TRIDIAG_ACCEPT = 1e-6
LAST_TRIDIAG = {}      # certificate of the last call (tests, bench)
LAST_SOLVER = {"sqrt_sigma": None, "trace_sqrt": None}   # what the last call used (tests, bench)
LAST_NEWTON = []   # one record per _sqrt_newton_schulz call of the last frechet_distance()


def _sqrt_newton_schulz(a, want_matrix):
  """Symmetric square root of a positive definite matrix by the coupled, inverse-free Newton-Schulz
  iteration (Higham, Functions of Matrices, eq. 6.35): with c = |a|_F (>= the largest eigenvalue),
      Y_0 = a / c, Z_0 = I;   T_k = (3 I - Z_k Y_k) / 2;   Y_{k+1} = Y_k T_k;   Z_{k+1} = T_k Z_k
  converges quadratically to Y = (a/c)^(1/2), Z = (a/c)^(-1/2): three 2048^3 fp64 GEMMs per step
  instead of thousands of latency-bound Jacobi rounds (profiles/r02_fid10k_kernel_stats.csv: 15,240
  launches, 1.34 s).  tfgan's rule (_symmetric_matrix_square_root: eigenvalues below 1e-10 are NOT
  square-rooted) differs from the true square root only if such eigenvalues exist, so the result is
  accepted only when it CERTIFIES that none does: converged (|Z Y - I|_F < 1e-7, one more
  quadratic step) and lambda_min(a) >= c / |Z|_F^2 > 10 * 1e-10.  Returns (sqrt(a) or None,
  device scalar trace(sqrt(a))) or None when not certified (the caller falls back to Jacobi)."""
  st = K.mat_stats_f64(a).tolist()          # one host read: the scale
  c = float(st[1]) ** 0.5
  rec = {"fro": c, "trace": float(st[0]), "min_diag": float(st[2]), "iters": 0, "residual2": None,
         "lam_min_bound": None, "accepted": False}
  LAST_NEWTON.append(rec)
  if not (c > 0.0) or not np.isfinite(c):
    return None
  # the smallest diagonal entry bounds the smallest eigenvalue from above: a feature without
  # variance (a dead ReLU channel) already rules the certificate out -- no iteration is spent
  if not (float(st[2]) > 10.0 * _EPS):
    return None
  # an eigenvalue lambda reaches Z Y = 1 after about log_2.25(c / lambda) steps: past that count
  # for lambda = 1e-9 (+ the quadratic tail) the certificate cannot hold any more
  max_iter = min(_NS_MAX_ITER, int(np.ceil(np.log(c / (10.0 * _EPS)) / np.log(2.25))) + 8)
  t = K.axpby_eye_f64(a, -0.5 / c, 1.5)      # T_0 = (3 I - a / c) / 2
  y = K.gemm_f64(a, t, alpha=1.0 / c)        # Y_1 = Y_0 T_0
  z = t                                      # Z_1 = T_0 Z_0
  converged = False
  for it in range(1, max(max_iter, 5)):
    t = K.gemm_f64(z, y, alpha=-0.5, eye=1.5)
    if it % 4 == 0:
      # T - I = (I - Z Y) / 2, formed explicitly: |T|_F^2 - 2 tr(T) + n would cancel to noise
      sq_e = K.mat_stats_f64(K.axpby_eye_f64(t, 1.0, -1.0)).tolist()[1]
      rec["iters"], rec["residual2"] = it, sq_e
      if not np.isfinite(sq_e):
        return None
      if sq_e < 0.25e-14:                    # |Z Y - I|_F < 1e-7
        converged = True
    y = K.gemm_f64(y, t)
    z = K.gemm_f64(t, z)
    if converged:
      break
  if not converged:
    return None
  tr_y = K.mat_stats_f64(y).tolist()[0]
  sq_z = K.mat_stats_f64(z).tolist()[1]
  lam_min_lower_bound = c / sq_z             # |Z|_F^2 = sum_i c / lambda_i >= c / lambda_min
  rec["lam_min_bound"] = lam_min_lower_bound
  if not (lam_min_lower_bound > 10.0 * _EPS):
    return None
  rec["accepted"] = True
  root = None
  if want_matrix:
    root = K.axpby_eye_f64(y, c ** 0.5, 0.0)
  return root, (c ** 0.5) * tr_y


def _trace_sqrt_values_only(inner, scale):
  """Device scalar [1] = sum_i f(|lambda_i(inner)|), f = tfgan's square-root rule: tridiagonalisation +
  bisection under its certificate (see _TRIDIAG above; `scale` = tr sigma + tr sigma_v), else the
  values-only Jacobi solve.  Destroys `inner`."""
  d = inner.shape[0]
  LAST_TRIDIAG.clear()
  if _TRIDIAG and _TRIDIAG_MIN_D <= d <= _TRIDIAG_MAX_D:
    keep = inner.clone()
    w, fro = K.sytrd_eigvals_f64(inner)
    delta_rel = _TRIDIAG_C * (d ** 0.5) * 2.220446049250313e-16
    out = K.spectral_sqrt_bound_f64(w, _EPS, delta_rel, delta_rel / (4.0 if d >= 64 else 1.0), fro)
    total, bound = out.tolist()              # one host read: the certificate
    ok = np.isfinite(total) and np.isfinite(bound) and 2.0 * bound <= _TRIDIAG_ACCEPT * scale
    LAST_TRIDIAG.update({"fro": float(fro.item()), "delta": delta_rel * float(fro.item()),
                         "bound": bound, "scale": scale, "accepted": bool(ok)})
    if ok:
      LAST_SOLVER["trace_sqrt"] = "tridiagonal+bisection"
      return out[:1]
    inner = keep
  w2, _ = K.syevj_f64(inner, max_sweeps=_SWEEPS, tol=_TOL, want_vectors=_VALUES_ONLY == 0)
  _, sqrt_trace = K.spectral_sqrt_f64(w2, _EPS, want_values=False)
  return sqrt_trace


def frechet_distance(real_activations, generated_activations, device="cuda:0"):
  """tfgan.eval.frechet_classifier_distance_from_activations(real, generated) -> float."""
  real = _activations_on_device(real_activations, device)
  gen = _activations_on_device(generated_activations, device)
  if real.dim() != 2 or gen.dim() != 2 or real.shape[1] != gen.shape[1]:
    raise ValueError("activations must be [n, d] with equal d, got %s and %s" % (
        tuple(real.shape), tuple(gen.shape)))
  m, sigma = K.mean_cov_f64(real)
  m_v, sigma_v = K.mean_cov_f64(gen)
  d = real.shape[1]
  try_newton = _SOLVER == "newton" or (_SOLVER == "auto" and min(real.shape[0], gen.shape[0]) > d
                                       and d >= 64)
  sqrt_sigma = None
  del LAST_NEWTON[:]
  LAST_SOLVER["sqrt_sigma"] = LAST_SOLVER["trace_sqrt"] = "jacobi"
  if try_newton:
    res = _sqrt_newton_schulz(sigma, True)
    if res is not None:
      sqrt_sigma = res[0]
      LAST_SOLVER["sqrt_sigma"] = "newton-schulz"
  if sqrt_sigma is not None:
    inner = K.gemm_f64(K.gemm_f64(sqrt_sigma, sigma_v), sqrt_sigma)
    res = _sqrt_newton_schulz(inner, False)
    if res is not None:
      LAST_SOLVER["trace_sqrt"] = "newton-schulz"
      sqrt_trace = torch.tensor([res[1]], dtype=torch.float64, device=sigma.device)
      return float(K.fid_combine_f64(sigma, sigma_v, m, m_v, sqrt_trace).item())
    sqrt_trace = _trace_sqrt_values_only(inner, _trace_scale(sigma, sigma_v))
    return float(K.fid_combine_f64(sigma, sigma_v, m, m_v, sqrt_trace).item())
  # sqrt(sigma) = V^T diag(f(w)) V  (rows of V are eigenvectors); f and every scalar stay on the
  # device: one host read at the very end
  # One-sided Jacobi rotates the rows of G (= sigma at the start) until they are orthogonal: g_i =
  # lambda_i v_i.  The root sum_i f(lambda_i) v_i v_i^T is then G^T diag(f(|lambda_i|) / lambda_i^2) G:
  # no eigenvector matrix has to be carried through the sweeps (half the traffic of every apply pass).
  # (A covariance is positive semi-definite: a negative lambda_i is rounding noise of size ~1e-16
  # |sigma|, far under the 1e-10 cut-off, where f(|lambda|) = |lambda| -- its sign does not matter.)
  g = sigma.clone()
  w, v = K.syevj_f64(g, max_sweeps=_SWEEPS, tol=_TOL, want_vectors=_ROOT_FROM_G == 0)
  if v is None:
    LAST_SOLVER["sqrt_sigma"] = "jacobi (rows)"
    sqrt_sigma = K.gemm_f64(g, K.rowscale_f64(g, K.spectral_root_scale_f64(w, _EPS)), ta=True)
  else:
    f, _ = K.spectral_sqrt_f64(w, _EPS)
    sqrt_sigma = K.gemm_f64(v, K.rowscale_f64(v, f), ta=True)
  inner = K.gemm_f64(K.gemm_f64(sqrt_sigma, sigma_v), sqrt_sigma)
  sqrt_trace = _trace_sqrt_values_only(inner, _trace_scale(sigma, sigma_v))
  return float(K.fid_combine_f64(sigma, sigma_v, m, m_v, sqrt_trace).item())


def _trace_scale(sigma, sigma_v):
  return float(K.mat_stats_f64(sigma).tolist()[0]) + float(K.mat_stats_f64(sigma_v).tolist()[0])


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
