"""ORACLE -- test infrastructure only.

FID / Inception score as the reference computes them through the un-vendored dependency
tensorflow-gan==0.0.0.dev0 (setup.py:38; call sites metrics/fid_score.py:49-51,72-74 and
metrics/inception_score.py:44).  The algorithm is restated from tfgan's published
frechet_classifier_distance_from_activations / classifier_score_from_logits:
  * cast to float64; m = mean; sigma = Xc^T Xc / (n-1)
  * trace_sqrt_product(sigma, sigma_v) = trace(sqrtm(sqrt(sigma) sigma_v sqrt(sigma))) with
    _symmetric_matrix_square_root(mat, eps=1e-10) = U diag(where(s<eps, s, sqrt(s))) V^T  (SVD)
  * fid = trace(sigma) + trace(sigma_v) - 2 trace_sqrt_product + |m - m_v|^2
Pinned by the reference's golden value FID = 89.091 +- 1e-4 (metrics/fid_score_test.py:31-40).
IS is "parity unpinned" (no reference test asserts a value).
"""
import numpy as np


def symmetric_matrix_square_root(mat, eps=1e-10):
  u, s, vt = np.linalg.svd(mat)
  si = np.where(s < eps, s, np.sqrt(s))
  return (u * si) @ vt


def trace_sqrt_product(sigma, sigma_v):
  sqrt_sigma = symmetric_matrix_square_root(sigma)
  sqrt_a_sigmav_a = sqrt_sigma @ sigma_v @ sqrt_sigma
  return np.trace(symmetric_matrix_square_root(sqrt_a_sigmav_a))


def mean_cov(acts):
  acts = np.asarray(acts, dtype=np.float64)
  n = acts.shape[0]
  m = acts.mean(axis=0)
  c = acts - m
  return m, c.T @ c / (n - 1)


def frechet_distance(real_activations, generated_activations):
  m, sigma = mean_cov(real_activations)
  m_w, sigma_w = mean_cov(generated_activations)
  sqrt_trace_component = trace_sqrt_product(sigma, sigma_w)
  trace = np.trace(sigma + sigma_w) - 2.0 * sqrt_trace_component
  mean = np.sum((m - m_w) ** 2)
  return trace + mean


def compute_fid_from_activations(fake_activations, real_activations):
  """metrics/fid_score.py:58-75 argument order."""
  assert fake_activations.shape == real_activations.shape
  return frechet_distance(real_activations, fake_activations)


def classifier_score_from_logits(logits):
  """exp(mean_i KL(p(y|x_i) || p(y))) in float64 (tfgan classifier_score_from_logits)."""
  logits = np.asarray(logits, dtype=np.float64)
  mx = logits.max(axis=1, keepdims=True)
  lse = mx + np.log(np.exp(logits - mx).sum(axis=1, keepdims=True))
  log_p = logits - lse
  p = np.exp(log_p)
  q = p.mean(axis=0, keepdims=True)
  kl = (p * (log_p - np.log(q))).sum(axis=1)
  return float(np.exp(kl.mean()))


def resize_bilinear_tf1(images, out_h, out_w):
  """TF1 tf.image.resize_bilinear(align_corners=False), legacy (no half-pixel centres):
  src = dst * in/out; lo = floor(src); hi = min(lo+1, in-1). images NHWC float."""
  images = np.asarray(images, dtype=np.float64)
  n, h, w, c = images.shape
  ys = np.arange(out_h, dtype=np.float32) * (np.float32(h) / np.float32(out_h))
  xs = np.arange(out_w, dtype=np.float32) * (np.float32(w) / np.float32(out_w))
  y0 = np.floor(ys).astype(np.int64)
  x0 = np.floor(xs).astype(np.int64)
  y1 = np.minimum(y0 + 1, h - 1)
  x1 = np.minimum(x0 + 1, w - 1)
  wy = (ys - y0).astype(np.float64)[None, :, None, None]
  wx = (xs - x0).astype(np.float64)[None, None, :, None]
  tl = images[:, y0][:, :, x0]
  tr = images[:, y0][:, :, x1]
  bl = images[:, y1][:, :, x0]
  br = images[:, y1][:, :, x1]
  top = tl + (tr - tl) * wx
  bot = bl + (br - bl) * wx
  return top + (bot - top) * wy


def inception_preprocess(images_0_255, size=299):
  """tfgan.eval.preprocess_image: resize_bilinear to 299x299 then (x - 128) / 128
  (eval_utils.py:165-175)."""
  return (resize_bilinear_tf1(images_0_255, size, size) - 128.0) / 128.0


def kid(fake_activations, real_activations, max_batch_size=1024):
  """Restatement of metrics/kid_score.py:44-149 in NumPy fp64 (block estimator, cubic polynomial
  kernel, including the reference's use of the REAL block size for both normalisers and of
  bins_r[0] when trimming both bin arrays).  No golden value exists in the reference
  (kid_score has no test): parity unpinned, checked by the estimator's closed form on tiny inputs
  in tests/test_oracle_pins.py."""
  import math
  real = np.asarray(real_activations, dtype=np.float64)
  fake = np.asarray(fake_activations, dtype=np.float64)
  n_real, dim = real.shape
  n_gen, _ = fake.shape
  n_bins = int(math.ceil(max(n_real, n_gen) / max_batch_size))
  bins_r = np.full(n_bins, int(math.ceil(n_real / n_bins)))
  bins_g = np.full(n_bins, int(math.ceil(n_gen / n_bins)))
  bins_r[:(n_bins * bins_r[0]) - n_real] -= 1
  bins_g[:(n_bins * bins_r[0]) - n_gen] -= 1
  inds_r = np.r_[0, np.cumsum(bins_r)]
  inds_g = np.r_[0, np.cumsum(bins_g)]
  ests = []
  for i in range(n_bins):
    r = real[inds_r[i]:inds_r[i + 1]]
    g = fake[inds_g[i]:inds_g[i + 1]]
    m = n = float(r.shape[0])
    k_rr = (r @ r.T / dim + 1) ** 3
    k_rg = (r @ g.T / dim + 1) ** 3
    k_gg = (g @ g.T / dim + 1) ** 3
    ests.append(-2 * k_rg.mean() + (k_rr.sum() - np.trace(k_rr)) / (m * (m - 1)) +
                (k_gg.sum() - np.trace(k_gg)) / (n * (n - 1)))
  return float(np.mean(ests))
