// Spectral normalisation (arch_ops.py:453-535): one power-iteration round in two streaming passes
// over the fp32 weight (w^T a as a split column reduction, w b as one wave per row), sigma from
// the second product's norm (sigma = u'^T w v = ||w v||^2 * rsqrt(max(||w v||^2, eps))), and the
// gradient through w / sigma with u, v held constant.  HBM-bound: 8 bytes per weight element.
#include "cg_common.h"

namespace {

// part[z][co] = sum_{k in slab z} a[k] * w[k,co]
__global__ __launch_bounds__(256) void colred_part_kernel(const float* __restrict__ w,
                                                          const float* __restrict__ a, int K,
                                                          int Co, int k_per_split,
                                                          float* __restrict__ part) {
  __shared__ float sm[4][64];
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int co = blockIdx.x * 64 + l;
  const int k0 = blockIdx.y * k_per_split, k1 = min(K, k0 + k_per_split);
  float s = 0.f;
  if (co < Co)
    for (int k = k0 + wv; k < k1; k += 4) s += a[k] * w[(int64_t)k * Co + co];
  sm[wv][l] = s;
  __syncthreads();
  if (wv == 0 && co < Co)
    part[(int64_t)blockIdx.y * Co + co] = sm[0][l] + sm[1][l] + sm[2][l] + sm[3][l];
}

// t[k] = sum_co w[k,co] * b[co]; one wave per row, 4 rows per block.
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ w,
                                                     const float* __restrict__ b, int K, int Co,
                                                     float* __restrict__ t) {
  const int l = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= K) return;
  float s = 0.f;
  for (int co = l; co < Co; co += 64) s += w[(int64_t)k * Co + co] * b[co];
  s = wave_sum(s);
  if (l == 0) t[k] = s;
}

// Single block (1024 threads): raw[i] = sum_z part[z][i]; out = raw * rsqrt(max(sum raw^2, eps)).
// Optionally also writes sigma = sum raw^2 * rsqrt(max(sum raw^2, eps)) and 1/sigma.
__global__ __launch_bounds__(1024) void l2n_final_kernel(const float* __restrict__ part,
                                                         int splits, int n, float eps,
                                                         float* __restrict__ out,
                                                         float* __restrict__ sigma,
                                                         float* __restrict__ inv_sigma) {
  __shared__ float sm[16];
  float ss = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float r = 0.f;
    for (int z = 0; z < splits; ++z) r += part[(int64_t)z * n + i];
    out[i] = r;  // raw for now
    ss += r * r;
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) tot += sm[i];
  const float rs = rsqrtf(fmaxf(tot, eps));
  for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] *= rs;
  if (threadIdx.x == 0 && sigma) {
    const float s = tot * rs;
    *sigma = s;
    if (inv_sigma) *inv_sigma = 1.f / s;
  }
}

// <dwbar, w> partial sums
__global__ __launch_bounds__(256) void dot_part_kernel(const float* __restrict__ a,
                                                       const float* __restrict__ b, int64_t n,
                                                       float* __restrict__ part) {
  __shared__ float sm4[4];
  float s = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    s += a[i] * b[i];
  s = block_sum_256(s, sm4);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void sn_bwd_kernel(const float* __restrict__ dwbar, const float* __restrict__ part,
                              int nparts, int64_t n, int Co, const float* __restrict__ a_k,
                              const float* __restrict__ b_co, const float* __restrict__ sigma,
                              float* __restrict__ dw) {
  // every thread re-sums the (<= 512) partials: fixed order -> deterministic
  float dot = 0.f;
  for (int i = 0; i < nparts; ++i) dot += part[i];
  const float inv = 1.f / *sigma;
  const float coef = dot * inv;  // <dwbar, w_bar>
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t k = i / Co;
    const int co = (int)(i - k * Co);
    dw[i] = (dwbar[i] - coef * a_k[k] * b_co[co]) * inv;
  }
}

__global__ void scale_f32_kernel(const float* __restrict__ x, const float* __restrict__ sdev,
                                 float shost, float* __restrict__ out, int64_t n) {
  const float s = (sdev ? *sdev : 1.f) * shost;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = x[i] * s;
}

inline int colred_splits(int K, int Co) {
  // enough blocks to pull the matrix at full rate, few enough that the single-block finalisation
  // (which re-reads every partial) stays short
  const int ct = cdiv(Co, 64);
  int s = cdiv(128, ct);
  const int maxs = K / 64 > 0 ? K / 64 : 1;
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  return s;
}
inline int dot_blocks(int64_t n) {
  int64_t b = (n + 1023) / 1024;
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" size_t cg_spectral_norm_workspace_bytes(int K, int Co) {
  if (K <= 0 || Co <= 0) return 0;
  // colred partials [splits][Co] + one K-vector (t = w b raw)
  return align_up(((size_t)colred_splits(K, Co) * Co + (size_t)K + (size_t)Co) * sizeof(float),
                  256);
}

extern "C" int cg_spectral_norm(const float* w, int K, int Co, int mode, float eps,
                                const float* u_in, float* u_out, float* v_out, float* sigma,
                                float* inv_sigma, void* ws, size_t ws_bytes, cgStream stream) {
  if (!w || !u_in || !u_out || !v_out || !sigma || K <= 0 || Co <= 0 || (mode != 0 && mode != 1))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_spectral_norm: bad argument");
  if (!ws || ws_bytes < cg_spectral_norm_workspace_bytes(K, Co))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_spectral_norm: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int splits = colred_splits(K, Co);
  const int kps = (K + splits - 1) / splits;
  float* part = (float*)ws;                    // [splits][Co]
  float* tk = part + (size_t)splits * Co;      // [K]
  dim3 cgrid(cdiv(Co, 64), splits);
  if (mode == 0) {
    // left: u [K];  v = l2n(w^T u) [Co];  u' = l2n(w v) [K];  sigma = ||w v||^2 rs
    colred_part_kernel<<<cgrid, 256, 0, st>>>(w, u_in, K, Co, kps, part);
    CG_CHECK_LAUNCH("cg_spectral_norm(colred)");
    l2n_final_kernel<<<1, 1024, 0, st>>>(part, splits, Co, eps, v_out, nullptr, nullptr);
    CG_CHECK_LAUNCH("cg_spectral_norm(l2n v)");
    rowdot_kernel<<<cdiv(K, 4), 256, 0, st>>>(w, v_out, K, Co, tk);
    CG_CHECK_LAUNCH("cg_spectral_norm(rowdot)");
    l2n_final_kernel<<<1, 1024, 0, st>>>(tk, 1, K, eps, u_out, sigma, inv_sigma);
    CG_CHECK_LAUNCH("cg_spectral_norm(l2n u)");
  } else {
    // right: u [Co];  v = l2n(w u^T) [K];  u' = l2n(v^T w) [Co];  sigma = ||v^T w||^2 rs
    rowdot_kernel<<<cdiv(K, 4), 256, 0, st>>>(w, u_in, K, Co, tk);
    CG_CHECK_LAUNCH("cg_spectral_norm(rowdot)");
    l2n_final_kernel<<<1, 1024, 0, st>>>(tk, 1, K, eps, v_out, nullptr, nullptr);
    CG_CHECK_LAUNCH("cg_spectral_norm(l2n v)");
    colred_part_kernel<<<cgrid, 256, 0, st>>>(w, v_out, K, Co, kps, part);
    CG_CHECK_LAUNCH("cg_spectral_norm(colred)");
    l2n_final_kernel<<<1, 1024, 0, st>>>(part, splits, Co, eps, u_out, sigma, inv_sigma);
    CG_CHECK_LAUNCH("cg_spectral_norm(l2n u)");
  }
  return CG_OK;
}

extern "C" size_t cg_sn_backward_workspace_bytes(int K, int Co) {
  if (K <= 0 || Co <= 0) return 0;
  return align_up((size_t)dot_blocks((int64_t)K * Co) * sizeof(float), 256);
}

extern "C" int cg_sn_backward(const float* dwbar, const float* w, int K, int Co, const float* a_k,
                              const float* b_co, const float* sigma, float* dw, void* ws,
                              size_t ws_bytes, cgStream stream) {
  if (!dwbar || !w || !a_k || !b_co || !sigma || !dw || K <= 0 || Co <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_sn_backward: bad argument");
  if (!ws || ws_bytes < cg_sn_backward_workspace_bytes(K, Co))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_sn_backward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = (int64_t)K * Co;
  const int nb = dot_blocks(n);
  dot_part_kernel<<<nb, 256, 0, st>>>(dwbar, w, n, (float*)ws);
  CG_CHECK_LAUNCH("cg_sn_backward(dot)");
  int64_t eb = (n + 255) / 256;
  if (eb > 2048) eb = 2048;
  sn_bwd_kernel<<<(int)eb, 256, 0, st>>>(dwbar, (const float*)ws, nb, n, Co, a_k, b_co, sigma, dw);
  CG_CHECK_LAUNCH("cg_sn_backward(apply)");
  return CG_OK;
}

extern "C" int cg_scale_f32(const float* x, const float* scale_dev, float scale_host, float* out,
                            int64_t n, cgStream stream) {
  if (n < 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_scale_f32: negative size");
  if (n == 0) return CG_OK;
  if (!x || !out) CG_FAIL(CG_ERR_BAD_ARG, "cg_scale_f32: null pointer");
  int64_t b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  scale_f32_kernel<<<(int)b, 256, 0, (hipStream_t)stream>>>(x, scale_dev, scale_host, out, n);
  CG_CHECK_LAUNCH("cg_scale_f32");
  return CG_OK;
}
