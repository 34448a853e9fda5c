// Batch-norm family (arch_ops.py:194-319 standardize_batch, :327-367 batch_norm, :423-445
// conditional_batch_norm).  x is viewed as [N, HW, C] bf16; statistics and all reductions fp32.
// HBM-bound: every pass streams the activation once with 16-byte loads (32 lanes x 16 B = one
// contiguous 512 B row segment), partial sums go through a caller workspace (deterministic).
#include "cg_common.h"

namespace {

union V8 {
  uint4 q;
  bf16_t h[8];
};

// ---- statistics ------------------------------------------------------------------------------
// grid (ceil(CV/32), splits); thread (cg = tid%32, rl = tid/32) accumulates rows rl, rl+8, ...
__global__ __launch_bounds__(256) void bn_stats_part_kernel(const bf16_t* __restrict__ x,
                                                            int64_t rows, int C,
                                                            int64_t rows_per_split,
                                                            float* __restrict__ part) {
  __shared__ float sm[8][32][17];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int cv = blockIdx.x * 32 + cg;
  const int CV = C / 8;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_split;
  const int64_t r1 = min(rows, r0 + rows_per_split);
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
  if (cv < CV) {
    for (int64_t r = r0 + rl; r < r1; r += 8) {
      V8 v;
      v.q = *reinterpret_cast<const uint4*>(x + r * C + (int64_t)cv * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = bf2f(v.h[e]);
        s[e] += f;
        q[e] += f * f;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sm[rl][cg][e] = s[e];
    sm[rl][cg][8 + e] = q[e];
  }
  __syncthreads();
  // 256 threads: 32 channel groups x 8 elements -> sum over the 8 row lanes
  const int cg2 = threadIdx.x >> 3, e2 = threadIdx.x & 7;
  const int cv2 = blockIdx.x * 32 + cg2;
  if (cv2 < CV) {
    float ss = 0.f, qq = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      ss += sm[r][cg2][e2];
      qq += sm[r][cg2][8 + e2];
    }
    float* p = part + (int64_t)blockIdx.y * 2 * C;
    p[cv2 * 8 + e2] = ss;
    p[C + cv2 * 8 + e2] = qq;
  }
}
// scalar fallback (C % 8 != 0): grid (ceil(C/64), splits), 4 waves split rows.
__global__ __launch_bounds__(256) void bn_stats_part_scalar_kernel(const bf16_t* __restrict__ x,
                                                                   int64_t rows, int C,
                                                                   int64_t rows_per_split,
                                                                   float* __restrict__ part) {
  __shared__ float sm[2][4][64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + l;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_split;
  const int64_t r1 = min(rows, r0 + rows_per_split);
  float s = 0.f, q = 0.f;
  if (c < C)
    for (int64_t r = r0 + w; r < r1; r += 4) {
      const float f = bf2f(x[r * C + c]);
      s += f;
      q += f * f;
    }
  sm[0][w][l] = s;
  sm[1][w][l] = q;
  __syncthreads();
  if (w == 0 && c < C) {
    float* p = part + (int64_t)blockIdx.y * 2 * C;
    p[c] = sm[0][0][l] + sm[0][1][l] + sm[0][2][l] + sm[0][3][l];
    p[C + c] = sm[1][0][l] + sm[1][1][l] + sm[1][2][l] + sm[1][3][l];
  }
}
__global__ void bn_stats_final_kernel(const float* __restrict__ part, int splits, int C,
                                      float inv_rows, float* __restrict__ mean,
                                      float* __restrict__ var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f, q = 0.f;
  for (int z = 0; z < splits; ++z) {
    s += part[(int64_t)z * 2 * C + c];
    q += part[(int64_t)z * 2 * C + C + c];
  }
  const float m = s * inv_rows;
  mean[c] = m;
  var[c] = q * inv_rows - m * m;  // tf.nn.normalize_moments(shift=None)
}
inline int stats_splits(int64_t rows, int C) {
  const int ct = (C % 8 == 0) ? cdiv(C / 8, 32) : cdiv(C, 64);
  int s = cdiv(1024, ct);
  const int64_t maxs = rows / 32 > 0 ? rows / 32 : 1;
  if (s > maxs) s = (int)maxs;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return s;
}

// ---- apply -----------------------------------------------------------------------------------
template <bool VEC>
__global__ void bn_apply_kernel(const bf16_t* __restrict__ x, int HW, int C, int64_t total_units,
                                const float* __restrict__ mean, const float* __restrict__ var,
                                float eps, const float* __restrict__ gamma,
                                const float* __restrict__ beta, int per_sample, int relu,
                                bf16_t* __restrict__ y) {
  const int CV = VEC ? C / 8 : C;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_units; i += stride) {
    const int cv = (int)(i % CV);
    const int64_t row = i / CV;
    const int64_t n = row / HW;
    const int c0 = VEC ? cv * 8 : cv;
    const int64_t pidx = per_sample ? n * C + c0 : c0;
    if (VEC) {
      V8 v, o;
      v.q = *reinterpret_cast<const uint4*>(x + row * C + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float rstd = rsqrtf(var[c0 + e] + eps);
        float t = (bf2f(v.h[e]) - mean[c0 + e]) * rstd;
        if (gamma) t *= gamma[pidx + e];
        if (beta) t += beta[pidx + e];
        if (relu) t = fmaxf(t, 0.f);
        o.h[e] = f2bf(t);
      }
      *reinterpret_cast<uint4*>(y + row * C + c0) = o.q;
    } else {
      const float rstd = rsqrtf(var[c0] + eps);
      float t = (bf2f(x[row * C + c0]) - mean[c0]) * rstd;
      if (gamma) t *= gamma[pidx];
      if (beta) t += beta[pidx];
      if (relu) t = fmaxf(t, 0.f);
      y[row * C + c0] = f2bf(t);
    }
  }
}

// ---- backward --------------------------------------------------------------------------------
// pass 1: S1[n,c] = sum_hw dz, S2[n,c] = sum_hw dz*xhat, dz = dy * (y>0 if relu).
// grid (ceil(C/64), N, hsplits): 4 waves split hw; partial [hs][2][N][C].
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ y, const bf16_t* __restrict__ dy,
    int HW, int C, int hw_per_split, const float* __restrict__ mean,
    const float* __restrict__ var, float eps, int relu, float* __restrict__ part) {
  __shared__ float sm[2][4][64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + l;
  const int n = blockIdx.y, N = gridDim.y;
  const int h0 = blockIdx.z * hw_per_split, h1 = min(HW, h0 + hw_per_split);
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
    const float mu = mean[c], rstd = rsqrtf(var[c] + eps);
    for (int p = h0 + w; p < h1; p += 4) {
      const int64_t o = ((int64_t)n * HW + p) * C + c;
      float g = bf2f(dy[o]);
      if (relu && !(bf2f(y[o]) > 0.f)) g = 0.f;
      s1 += g;
      s2 += g * (bf2f(x[o]) - mu) * rstd;
    }
  }
  sm[0][w][l] = s1;
  sm[1][w][l] = s2;
  __syncthreads();
  if (w == 0 && c < C) {
    float* p = part + (int64_t)blockIdx.z * 2 * N * C;
    p[(int64_t)n * C + c] = sm[0][0][l] + sm[0][1][l] + sm[0][2][l] + sm[0][3][l];
    p[(int64_t)N * C + (int64_t)n * C + c] = sm[1][0][l] + sm[1][1][l] + sm[1][2][l] + sm[1][3][l];
  }
}
// finalize: one thread per channel
__global__ void bn_bwd_final_kernel(const float* __restrict__ part, int hsplits, int N, int C,
                                    float inv_rows, const float* __restrict__ gamma,
                                    int per_sample, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta, float* __restrict__ m12) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float t1 = 0.f, t2 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int n = 0; n < N; ++n) {
    float s1 = 0.f, s2 = 0.f;
    for (int z = 0; z < hsplits; ++z) {
      const float* p = part + (int64_t)z * 2 * N * C;
      s1 += p[(int64_t)n * C + c];
      s2 += p[(int64_t)N * C + (int64_t)n * C + c];
    }
    if (per_sample) {
      const float g = gamma ? gamma[(int64_t)n * C + c] : 1.f;
      if (dgamma) dgamma[(int64_t)n * C + c] = s2;
      if (dbeta) dbeta[(int64_t)n * C + c] = s1;
      a1 += g * s1;
      a2 += g * s2;
    } else {
      t1 += s1;
      t2 += s2;
    }
  }
  if (!per_sample) {
    const float g = gamma ? gamma[c] : 1.f;
    if (dgamma) dgamma[c] = t2;
    if (dbeta) dbeta[c] = t1;
    a1 = g * t1;
    a2 = g * t2;
  }
  m12[c] = a1 * inv_rows;
  m12[C + c] = a2 * inv_rows;
}
// pass 2
__global__ void bn_bwd_dx_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ y,
                                 const bf16_t* __restrict__ dy, int HW, int C, int64_t total,
                                 const float* __restrict__ mean, const float* __restrict__ var,
                                 float eps, const float* __restrict__ gamma, int per_sample,
                                 int relu, int batch_stats, const float* __restrict__ m12,
                                 bf16_t* __restrict__ dx) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C);
    const int64_t n = i / ((int64_t)HW * C);
    const float rstd = rsqrtf(var[c] + eps);
    float g = bf2f(dy[i]);
    if (relu && !(bf2f(y[i]) > 0.f)) g = 0.f;
    const float gm = gamma ? gamma[per_sample ? n * C + c : c] : 1.f;
    float d = gm * g;
    if (batch_stats) {
      const float xhat = (bf2f(x[i]) - mean[c]) * rstd;
      d = d - m12[c] - xhat * m12[C + c];
    }
    dx[i] = f2bf(d * rstd);
  }
}

__global__ void bn_update_moving_kernel(float* __restrict__ mm, float* __restrict__ mv,
                                        const float* __restrict__ mean,
                                        const float* __restrict__ var, int C, float decay) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mm[c] -= (1.f - decay) * (mm[c] - mean[c]);
  mv[c] -= (1.f - decay) * (mv[c] - var[c]);
}

// to_variance == 0: second <- var + mean^2 (E[x^2]);  == 1: mean *= scale, second *= scale, then
// second <- second - mean^2  (tpu_ops.py:112-120 parallel moments after the all-reduce SUM).
__global__ void bn_moments_convert_kernel(float* __restrict__ mean, float* __restrict__ second,
                                          int C, int to_variance, float scale) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (!to_variance) {
    second[c] = second[c] + mean[c] * mean[c];
  } else {
    const float m = mean[c] * scale, q = second[c] * scale;
    mean[c] = m;
    second[c] = q - m * m;
  }
}

inline int bwd_hsplits(int N, int HW, int C) {
  const int blocks = cdiv(C, 64) * N;
  int s = cdiv(1024, blocks);
  const int maxs = HW / 16 > 0 ? HW / 16 : 1;
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  return s;
}
inline int grid_cap(int64_t work) {
  int64_t b = (work + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" size_t cg_bn_stats_workspace_bytes(int64_t rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  return align_up((size_t)stats_splits(rows, C) * 2 * C * sizeof(float), 256);
}

extern "C" int cg_bn_stats(const void* x, int64_t rows, int C, float* mean, float* var, void* ws,
                           size_t ws_bytes, cgStream stream) {
  if (!x || !mean || !var || rows <= 0 || C <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_bn_stats: bad argument");
  if (!ws || ws_bytes < cg_bn_stats_workspace_bytes(rows, C))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_bn_stats: workspace too small");
  const int splits = stats_splits(rows, C);
  const int64_t rps = (rows + splits - 1) / splits;
  hipStream_t st = (hipStream_t)stream;
  if (C % 8 == 0) {
    dim3 grid(cdiv(C / 8, 32), splits);
    bn_stats_part_kernel<<<grid, 256, 0, st>>>((const bf16_t*)x, rows, C, rps, (float*)ws);
  } else {
    dim3 grid(cdiv(C, 64), splits);
    bn_stats_part_scalar_kernel<<<grid, 256, 0, st>>>((const bf16_t*)x, rows, C, rps, (float*)ws);
  }
  CG_CHECK_LAUNCH("cg_bn_stats(part)");
  bn_stats_final_kernel<<<cdiv(C, 256), 256, 0, st>>>((const float*)ws, splits, C,
                                                      1.0f / (float)rows, mean, var);
  CG_CHECK_LAUNCH("cg_bn_stats(final)");
  return CG_OK;
}

extern "C" int cg_bn_apply(const void* x, int N, int HW, int C, const float* mean,
                           const float* var, float eps, const float* gamma, const float* beta,
                           int per_sample, int relu, void* y, cgStream stream) {
  if (!x || !y || !mean || !var || N <= 0 || HW <= 0 || C <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_bn_apply: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (C % 8 == 0) {
    const int64_t units = (int64_t)N * HW * (C / 8);
    bn_apply_kernel<true><<<grid_cap(units), 256, 0, st>>>((const bf16_t*)x, HW, C, units, mean,
                                                           var, eps, gamma, beta, per_sample,
                                                           relu, (bf16_t*)y);
  } else {
    const int64_t units = (int64_t)N * HW * C;
    bn_apply_kernel<false><<<grid_cap(units), 256, 0, st>>>((const bf16_t*)x, HW, C, units, mean,
                                                            var, eps, gamma, beta, per_sample,
                                                            relu, (bf16_t*)y);
  }
  CG_CHECK_LAUNCH("cg_bn_apply");
  return CG_OK;
}

extern "C" size_t cg_bn_backward_workspace_bytes(int N, int HW, int C) {
  if (N <= 0 || HW <= 0 || C <= 0) return 0;
  const size_t hs = bwd_hsplits(N, HW, C);
  return align_up(hs * 2 * (size_t)N * C * sizeof(float), 256);
}

extern "C" int cg_bn_backward_reduce(const void* x, const void* y, const void* dy, int N, int HW,
                                     int C, const float* mean, const float* var, float eps,
                                     const float* gamma, int per_sample, int relu, float* dgamma,
                                     float* dbeta, float* m12, void* ws, size_t ws_bytes,
                                     cgStream stream) {
  if (!x || !dy || !m12 || !mean || !var || N <= 0 || HW <= 0 || C <= 0 || (relu && !y))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_bn_backward_reduce: bad argument");
  if (!ws || ws_bytes < cg_bn_backward_workspace_bytes(N, HW, C))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_bn_backward_reduce: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int hs = bwd_hsplits(N, HW, C);
  const int hps = (HW + hs - 1) / hs;
  float* part = (float*)ws;
  dim3 grid(cdiv(C, 64), N, hs);
  bn_bwd_sums_kernel<<<grid, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy,
                                           HW, C, hps, mean, var, eps, relu, part);
  CG_CHECK_LAUNCH("cg_bn_backward_reduce(sums)");
  bn_bwd_final_kernel<<<cdiv(C, 256), 256, 0, st>>>(part, hs, N, C,
                                                    1.0f / ((float)N * (float)HW), gamma,
                                                    per_sample, dgamma, dbeta, m12);
  CG_CHECK_LAUNCH("cg_bn_backward_reduce(final)");
  return CG_OK;
}

extern "C" int cg_bn_backward_apply(const void* x, const void* y, const void* dy, int N, int HW,
                                    int C, const float* mean, const float* var, float eps,
                                    const float* gamma, int per_sample, int relu, int batch_stats,
                                    const float* m12, void* dx, cgStream stream) {
  if (!x || !dy || !dx || !mean || !var || N <= 0 || HW <= 0 || C <= 0 || (relu && !y) ||
      (batch_stats && !m12))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_bn_backward_apply: bad argument");
  const int64_t total = (int64_t)N * HW * C;
  bn_bwd_dx_kernel<<<grid_cap(total), 256, 0, (hipStream_t)stream>>>(
      (const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, HW, C, total, mean, var, eps, gamma,
      per_sample, relu, batch_stats, m12, (bf16_t*)dx);
  CG_CHECK_LAUNCH("cg_bn_backward_apply");
  return CG_OK;
}

extern "C" int cg_bn_moments_convert(float* mean, float* second, int C, int to_variance,
                                     float scale, cgStream stream) {
  if (!mean || !second || C <= 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_bn_moments_convert: bad argument");
  bn_moments_convert_kernel<<<cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(mean, second, C,
                                                                           to_variance, scale);
  CG_CHECK_LAUNCH("cg_bn_moments_convert");
  return CG_OK;
}

extern "C" int cg_bn_update_moving(float* moving_mean, float* moving_var, const float* mean,
                                   const float* var, int C, float decay, cgStream stream) {
  if (!moving_mean || !moving_var || !mean || !var || C <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_bn_update_moving: bad argument");
  bn_update_moving_kernel<<<cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(moving_mean, moving_var,
                                                                         mean, var, C, decay);
  CG_CHECK_LAUNCH("cg_bn_update_moving");
  return CG_OK;
}
