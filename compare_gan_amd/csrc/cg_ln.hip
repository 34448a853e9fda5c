// Layer normalisation (arch_ops.py:448-450 -> tf.contrib.layers.layer_norm with its defaults:
// begin_norm_axis = 1, begin_params_axis = -1, variance_epsilon = 1e-12): per SAMPLE mean / variance
// over all of (H, W, C), per-CHANNEL gamma / beta:
//   y[n,p,c] = ((x[n,p,c] - mean_n) * rsqrt(var_n + eps)) * gamma[c] + beta[c]
// Used by ResNetBlock / BigGanResNetBlock when D.layer_norm = True (resnet_ops.py:162-173,
// resnet_biggan.py:123-134): the WGAN-GP paper's discriminator normaliser.  x, y bf16 [N, M, C]
// (M = H*W, C % 8 == 0), statistics fp32 (sums in fp64), HBM-bound: one workgroup per sample for the
// two per-sample reductions, grid-stride vector kernels for the element-wise passes, deterministic
// two-stage column sums for dgamma / dbeta.
#include "cg_common.h"

namespace {

constexpr int LN_T = 1024;   // threads of a per-sample reduction workgroup

// channel of element e of the 8-element vector i of a sample: vectors are channel-aligned when
// C % 8 == 0; otherwise (the RGB input of a discriminator's first block: C = 3, H * W * C % 8 == 0)
// the channel is the flat index modulo C
__device__ __forceinline__ int ln_ch(int64_t i, int e, int C) {
  return (C & 7) == 0 ? (int)(i % (C >> 3)) * 8 + e : (int)((i * 8 + e) % C);
}

__device__ __forceinline__ double ln_block_sum(double v, double* sm) {
  v = wave_sum_d(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < nw; ++i) s += sm[i];
  return s;
}

// stats[n] = (mean, rstd) of sample n
__global__ __launch_bounds__(LN_T) void ln_stats_kernel(const bf16_t* __restrict__ x, int64_t per,
                                                        float eps, float* __restrict__ mean,
                                                        float* __restrict__ rstd) {
  __shared__ double sm[LN_T / 64];
  const uint4* xp = reinterpret_cast<const uint4*>(x + (int64_t)blockIdx.x * per);
  double s1 = 0.0, s2 = 0.0;
  for (int64_t i = threadIdx.x; i < per / 8; i += LN_T) {
    float v[8];
    unpack8_bf16(xp[i], v);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a += v[e];
      b += v[e] * v[e];
    }
    s1 += a;
    s2 += b;
  }
  s1 = ln_block_sum(s1, sm);
  s2 = ln_block_sum(s2, sm);
  if (threadIdx.x == 0) {
    const double m = s1 / (double)per;
    double var = s2 / (double)per - m * m;   // tf.nn.moments: mean of squared deviations
    if (var < 0.0) var = 0.0;
    mean[blockIdx.x] = (float)m;
    rstd[blockIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

__global__ __launch_bounds__(256) void ln_apply_kernel(const bf16_t* __restrict__ x, int64_t per,
                                                       int C, int64_t total8,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       bf16_t* __restrict__ y) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t per8 = per / 8;
  const int c8n = C / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += stride) {
    const int64_t n = i / per8, iv = i - n * per8;
    const float m = mean[n], r = rstd[n];
    float v[8];
    unpack8_bf16(reinterpret_cast<const uint4*>(x)[i], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = ln_ch(iv, e, C);
      v[e] = ((v[e] - m) * r) * gamma[c] + beta[c];
    }
    reinterpret_cast<uint4*>(y)[i] = pack8_bf16(v);
  }
}

// per sample: a_n = sum dy * gamma, b_n = sum dy * gamma * xhat  (both divided by `per` later)
__global__ __launch_bounds__(LN_T) void ln_bwd_sample_kernel(const bf16_t* __restrict__ x,
                                                             const bf16_t* __restrict__ dy,
                                                             int64_t per, int C,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma,
                                                             float* __restrict__ ab) {
  __shared__ double sm[LN_T / 64];
  const int64_t base = (int64_t)blockIdx.x * per;
  const uint4* xp = reinterpret_cast<const uint4*>(x + base);
  const uint4* dp = reinterpret_cast<const uint4*>(dy + base);
  const float m = mean[blockIdx.x], r = rstd[blockIdx.x];
  const int c8n = C / 8;
  double sa = 0.0, sb = 0.0;
  for (int64_t i = threadIdx.x; i < per / 8; i += LN_T) {
    float xv[8], dv[8];
    unpack8_bf16(xp[i], xv);
    unpack8_bf16(dp[i], dv);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float g = dv[e] * gamma[ln_ch(i, e, C)];
      a += g;
      b += g * ((xv[e] - m) * r);
    }
    sa += a;
    sb += b;
  }
  sa = ln_block_sum(sa, sm);
  sb = ln_block_sum(sb, sm);
  if (threadIdx.x == 0) {
    ab[blockIdx.x * 2 + 0] = (float)(sa / (double)per);
    ab[blockIdx.x * 2 + 1] = (float)(sb / (double)per);
  }
}

// dx = rstd * (dy * gamma - a_n - xhat * b_n)
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const bf16_t* __restrict__ x,
                                                        const bf16_t* __restrict__ dy, int64_t per,
                                                        int C, int64_t total8,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ ab,
                                                        bf16_t* __restrict__ dx) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t per8 = per / 8;
  const int c8n = C / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += stride) {
    const int64_t n = i / per8, iv = i - n * per8;
    const float m = mean[n], r = rstd[n], a = ab[n * 2], b = ab[n * 2 + 1];
    float xv[8], dv[8], o[8];
    unpack8_bf16(reinterpret_cast<const uint4*>(x)[i], xv);
    unpack8_bf16(reinterpret_cast<const uint4*>(dy)[i], dv);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      o[e] = r * (dv[e] * gamma[ln_ch(iv, e, C)] - a - ((xv[e] - m) * r) * b);
    reinterpret_cast<uint4*>(dx)[i] = pack8_bf16(o);
  }
}

// partial column sums over the rows (n, p) of dy (-> dbeta) and dy * xhat (-> dgamma):
// block b owns rows b, b + gridDim.x, ...; thread = (row lane, 8-channel group)
constexpr int LN_PB = 256;   // partial blocks
__global__ __launch_bounds__(256) void ln_bwd_param_part_kernel(const bf16_t* __restrict__ x,
                                                                const bf16_t* __restrict__ dy,
                                                                int64_t rows, int64_t M, int C,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd,
                                                                float* __restrict__ part) {
  const int c8n = C / 8;
  // each thread walks (row, group) items r * c8n + g of this block's rows, fixed group per thread
  // when 256 % c8n == 0 or c8n % 256 == 0; general: recompute the group per item
  for (int g0 = threadIdx.x; g0 < c8n; g0 += 256) {
    float sg[8], sb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sg[e] = sb[e] = 0.f;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
      const int64_t n = r / M;
      const float m = mean[n], rs = rstd[n];
      float xv[8], dv[8];
      unpack8_bf16(reinterpret_cast<const uint4*>(x + r * C)[g0], xv);
      unpack8_bf16(reinterpret_cast<const uint4*>(dy + r * C)[g0], dv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sb[e] += dv[e];
        sg[e] += dv[e] * ((xv[e] - m) * rs);
      }
    }
    float* p = part + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      p[g0 * 8 + e] = sg[e];
      p[C + g0 * 8 + e] = sb[e];
    }
  }
}
// the same partial sums for channel counts that are not a multiple of 8 (RGB): thread = channel,
// scalar loads; U != nullptr selects the second-order form (sum of dy * t, see ln_bb_gamma_part_kernel)
__global__ __launch_bounds__(256) void ln_param_part_scalar_kernel(const bf16_t* __restrict__ x,
                                                                  const bf16_t* __restrict__ dy,
                                                                  const bf16_t* __restrict__ U,
                                                                  int64_t rows, int64_t M, int C,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd,
                                                                  const float* __restrict__ sums,
                                                                  float* __restrict__ part) {
  for (int c = threadIdx.x; c < C; c += 256) {
    float sg = 0.f, sb = 0.f;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
      const int64_t n = r / M;
      const float m = mean[n], rs = rstd[n];
      const float xh = (bf2f(x[r * C + c]) - m) * rs, d = bf2f(dy[r * C + c]);
      if (U) {
        sg += d * (rs * (bf2f(U[r * C + c]) - sums[n * 5] - xh * sums[n * 5 + 1]));
      } else {
        sb += d;
        sg += d * xh;
      }
    }
    part[(int64_t)blockIdx.x * 2 * C + c] = sg;
    part[(int64_t)blockIdx.x * 2 * C + C + c] = sb;
  }
}

__global__ __launch_bounds__(256) void ln_bwd_param_final_kernel(const float* __restrict__ part,
                                                                 int blocks, int C,
                                                                 float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= 2 * C) return;
  double s = 0.0;
  for (int b = 0; b < blocks; ++b) s += part[(int64_t)b * 2 * C + c];
  if (c < C) {
    if (dgamma) dgamma[c] = (float)s;
  } else if (dbeta) {
    dbeta[c - C] = (float)s;
  }
}

// ---- second order: the gradient of  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// (ln_bwd_dx_kernel) with respect to dy, x and gamma, for an upstream gradient u = dL/d(dx) -- what
// a gradient penalty needs when D.layer_norm = True (resnet_ops.py:162-173 under penalty_lib.py:59-82).
// With S(a, b) = mean over the sample of a * b and the SYMMETRIC map P = rstd * (I - 11'/D - xhat xhat'/D)
// (dx = P g, and d(xhat) = P dx):
//   t      = P u = rstd * (u - mean(u) - xhat * S(u, xhat))
//   dL/ddy = gamma * t                    dL/dgamma[c] = sum over (n, p) of dy * t
//   dL/dx  = -rstd^2 * S(u, h) * xhat - rstd * P (m2 * u + S(u, xhat) * g),   h = g - m1 - xhat * m2,
//            m1 = mean(g), m2 = S(g, xhat)
// per sample five means: sums[n] = (mean u, S(u, xhat), S(u, g), m1, m2)
__global__ __launch_bounds__(LN_T) void ln_bb_sample_kernel(const bf16_t* __restrict__ x,
                                                            const bf16_t* __restrict__ dy,
                                                            const bf16_t* __restrict__ u, int64_t per,
                                                            int C, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            float* __restrict__ sums) {
  __shared__ double sm[LN_T / 64];
  const int64_t base = (int64_t)blockIdx.x * per;
  const uint4* xp = reinterpret_cast<const uint4*>(x + base);
  const uint4* dp = reinterpret_cast<const uint4*>(dy + base);
  const uint4* up = reinterpret_cast<const uint4*>(u + base);
  const float m = mean[blockIdx.x], r = rstd[blockIdx.x];
  const int c8n = C / 8;
  double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int64_t i = threadIdx.x; i < per / 8; i += LN_T) {
    float xv[8], dv[8], uv[8];
    unpack8_bf16(xp[i], xv);
    unpack8_bf16(dp[i], dv);
    unpack8_bf16(up[i], uv);
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float g = dv[e] * gamma[ln_ch(i, e, C)], xh = (xv[e] - m) * r;
      a[0] += uv[e];
      a[1] += uv[e] * xh;
      a[2] += uv[e] * g;
      a[3] += g;
      a[4] += g * xh;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) s[k] += a[k];
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) s[k] = ln_block_sum(s[k], sm);
  if (threadIdx.x == 0)
#pragma unroll
    for (int k = 0; k < 5; ++k) sums[blockIdx.x * 5 + k] = (float)(s[k] / (double)per);
}

__global__ __launch_bounds__(256) void ln_bb_elem_kernel(const bf16_t* __restrict__ x,
                                                         const bf16_t* __restrict__ dy,
                                                         const bf16_t* __restrict__ u, int64_t per,
                                                         int C, int64_t total8,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ sums,
                                                         bf16_t* __restrict__ d_dy,
                                                         bf16_t* __restrict__ d_x) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t per8 = per / 8;
  const int c8n = C / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += stride) {
    const int64_t n = i / per8, iv = i - n * per8;
    const float m = mean[n], r = rstd[n];
    const float ua = sums[n * 5], sux = sums[n * 5 + 1], sug = sums[n * 5 + 2], m1 = sums[n * 5 + 3],
                m2 = sums[n * 5 + 4];
    const float suh = sug - ua * m1 - sux * m2;          // S(u, h)
    const float wm = m2 * ua + sux * m1, swx = 2.f * m2 * sux;   // mean(w), S(w, xhat), w = m2 u + sux g
    float xv[8], dv[8], uv[8], o1[8], o2[8];
    unpack8_bf16(reinterpret_cast<const uint4*>(x)[i], xv);
    unpack8_bf16(reinterpret_cast<const uint4*>(dy)[i], dv);
    unpack8_bf16(reinterpret_cast<const uint4*>(u)[i], uv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gm = gamma[ln_ch(iv, e, C)], g = dv[e] * gm, xh = (xv[e] - m) * r;
      o1[e] = gm * (r * (uv[e] - ua - xh * sux));
      o2[e] = -(r * r) * (suh * xh + (m2 * uv[e] + sux * g - wm - xh * swx));
    }
    reinterpret_cast<uint4*>(d_dy)[i] = pack8_bf16(o1);
    reinterpret_cast<uint4*>(d_x)[i] = pack8_bf16(o2);
  }
}

// partial column sums of dy * t over the rows (layout of ln_bwd_param_part_kernel, first C columns)
__global__ __launch_bounds__(256) void ln_bb_gamma_part_kernel(const bf16_t* __restrict__ x,
                                                               const bf16_t* __restrict__ dy,
                                                               const bf16_t* __restrict__ u,
                                                               int64_t rows, int64_t M, int C,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd,
                                                               const float* __restrict__ sums,
                                                               float* __restrict__ part) {
  const int c8n = C / 8;
  for (int g0 = threadIdx.x; g0 < c8n; g0 += 256) {
    float sg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sg[e] = 0.f;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
      const int64_t n = r / M;
      const float m = mean[n], rs = rstd[n], ua = sums[n * 5], sux = sums[n * 5 + 1];
      float xv[8], dv[8], uv[8];
      unpack8_bf16(reinterpret_cast<const uint4*>(x + r * C)[g0], xv);
      unpack8_bf16(reinterpret_cast<const uint4*>(dy + r * C)[g0], dv);
      unpack8_bf16(reinterpret_cast<const uint4*>(u + r * C)[g0], uv);
#pragma unroll
      for (int e = 0; e < 8; ++e) sg[e] += dv[e] * (rs * (uv[e] - ua - ((xv[e] - m) * rs) * sux));
    }
    float* p = part + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      p[g0 * 8 + e] = sg[e];
      p[C + g0 * 8 + e] = 0.f;
    }
  }
}

int ln_grid(int64_t total8) {
  int64_t b = (total8 + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int cg_layer_norm_fwd(const void* x, int N, int64_t M, int C, const float* gamma,
                                 const float* beta, float eps, void* y, float* mean, float* rstd,
                                 cgStream stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || N <= 0 || M <= 0 || C <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_layer_norm_fwd: bad argument");
  if ((M * C) % 8) CG_FAIL(CG_ERR_UNSUPPORTED, "cg_layer_norm_fwd: H * W * C must be a multiple of 8");
  hipStream_t st = (hipStream_t)stream;
  const int64_t per = M * C, total8 = (int64_t)N * per / 8;
  ln_stats_kernel<<<N, LN_T, 0, st>>>((const bf16_t*)x, per, eps, mean, rstd);
  ln_apply_kernel<<<ln_grid(total8), 256, 0, st>>>((const bf16_t*)x, per, C, total8, mean, rstd,
                                                   gamma, beta, (bf16_t*)y);
  CG_CHECK_LAUNCH("cg_layer_norm_fwd");
  return CG_OK;
}

extern "C" size_t cg_layer_norm_bwd_workspace_bytes(int N, int C) {
  return align_up((size_t)N * 2 * sizeof(float), 256) + (size_t)LN_PB * 2 * C * sizeof(float);
}

extern "C" int cg_layer_norm_bwd(const void* x, const void* dy, const float* mean,
                                 const float* rstd, const float* gamma, int N, int64_t M, int C,
                                 void* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                 cgStream stream) {
  if (!x || !dy || !mean || !rstd || !gamma || !dx || N <= 0 || M <= 0 || C <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_layer_norm_bwd: bad argument");
  if ((M * C) % 8) CG_FAIL(CG_ERR_UNSUPPORTED, "cg_layer_norm_bwd: H * W * C must be a multiple of 8");
  if (!ws || ws_bytes < cg_layer_norm_bwd_workspace_bytes(N, C))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_layer_norm_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t per = M * C, total8 = (int64_t)N * per / 8;
  float* ab = (float*)ws;
  float* part = (float*)((char*)ws + align_up((size_t)N * 2 * sizeof(float), 256));
  ln_bwd_sample_kernel<<<N, LN_T, 0, st>>>((const bf16_t*)x, (const bf16_t*)dy, per, C, mean, rstd,
                                           gamma, ab);
  ln_bwd_dx_kernel<<<ln_grid(total8), 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)dy, per, C,
                                                    total8, mean, rstd, gamma, ab, (bf16_t*)dx);
  if (dgamma || dbeta) {
    if (C % 8)
      ln_param_part_scalar_kernel<<<LN_PB, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)dy, nullptr,
                                                         (int64_t)N * M, M, C, mean, rstd, nullptr, part);
    else
      ln_bwd_param_part_kernel<<<LN_PB, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)dy,
                                                      (int64_t)N * M, M, C, mean, rstd, part);
    ln_bwd_param_final_kernel<<<cdiv(2 * C, 256), 256, 0, st>>>(part, LN_PB, C, dgamma, dbeta);
  }
  CG_CHECK_LAUNCH("cg_layer_norm_bwd");
  return CG_OK;
}

extern "C" size_t cg_layer_norm_bwd_bwd_workspace_bytes(int N, int C) {
  return align_up((size_t)N * 5 * sizeof(float), 256) + (size_t)LN_PB * 2 * C * sizeof(float);
}

extern "C" int cg_layer_norm_bwd_bwd(const void* x, const void* dy, const void* u, const float* mean,
                                     const float* rstd, const float* gamma, int N, int64_t M, int C,
                                     void* d_dy, void* d_x, float* d_gamma, void* ws, size_t ws_bytes,
                                     cgStream stream) {
  if (!x || !dy || !u || !mean || !rstd || !gamma || !d_dy || !d_x || N <= 0 || M <= 0 || C <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_layer_norm_bwd_bwd: bad argument");
  if ((M * C) % 8) CG_FAIL(CG_ERR_UNSUPPORTED, "cg_layer_norm_bwd_bwd: H * W * C must be a multiple of 8");
  if (!ws || ws_bytes < cg_layer_norm_bwd_bwd_workspace_bytes(N, C))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_layer_norm_bwd_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t per = M * C, total8 = (int64_t)N * per / 8;
  float* sums = (float*)ws;
  float* part = (float*)((char*)ws + align_up((size_t)N * 5 * sizeof(float), 256));
  ln_bb_sample_kernel<<<N, LN_T, 0, st>>>((const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)u, per, C,
                                          mean, rstd, gamma, sums);
  ln_bb_elem_kernel<<<ln_grid(total8), 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)dy,
                                                     (const bf16_t*)u, per, C, total8, mean, rstd,
                                                     gamma, sums, (bf16_t*)d_dy, (bf16_t*)d_x);
  if (d_gamma) {
    if (C % 8)
      ln_param_part_scalar_kernel<<<LN_PB, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)dy,
                                                         (const bf16_t*)u, (int64_t)N * M, M, C, mean,
                                                         rstd, sums, part);
    else
      ln_bb_gamma_part_kernel<<<LN_PB, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)dy,
                                                     (const bf16_t*)u, (int64_t)N * M, M, C, mean, rstd,
                                                     sums, part);
    ln_bwd_param_final_kernel<<<cdiv(2 * C, 256), 256, 0, st>>>(part, LN_PB, C, d_gamma, nullptr);
  }
  CG_CHECK_LAUNCH("cg_layer_norm_bwd_bwd");
  return CG_OK;
}
