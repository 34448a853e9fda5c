// The discriminator's output head in one launch per direction (+ one small reduction):
//   pooled[n,c] = bf16( scale * sum_hw relu(x[n,hw,c]) )        tf.reduce_mean / reduce_sum over [1, 2]
//   logit[n]    = sum_c pooled[n,c] * bf16(w[c]) + bias          arch_ops.linear(h, 1)
// Reference call sites: resnet_cifar.py:154-157, resnet5.py:141-145, resnet_biggan.py:404-407
// (`h = tf.nn.relu(net); h = tf.math.reduce_sum(h, [1, 2]); out_logit = ops.linear(h, 1, ...)`),
// arch_ops.py:538-556.  Contract: include/cgamd.h (cg_pooled_head_fwd / _bwd).
//
// Why (profiles/r04_cifar_kernel_stats_after_small_kernels.csv): per discriminator call the chain
// spatial_reduce -> linear(C -> 1) -> ... -> linear dgrad -> linear wgrad -> colsum (2 launches) ->
// spatial_reduce_bwd is 9 launches of 5-8 us each -- 0.35 ms of the 6.7 ms ResNet-CIFAR step for a few
// hundred kFLOP.  Here one workgroup per sample walks its [HW, C] map once per direction; the bf16
// rounding points of the separate launches are kept (pooled, d_pooled and dx are bf16 tensors there),
// so the two forms agree to fp32 summation order.
#include "cg_common.h"

namespace {

constexpr int HB = 256;   // threads per workgroup

// channel-group layout of one workgroup: G = C / 8 groups of 8 channels (16 bytes); with G <= 256
// R = 256 / G rows are in flight (thread -> group tid % G, row lane tid / G); with G > 256 every
// thread walks groups tid, tid + 256, ... and all rows
struct HeadLayout {
  int G, R;
};
__device__ __forceinline__ HeadLayout head_layout(int C) {
  HeadLayout l;
  l.G = C >> 3;
  l.R = l.G <= HB ? HB / l.G : 1;
  return l;
}

__global__ __launch_bounds__(HB) void pooled_head_fwd_kernel(
    const bf16_t* __restrict__ x, int HW, int C, float scale, const float* __restrict__ w,
    const float* __restrict__ bias, bf16_t* __restrict__ pooled, float* __restrict__ logit) {
  extern __shared__ float sm[];          // [R][C] partial sums, then 4 floats of the block sum
  const int n = blockIdx.x, tid = threadIdx.x;
  const HeadLayout l = head_layout(C);
  const bf16_t* xn = x + (int64_t)n * HW * C;
  if (l.G <= HB) {
    const int g = tid % l.G, r = tid / l.G;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (r < l.R) {
      for (int p = r; p < HW; p += l.R) {
        float v[8];
        unpack8_bf16(*reinterpret_cast<const uint4*>(xn + (int64_t)p * C + g * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += fmaxf(v[e], 0.f);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sm[r * C + g * 8 + e] = s[e];
    }
  } else {
    for (int g = tid; g < l.G; g += HB) {
      float s[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] = 0.f;
      for (int p = 0; p < HW; ++p) {
        float v[8];
        unpack8_bf16(*reinterpret_cast<const uint4*>(xn + (int64_t)p * C + g * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += fmaxf(v[e], 0.f);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sm[g * 8 + e] = s[e];
    }
  }
  __syncthreads();
  float dot = 0.f;
  for (int c = tid; c < C; c += HB) {
    float t = 0.f;
    for (int r = 0; r < l.R; ++r) t += sm[r * C + c];
    const bf16_t p16 = f2bf(scale * t);
    pooled[(int64_t)n * C + c] = p16;
    dot += bf2f(p16) * bf2f(f2bf(w[c]));
  }
  float* sm4 = sm + (size_t)l.R * C;
  dot = block_sum_256(dot, sm4);
  if (tid == 0) logit[n] = dot + (bias ? bias[0] : 0.f);
}

// dx[n,hw,c] = x > 0 ? bf16(scale * dp[n,c]) : 0 with dp = bf16(bf16(dlogit[n]) * bf16(w[c]))
// (+ the gradient that arrives through `pooled` itself); dw_part[n][c] = pooled[n,c] * bf16(dlogit[n])
__global__ __launch_bounds__(HB) void pooled_head_bwd_kernel(
    const bf16_t* __restrict__ x, int HW, int C, float scale, const float* __restrict__ w,
    const float* __restrict__ dlogit, const bf16_t* __restrict__ dpooled_ext,
    const bf16_t* __restrict__ pooled, bf16_t* __restrict__ dx, float* __restrict__ dw_part) {
  const int n = blockIdx.x, tid = threadIdx.x;
  const HeadLayout l = head_layout(C);
  const float dl = dlogit ? bf2f(f2bf(dlogit[n])) : 0.f;
  const bf16_t* xn = x + (int64_t)n * HW * C;
  bf16_t* dxn = dx + (int64_t)n * HW * C;
  const int g0 = l.G <= HB ? tid % l.G : tid;
  const int gstep = l.G <= HB ? l.G : HB;          // (one trip when G <= 256)
  const int r = l.G <= HB ? tid / l.G : 0;
  if (r >= l.R) return;
  for (int g = g0; g < l.G; g += gstep) {
    float dv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = g * 8 + e;
      float dp = bf2f(f2bf(dl * bf2f(f2bf(w[c]))));
      if (dpooled_ext) dp = bf2f(f2bf(dp + bf2f(dpooled_ext[(int64_t)n * C + c])));
      dv[e] = bf2f(f2bf(scale * dp));
      if (dw_part && r == 0) dw_part[(int64_t)n * C + c] = bf2f(pooled[(int64_t)n * C + c]) * dl;
    }
    for (int p = r; p < HW; p += l.R) {
      float v[8], o[8];
      unpack8_bf16(*reinterpret_cast<const uint4*>(xn + (int64_t)p * C + g * 8), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = v[e] > 0.f ? dv[e] : 0.f;
      *reinterpret_cast<uint4*>(dxn + (int64_t)p * C + g * 8) = pack8_bf16(o);
    }
    if (l.G <= HB) break;
  }
}

// dw[c] = sum_n dw_part[n][c], dbias = sum_n bf16(dlogit[n]) (fixed order): a workgroup owns 32
// channels, 8 lanes per channel walk the samples (n = lane, lane + 8, ...), LDS combines them
__global__ __launch_bounds__(HB) void pooled_head_reduce_kernel(const float* __restrict__ dw_part,
                                                                const float* __restrict__ dlogit,
                                                                int N, int C, float* __restrict__ dw,
                                                                float* __restrict__ dbias) {
  __shared__ float sm[8][33];
  __shared__ float sm4[4];
  const int cl = threadIdx.x & 31, nl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (dw && c < C)
    for (int n = nl; n < N; n += 8) s += dw_part[(int64_t)n * C + c];
  sm[nl][cl] = s;
  __syncthreads();
  if (dw && nl == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += sm[r][cl];
    dw[c] = t;
  }
  if (dbias && blockIdx.x == 0) {   // (block-uniform)
    float t = 0.f;
    for (int n = threadIdx.x; n < N; n += HB) t += bf2f(f2bf(dlogit[n]));
    t = block_sum_256(t, sm4);
    if (threadIdx.x == 0) dbias[0] = t;
  }
}

// out[0] = sum_i a[i] * b[i] on fp32 (fixed order: 1024 strided lanes, then a tree): the gradient of
// a scalar that multiplies a weight tensor
__global__ __launch_bounds__(1024) void dot_f32_kernel(const float* __restrict__ a,
                                                       const float* __restrict__ b, int64_t n,
                                                       float* __restrict__ out) {
  __shared__ float sm[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s += a[i] * b[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += sm[w];
    out[0] = t;
  }
}

size_t head_lds_bytes(int C) {
  const int G = C >> 3, R = G <= HB ? HB / G : 1;
  return ((size_t)R * C + 4) * sizeof(float);
}

}  // namespace

extern "C" int cg_pooled_head_supported(int HW, int C) {
  return HW > 0 && C >= 8 && C % 8 == 0 && head_lds_bytes(C) <= 64 * 1024 ? 1 : 0;
}

extern "C" int cg_pooled_head_fwd(const void* x, int N, int HW, int C, float scale, const float* w,
                                  const float* bias, void* pooled, float* logit, cgStream stream) {
  if (!x || !w || !pooled || !logit || N <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_pooled_head_fwd: bad argument");
  if (!cg_pooled_head_supported(HW, C))
    CG_FAIL(CG_ERR_UNSUPPORTED, "cg_pooled_head_fwd: C must be a multiple of 8 (and at most 16k)");
  pooled_head_fwd_kernel<<<N, HB, head_lds_bytes(C), (hipStream_t)stream>>>(
      (const bf16_t*)x, HW, C, scale, w, bias, (bf16_t*)pooled, logit);
  CG_CHECK_LAUNCH("cg_pooled_head_fwd");
  return CG_OK;
}

extern "C" int cg_dot_f32(const float* a, const float* b, int64_t n, float* out, cgStream stream) {
  if (!a || !b || !out || n <= 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_dot_f32: bad argument");
  dot_f32_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(a, b, n, out);
  CG_CHECK_LAUNCH("cg_dot_f32");
  return CG_OK;
}

extern "C" size_t cg_pooled_head_bwd_workspace_bytes(int N, int C) {
  return N > 0 && C > 0 ? align_up((size_t)N * C * sizeof(float), 256) : 0;
}

extern "C" int cg_pooled_head_bwd(const void* x, int N, int HW, int C, float scale, const float* w,
                                  const float* dlogit, const void* dpooled_ext, const void* pooled,
                                  void* dx, float* dw, float* dbias, void* ws, size_t ws_bytes,
                                  cgStream stream) {
  if (!x || !w || !pooled || !dx || N <= 0 || (!dlogit && !dpooled_ext))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_pooled_head_bwd: bad argument");
  if ((dw || dbias) && !dlogit)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_pooled_head_bwd: parameter gradients need dlogit");
  if (!cg_pooled_head_supported(HW, C))
    CG_FAIL(CG_ERR_UNSUPPORTED, "cg_pooled_head_bwd: C must be a multiple of 8 (and at most 16k)");
  if (dw && (!ws || ws_bytes < cg_pooled_head_bwd_workspace_bytes(N, C)))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_pooled_head_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  pooled_head_bwd_kernel<<<N, HB, 0, st>>>((const bf16_t*)x, HW, C, scale, w, dlogit,
                                           (const bf16_t*)dpooled_ext, (const bf16_t*)pooled,
                                           (bf16_t*)dx, dw ? (float*)ws : nullptr);
  CG_CHECK_LAUNCH("cg_pooled_head_bwd");
  if (dw || dbias) {
    if (dw) {
      pooled_head_reduce_kernel<<<cdiv(C, 32), HB, 0, st>>>((const float*)ws, dlogit, N, C, dw, dbias);
    } else {
      pooled_head_reduce_kernel<<<1, HB, 0, st>>>(nullptr, dlogit, N, 0, nullptr, dbias);
    }
    CG_CHECK_LAUNCH("cg_pooled_head_bwd(reduce)");
  }
  return CG_OK;
}
