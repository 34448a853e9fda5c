// Losses (gans/loss_lib.py:53-148), WGAN-GP reductions (gans/penalty_lib.py:59-82), the fused
// multi-tensor TF-Adam(+EMA) update (modular_gan.py:480-508), gradient bucket
// gather/scatter for the RCCL all-reduce, step counters and the stateless Philox RNG
// (tpu/tpu_random.py semantics).  Contracts: include/cgamd.h.
#include "cg_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// GAN losses: single block, B <= a few thousand.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus_neg_abs(float x) { return log1pf(expf(-fabsf(x))); }
// sigmoid_cross_entropy_with_logits(x, z) = max(x,0) - x z + log(1 + exp(-|x|))
__device__ __forceinline__ float sce(float x, float z) {
  return fmaxf(x, 0.f) - x * z + softplus_neg_abs(x);
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void gan_loss_kernel(int kind, const float* __restrict__ logits,
                                                       int B, float* __restrict__ losses,
                                                       float* __restrict__ dd,
                                                       float* __restrict__ dg) {
  __shared__ float sm4[4];
  const float invB = 1.f / (float)B;
  float lr = 0.f, lf = 0.f, lg = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const float xr = logits[i], xf = logits[B + i];
    float gr = 0.f, gf = 0.f, gg = 0.f;
    if (kind == 0) {  // non_saturating
      lr += sce(xr, 1.f);
      lf += sce(xf, 0.f);
      lg += sce(xf, 1.f);
      gr = sigmoidf_(xr) - 1.f;
      gf = sigmoidf_(xf);
      gg = sigmoidf_(xf) - 1.f;
    } else if (kind == 1) {  // wasserstein
      lr += -xr;
      lf += xf;
      lg += -xf;
      gr = -1.f;
      gf = 1.f;
      gg = -1.f;
    } else if (kind == 2) {  // least_squares (on probabilities)
      const float pr = sigmoidf_(xr), pf = sigmoidf_(xf);
      lr += (pr - 1.f) * (pr - 1.f);
      lf += pf * pf;
      lg += 0.5f * (pf - 1.f) * (pf - 1.f);
      gr = (pr - 1.f) * pr * (1.f - pr);  // 0.5 * 2 (p-1) p (1-p)
      gf = pf * pf * (1.f - pf);
      gg = (pf - 1.f) * pf * (1.f - pf);
    } else {  // hinge
      lr += fmaxf(1.f - xr, 0.f);
      lf += fmaxf(1.f + xf, 0.f);
      lg += -xf;
      gr = (1.f - xr) > 0.f ? -1.f : 0.f;
      gf = (1.f + xf) > 0.f ? 1.f : 0.f;
      gg = -1.f;
    }
    if (dd) {
      dd[i] = gr * invB;
      dd[B + i] = gf * invB;
    }
    if (dg) {
      dg[i] = 0.f;
      dg[B + i] = gg * invB;
    }
  }
  lr = block_sum_256(lr, sm4);
  lf = block_sum_256(lf, sm4);
  lg = block_sum_256(lg, sm4);
  if (threadIdx.x == 0) {
    lr *= invB;
    lf *= invB;
    lg *= invB;
    const float d = kind == 2 ? 0.5f * (lr + lf) : lr + lf;
    losses[0] = d;
    losses[1] = lr;
    losses[2] = lf;
    losses[3] = lg;
  }
}

// SSGAN rotation head (gans/ssgan.py:186-203): loss = -mean_i log(softmax(logits_i)[label_i] +
// 1e-10) over n rows of k classes, and dlogits = d loss / d logits; one block, fixed order.
__global__ __launch_bounds__(256) void softmax_xent_eps_kernel(const float* __restrict__ logits,
                                                               const int* __restrict__ labels,
                                                               int n, int k, float eps,
                                                               float* __restrict__ loss,
                                                               float* __restrict__ dlogits) {
  __shared__ float sm4[4];
  const float inv_n = 1.f / (float)n;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float* row = logits + (int64_t)i * k;
    float mx = row[0];
    for (int c = 1; c < k; ++c) mx = fmaxf(mx, row[c]);
    float se = 0.f;
    for (int c = 0; c < k; ++c) se += expf(row[c] - mx);
    const int y = labels[i];
    const float py = expf(row[y] - mx) / se;
    acc += -logf(py + eps);
    // d(-log(p_y + eps)) / d logit_c = -(p_y / (p_y + eps)) * (delta_yc - p_c)
    const float w = py / (py + eps) * inv_n;
    for (int c = 0; c < k; ++c) {
      const float pc = expf(row[c] - mx) / se;
      dlogits[(int64_t)i * k + c] = -w * ((c == y ? 1.f : 0.f) - pc);
    }
  }
  acc = block_sum_256(acc, sm4);
  if (threadIdx.x == 0) *loss = acc * inv_n;
}

// S3GAN (gans/s3gan.py:118-160): one block per example.
//   avail[i] = sum_k y[i,k] > 0.5 (a label was passed);  y_out[i,:] = y[i,:] where a label is
//   available, else the predictor's label: softmax(aux[i,:]) (soft) or one_hot(argmax aux[i,:]).
//   aux may be NULL (no predictor): y_out = y.
__global__ __launch_bounds__(256) void s3gan_labels_kernel(const float* __restrict__ aux,
                                                           const bf16_t* __restrict__ y, int k,
                                                           int soft, bf16_t* __restrict__ y_out,
                                                           float* __restrict__ avail) {
  __shared__ float sm4[4];
  __shared__ float s_max;
  __shared__ int s_arg;
  const int i = blockIdx.x, t = threadIdx.x;
  const bf16_t* yr = y + (int64_t)i * k;
  bf16_t* yo = y_out + (int64_t)i * k;
  float s = 0.f;
  for (int c = t; c < k; c += 256) s += bf2f(yr[c]);
  s = block_sum_256(s, sm4);
  const bool has = s > 0.5f;
  if (t == 0) avail[i] = has ? 1.f : 0.f;
  if (has || aux == nullptr) {
    for (int c = t; c < k; c += 256) yo[c] = yr[c];
    return;
  }
  const float* ar = aux + (int64_t)i * k;
  if (t == 0) {   // k <= a few thousand: a serial scan keeps tf.arg_max's first-maximum rule
    float mx = ar[0];
    int arg = 0;
    for (int c = 1; c < k; ++c)
      if (ar[c] > mx) { mx = ar[c]; arg = c; }
    s_max = mx;
    s_arg = arg;
  }
  __syncthreads();
  if (!soft) {
    for (int c = t; c < k; c += 256) yo[c] = c == s_arg ? (bf16_t)0x3f80 : (bf16_t)0;
    return;
  }
  float se = 0.f;
  for (int c = t; c < k; c += 256) se += expf(ar[c] - s_max);
  se = block_sum_256(se, sm4);
  for (int c = t; c < k; c += 256) yo[c] = f2bf(expf(ar[c] - s_max) / se);
}

// tf.losses.softmax_cross_entropy(labels, logits, weights) with the default reduction
// SUM_BY_NONZERO_WEIGHTS (s3gan.py:311-313): loss = sum_i w_i * CE_i / #{w_i != 0},
// CE_i = -sum_k y_ik * log_softmax(logits_i)_k; dlogits = d loss / d logits (labels are constants).
__global__ __launch_bounds__(256) void softmax_xent_weighted_kernel(
    const float* __restrict__ logits, const bf16_t* __restrict__ labels,
    const float* __restrict__ weights, int n, int k, float* __restrict__ loss,
    float* __restrict__ dlogits) {
  __shared__ float sm4[4];
  const int t = threadIdx.x;
  float cnt = 0.f;
  for (int i = t; i < n; i += 256) cnt += weights[i] != 0.f ? 1.f : 0.f;
  cnt = block_sum_256(cnt, sm4);
  const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
  float acc = 0.f;
  for (int i = t; i < n; i += 256) {
    const float* row = logits + (int64_t)i * k;
    const bf16_t* yr = labels + (int64_t)i * k;
    float mx = row[0];
    for (int c = 1; c < k; ++c) mx = fmaxf(mx, row[c]);
    float se = 0.f, ys = 0.f, dot = 0.f;
    for (int c = 0; c < k; ++c) {
      se += expf(row[c] - mx);
      const float yv = bf2f(yr[c]);
      ys += yv;
      dot += yv * (row[c] - mx);
    }
    const float lse = logf(se);
    const float w = weights[i];
    acc += w * (ys * lse - dot);
    for (int c = 0; c < k; ++c)
      dlogits[(int64_t)i * k + c] = w * inv * (expf(row[c] - mx - lse) * ys - bf2f(yr[c]));
  }
  acc = block_sum_256(acc, sm4);
  if (t == 0) *loss = acc * inv;
}

__global__ void interpolate_kernel(const float* __restrict__ x, const float* __restrict__ xf,
                                   const float* __restrict__ alpha, int64_t per, int64_t total,
                                   bf16_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const float a = alpha[i / per];
    out[i] = f2bf(x[i] + a * (xf[i] - x[i]));
  }
}

// one block per sample: slopes[b] = sqrt(1e-4 + sum g^2)
__global__ __launch_bounds__(256) void gp_slopes_kernel(const float* __restrict__ g, int64_t per,
                                                        float* __restrict__ slopes) {
  __shared__ float sm4[4];
  const float* gp = g + (int64_t)blockIdx.x * per;
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < per; i += blockDim.x) s += gp[i] * gp[i];
  s = block_sum_256(s, sm4);
  if (threadIdx.x == 0) slopes[blockIdx.x] = sqrtf(1e-4f + s);
}
__global__ __launch_bounds__(256) void gp_mean_kernel(const float* __restrict__ slopes, int B,
                                                      float* __restrict__ penalty) {
  __shared__ float sm4[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) s += (slopes[i] - 1.f) * (slopes[i] - 1.f);
  s = block_sum_256(s, sm4);
  if (threadIdx.x == 0) *penalty = s / (float)B;
}
__global__ void gp_bwd_kernel(const float* __restrict__ g, const float* __restrict__ slopes,
                              const float* __restrict__ upstream, int B, int64_t per,
                              int64_t total, bf16_t* __restrict__ dg) {
  const float up = upstream ? *upstream : 1.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const float s = slopes[i / per];
    dg[i] = f2bf(up * (2.f / (float)B) * (s - 1.f) / s * g[i]);
  }
}

// ---------------------------------------------------------------------------------------------
// Multi-tensor Adam + EMA.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int find_entry(const cgAdamEntry* __restrict__ t, int n,
                                          int64_t chunk) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t[mid].chunk_begin <= chunk) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

// One chunk (CG_ADAM_CHUNK elements, the unit of the host's table) is spread over ADAM_SPLIT
// workgroups: the small networks (resnet_cifar D: 90 chunks) would otherwise run on a third of the
// CUs with a 64-iteration serial chain per thread.  16-byte accesses when every pointer of the
// entry is 16-byte aligned (separate tensors are; views into a flat all-reduce bucket need not be).
constexpr int ADAM_SPLIT = 4;
constexpr int ADAM_SUB = CG_ADAM_CHUNK / ADAM_SPLIT;

__device__ __forceinline__ void adam_one(float g, float& m, float& v, float& p, float beta1,
                                         float beta2, float eps, float lrt) {
  m = beta1 * m + (1.f - beta1) * g;
  v = beta2 * v + (1.f - beta2) * g * g;
  p = p - lrt * m / (sqrtf(v) + eps);
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const cgAdamEntry* __restrict__ table,
                                                         int n_entries, float lr, float beta1,
                                                         float beta2, float eps, float grad_scale,
                                                         const int64_t* __restrict__ step,
                                                         float ema_decay, int64_t ema_start) {
  __shared__ float s_lrt, s_omd;
  __shared__ int s_e;
  const int64_t chunk = blockIdx.x / ADAM_SPLIT;
  const int sub = blockIdx.x % ADAM_SPLIT;
  if (threadIdx.x == 0) s_e = find_entry(table, n_entries, chunk);
  __syncthreads();
  const cgAdamEntry e = table[s_e];
  const int64_t base = (chunk - e.chunk_begin) * CG_ADAM_CHUNK + (int64_t)sub * ADAM_SUB;
  const int64_t end = min(e.n, base + ADAM_SUB);
  const uintptr_t bits = (uintptr_t)e.grad | (uintptr_t)e.m | (uintptr_t)e.v | (uintptr_t)e.param |
                         (uintptr_t)e.ema;
  const bool vec = (bits & 15) == 0 && base < end;
  const int64_t vend = base + ((end - base) & ~(int64_t)3);
  // the first trip's operands leave before the step size is known: thread 0's two fp64 pow() calls
  // (a few microseconds of dependent arithmetic) run under the load latency instead of in front of it
  const int64_t ifirst = base + 4 * (int64_t)threadIdx.x;
  const bool pre = vec && ifirst < vend;
  float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f), m4 = g4, v4 = g4, p4 = g4;
  if (pre) {
    g4 = *reinterpret_cast<const float4*>(e.grad + ifirst);
    m4 = *reinterpret_cast<const float4*>(e.m + ifirst);
    v4 = *reinterpret_cast<const float4*>(e.v + ifirst);
    p4 = *reinterpret_cast<const float4*>(e.param + ifirst);
  }
  if (threadIdx.x == 0) {
    const int64_t t0 = step ? *step : 0;
    const double t = (double)(t0 + 1);
    s_lrt = (float)((double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
    const float d = (t0 >= ema_start) ? ema_decay : 0.f;
    s_omd = 1.f - d;
  }
  __syncthreads();
  const float lrt = s_lrt, omd = s_omd;
  if (base >= end) return;
  int64_t i0 = base;
  if (vec) {
    for (int64_t i = ifirst; i < vend; i += 1024) {
      if (i != ifirst) {
        g4 = *reinterpret_cast<const float4*>(e.grad + i);
        m4 = *reinterpret_cast<const float4*>(e.m + i);
        v4 = *reinterpret_cast<const float4*>(e.v + i);
        p4 = *reinterpret_cast<const float4*>(e.param + i);
      }
      adam_one(g4.x * grad_scale, m4.x, v4.x, p4.x, beta1, beta2, eps, lrt);
      adam_one(g4.y * grad_scale, m4.y, v4.y, p4.y, beta1, beta2, eps, lrt);
      adam_one(g4.z * grad_scale, m4.z, v4.z, p4.z, beta1, beta2, eps, lrt);
      adam_one(g4.w * grad_scale, m4.w, v4.w, p4.w, beta1, beta2, eps, lrt);
      *reinterpret_cast<float4*>(e.m + i) = m4;
      *reinterpret_cast<float4*>(e.v + i) = v4;
      *reinterpret_cast<float4*>(e.param + i) = p4;
      if (e.ema) {
        float4 s4 = *reinterpret_cast<const float4*>(e.ema + i);
        s4.x = s4.x - omd * (s4.x - p4.x);
        s4.y = s4.y - omd * (s4.y - p4.y);
        s4.z = s4.z - omd * (s4.z - p4.z);
        s4.w = s4.w - omd * (s4.w - p4.w);
        *reinterpret_cast<float4*>(e.ema + i) = s4;
      }
    }
    i0 = vend;
  }
  for (int64_t i = i0 + threadIdx.x; i < end; i += 256) {
    float m = e.m[i], v = e.v[i], p = e.param[i];
    adam_one(e.grad[i] * grad_scale, m, v, p, beta1, beta2, eps, lrt);
    e.m[i] = m;
    e.v[i] = v;
    e.param[i] = p;
    if (e.ema) {
      const float s = e.ema[i];
      e.ema[i] = s - omd * (s - p);
    }
  }
}

template <bool GATHER>
__global__ __launch_bounds__(256) void multi_copy_kernel(const cgAdamEntry* __restrict__ table,
                                                         const int64_t* __restrict__ offs,
                                                         int n_entries, float* flat) {
  __shared__ int s_e;
  if (threadIdx.x == 0) s_e = find_entry(table, n_entries, blockIdx.x);
  __syncthreads();
  const cgAdamEntry e = table[s_e];
  const int64_t base = ((int64_t)blockIdx.x - e.chunk_begin) * CG_ADAM_CHUNK;
  const int64_t end = min(e.n, base + CG_ADAM_CHUNK);
  float* f = flat + offs[s_e];
  float* g = const_cast<float*>(e.grad);
  for (int64_t i = base + threadIdx.x; i < end; i += 256) {
    if (GATHER) f[i] = g[i];
    else g[i] = f[i];
  }
}

__global__ void counter_add_kernel(int64_t* c, int64_t inc) { *c += inc; }

// ---------------------------------------------------------------------------------------------
// Philox4x32-10.
// ---------------------------------------------------------------------------------------------
struct U4 {
  uint32_t x, y, z, w;
};
__device__ __forceinline__ U4 philox(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    U4 n;
    n.x = hi1 ^ c.y ^ k0;
    n.y = lo1;
    n.z = hi0 ^ c.w ^ k1;
    n.w = lo0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}
__device__ __forceinline__ void make_key(uint64_t seed, uint32_t op_id, uint32_t* k0,
                                         uint32_t* k1) {
  *k0 = (uint32_t)seed ^ (op_id * 0x9E3779B1u);
  *k1 = (uint32_t)(seed >> 32) ^ 0x85EBCA6Bu;
}

__global__ void random_kernel(int kind, float lo, float hi, uint64_t seed, uint32_t op_id,
                              uint32_t stream_id, const int64_t* __restrict__ step_ptr,
                              float* __restrict__ out, int64_t n) {
  uint32_t k0, k1;
  make_key(seed, op_id, &k0, &k1);
  const uint64_t step = step_ptr ? (uint64_t)*step_ptr : 0ull;
  const int64_t nq = (n + 3) / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
    U4 c;
    c.x = (uint32_t)q;
    c.y = (uint32_t)((uint64_t)q >> 32) ^ (stream_id << 8);
    c.z = (uint32_t)step;
    c.w = (uint32_t)(step >> 32);
    const U4 r = philox(c, k0, k1);
    float v[4];
    if (kind == 0) {
      const float sc = hi - lo;
      v[0] = lo + sc * ((float)(r.x >> 8) * 5.9604644775390625e-8f);
      v[1] = lo + sc * ((float)(r.y >> 8) * 5.9604644775390625e-8f);
      v[2] = lo + sc * ((float)(r.z >> 8) * 5.9604644775390625e-8f);
      v[3] = lo + sc * ((float)(r.w >> 8) * 5.9604644775390625e-8f);
    } else {
      const float u1 = ((float)(r.x >> 8) + 1.f) * 5.9604644775390625e-8f;
      const float u2 = (float)(r.y >> 8) * 5.9604644775390625e-8f;
      const float u3 = ((float)(r.z >> 8) + 1.f) * 5.9604644775390625e-8f;
      const float u4 = (float)(r.w >> 8) * 5.9604644775390625e-8f;
      const float ra = sqrtf(-2.f * logf(u1)), rb = sqrtf(-2.f * logf(u3));
      const float ta = 6.283185307179586f * u2, tb = 6.283185307179586f * u4;
      v[0] = lo + hi * ra * cosf(ta);
      v[1] = lo + hi * ra * sinf(ta);
      v[2] = lo + hi * rb * cosf(tb);
      v[3] = lo + hi * rb * sinf(tb);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (q * 4 + e < n) out[q * 4 + e] = v[e];
  }
}

__global__ void random_labels_kernel(int K, uint64_t seed, uint32_t op_id, uint32_t stream_id,
                                     const int64_t* __restrict__ step_ptr,
                                     int32_t* __restrict__ out, int64_t n) {
  uint32_t k0, k1;
  make_key(seed, op_id, &k0, &k1);
  const uint64_t step = step_ptr ? (uint64_t)*step_ptr : 0ull;
  const int64_t nq = (n + 3) / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
    U4 c;
    c.x = (uint32_t)q;
    c.y = (uint32_t)((uint64_t)q >> 32) ^ (stream_id << 8);
    c.z = (uint32_t)step;
    c.w = (uint32_t)(step >> 32);
    const U4 r = philox(c, k0, k1);
    const uint32_t rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (q * 4 + e < n) out[q * 4 + e] = (int32_t)(((uint64_t)rv[e] * (uint64_t)K) >> 32);
  }
}

inline int grid_cap(int64_t work) {
  int64_t b = (work + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace


// ---- sums for the DRAGAN / L2 penalties (penalty_lib.py:33-56,85-102): two-stage, deterministic ----
constexpr int MOM_BLOCKS = 512;
__global__ __launch_bounds__(256) void moments_part_kernel(const float* __restrict__ x, int64_t n,
                                                           float* __restrict__ part) {
  __shared__ float sm[4];
  float s = 0.f, q = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    s += v;
    q += v * v;
  }
  s = block_sum_256(s, sm);
  q = block_sum_256(q, sm);
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2] = s;
    part[blockIdx.x * 2 + 1] = q;
  }
}
__global__ __launch_bounds__(256) void moments_final_kernel(const float* __restrict__ part, int nb,
                                                            float* __restrict__ out) {
  __shared__ float sm[4];
  float s = 0.f, q = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) {
    s += part[i * 2];
    q += part[i * 2 + 1];
  }
  s = block_sum_256(s, sm);
  q = block_sum_256(q, sm);
  if (threadIdx.x == 0) {
    out[0] = s;
    out[1] = q;
  }
}
// x_noisy = clip(x + std * (u - 0.5), 0, 1) * a + b -> bf16; std from the global sums (biased
// variance over ALL elements, tf.nn.moments over every axis)
__global__ void dragan_perturb_kernel(const float* __restrict__ x, const float* __restrict__ u,
                                      const float* __restrict__ sums, float inv_n, float a, float b,
                                      int64_t n, bf16_t* __restrict__ out) {
  const float mean = sums[0] * inv_n;
  const float var = fmaxf(sums[1] * inv_n - mean * mean, 0.f);
  const float sd = sqrtf(var);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float v = x[i] + sd * (u[i] - 0.5f);
    v = fminf(fmaxf(v, 0.f), 1.f);
    out[i] = f2bf(v * a + b);
  }
}

extern "C" int cg_gan_loss(int kind, const float* logits, int B, float* losses, float* dlogits_d,
                           float* dlogits_g, cgStream stream) {
  if (!logits || !losses || B <= 0 || kind < 0 || kind > 3)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_gan_loss: bad argument");
  gan_loss_kernel<<<1, 256, 0, (hipStream_t)stream>>>(kind, logits, B, losses, dlogits_d,
                                                      dlogits_g);
  CG_CHECK_LAUNCH("cg_gan_loss");
  return CG_OK;
}

extern "C" int cg_softmax_xent_eps(const float* logits, const int32_t* labels, int n, int k,
                                   float eps, float* loss, float* dlogits, cgStream stream) {
  if (!logits || !labels || !loss || !dlogits || n <= 0 || k <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_softmax_xent_eps: bad argument");
  softmax_xent_eps_kernel<<<1, 256, 0, (hipStream_t)stream>>>(logits, labels, n, k, eps, loss,
                                                              dlogits);
  CG_CHECK_LAUNCH("cg_softmax_xent_eps");
  return CG_OK;
}

extern "C" int cg_s3gan_labels(const float* aux_logits, const void* y, int n, int k, int soft,
                               void* y_out, float* is_label_available, cgStream stream) {
  if (!y || !y_out || !is_label_available || n <= 0 || k <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_s3gan_labels: bad argument");
  s3gan_labels_kernel<<<n, 256, 0, (hipStream_t)stream>>>(aux_logits, (const bf16_t*)y, k, soft,
                                                          (bf16_t*)y_out, is_label_available);
  CG_CHECK_LAUNCH("cg_s3gan_labels");
  return CG_OK;
}

extern "C" int cg_softmax_xent_weighted(const float* logits, const void* labels,
                                        const float* weights, int n, int k, float* loss,
                                        float* dlogits, cgStream stream) {
  if (!logits || !labels || !weights || !loss || !dlogits || n <= 0 || k <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_softmax_xent_weighted: bad argument");
  softmax_xent_weighted_kernel<<<1, 256, 0, (hipStream_t)stream>>>(
      logits, (const bf16_t*)labels, weights, n, k, loss, dlogits);
  CG_CHECK_LAUNCH("cg_softmax_xent_weighted");
  return CG_OK;
}

extern "C" int cg_interpolate(const float* x, const float* x_fake, const float* alpha, int B,
                              int64_t per, void* out, cgStream stream) {
  if (!x || !x_fake || !alpha || !out || B <= 0 || per <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_interpolate: bad argument");
  const int64_t total = (int64_t)B * per;
  interpolate_kernel<<<grid_cap(total), 256, 0, (hipStream_t)stream>>>(x, x_fake, alpha, per,
                                                                       total, (bf16_t*)out);
  CG_CHECK_LAUNCH("cg_interpolate");
  return CG_OK;
}

extern "C" int cg_gradient_penalty(const float* g, int B, int64_t per, float* slopes,
                                   float* penalty, cgStream stream) {
  if (!g || !slopes || !penalty || B <= 0 || per <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_gradient_penalty: bad argument");
  hipStream_t st = (hipStream_t)stream;
  gp_slopes_kernel<<<B, 256, 0, st>>>(g, per, slopes);
  CG_CHECK_LAUNCH("cg_gradient_penalty(slopes)");
  gp_mean_kernel<<<1, 256, 0, st>>>(slopes, B, penalty);
  CG_CHECK_LAUNCH("cg_gradient_penalty(mean)");
  return CG_OK;
}

extern "C" int cg_gradient_penalty_bwd(const float* g, const float* slopes, const float* upstream,
                                       int B, int64_t per, void* dg, cgStream stream) {
  if (!g || !slopes || !dg || B <= 0 || per <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_gradient_penalty_bwd: bad argument");
  const int64_t total = (int64_t)B * per;
  gp_bwd_kernel<<<grid_cap(total), 256, 0, (hipStream_t)stream>>>(g, slopes, upstream, B, per,
                                                                  total, (bf16_t*)dg);
  CG_CHECK_LAUNCH("cg_gradient_penalty_bwd");
  return CG_OK;
}

extern "C" int cg_adam_multi(const cgAdamEntry* table, int n_entries, int64_t total_chunks,
                             float lr, float beta1, float beta2, float eps, float grad_scale,
                             const int64_t* step, float ema_decay, int64_t ema_start_step,
                             cgStream stream) {
  if (!table || n_entries <= 0 || total_chunks <= 0 || total_chunks >= (1ll << 31))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_adam_multi: bad argument");
  if (total_chunks * ADAM_SPLIT >= (1ll << 31)) CG_FAIL(CG_ERR_BAD_ARG, "cg_adam_multi: too many chunks");
  adam_multi_kernel<<<(int)(total_chunks * ADAM_SPLIT), 256, 0, (hipStream_t)stream>>>(
      table, n_entries, lr, beta1, beta2, eps, grad_scale, step, ema_decay, ema_start_step);
  CG_CHECK_LAUNCH("cg_adam_multi");
  return CG_OK;
}

extern "C" int cg_counter_add(int64_t* counter, int64_t inc, cgStream stream) {
  if (!counter) CG_FAIL(CG_ERR_BAD_ARG, "cg_counter_add: null counter");
  counter_add_kernel<<<1, 1, 0, (hipStream_t)stream>>>(counter, inc);
  CG_CHECK_LAUNCH("cg_counter_add");
  return CG_OK;
}

extern "C" int cg_multi_gather(const cgAdamEntry* table, const int64_t* flat_offsets,
                               int n_entries, int64_t total_chunks, float* flat,
                               cgStream stream) {
  if (!table || !flat_offsets || !flat || n_entries <= 0 || total_chunks <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_multi_gather: bad argument");
  multi_copy_kernel<true><<<(int)total_chunks, 256, 0, (hipStream_t)stream>>>(
      table, flat_offsets, n_entries, flat);
  CG_CHECK_LAUNCH("cg_multi_gather");
  return CG_OK;
}
extern "C" int cg_multi_scatter(const cgAdamEntry* table, const int64_t* flat_offsets,
                                int n_entries, int64_t total_chunks, const float* flat,
                                cgStream stream) {
  if (!table || !flat_offsets || !flat || n_entries <= 0 || total_chunks <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_multi_scatter: bad argument");
  multi_copy_kernel<false><<<(int)total_chunks, 256, 0, (hipStream_t)stream>>>(
      table, flat_offsets, n_entries, const_cast<float*>(flat));
  CG_CHECK_LAUNCH("cg_multi_scatter");
  return CG_OK;
}

extern "C" int cg_random(int kind, float lo, float hi, uint64_t seed, uint32_t op_id,
                         uint32_t stream_id, const int64_t* step_ptr, float* out, int64_t n,
                         cgStream stream) {
  if (!out || n < 0 || (kind != 0 && kind != 1)) CG_FAIL(CG_ERR_BAD_ARG, "cg_random: bad argument");
  if (n == 0) return CG_OK;
  random_kernel<<<grid_cap((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(
      kind, lo, hi, seed, op_id, stream_id, step_ptr, out, n);
  CG_CHECK_LAUNCH("cg_random");
  return CG_OK;
}
extern "C" int cg_random_labels(int K, uint64_t seed, uint32_t op_id, uint32_t stream_id,
                                const int64_t* step_ptr, int32_t* out, int64_t n,
                                cgStream stream) {
  if (!out || n < 0 || K <= 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_random_labels: bad argument");
  if (n == 0) return CG_OK;
  random_labels_kernel<<<grid_cap((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(
      K, seed, op_id, stream_id, step_ptr, out, n);
  CG_CHECK_LAUNCH("cg_random_labels");
  return CG_OK;
}

extern "C" size_t cg_moments_workspace_bytes(void) { return MOM_BLOCKS * 2 * sizeof(float); }

extern "C" int cg_moments_f32(const float* x, int64_t n, float* sums, void* ws, size_t ws_bytes,
                              cgStream stream) {
  if (!x || !sums || n <= 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_moments_f32: bad argument");
  if (!ws || ws_bytes < cg_moments_workspace_bytes())
    CG_FAIL(CG_ERR_WORKSPACE, "cg_moments_f32: workspace too small");
  int nb = (int)((n + 255) / 256);
  if (nb > MOM_BLOCKS) nb = MOM_BLOCKS;
  hipStream_t st = (hipStream_t)stream;
  moments_part_kernel<<<nb, 256, 0, st>>>(x, n, (float*)ws);
  moments_final_kernel<<<1, 256, 0, st>>>((const float*)ws, nb, sums);
  CG_CHECK_LAUNCH("cg_moments_f32");
  return CG_OK;
}

extern "C" int cg_dragan_perturb(const float* x, const float* u, const float* sums, int64_t n,
                                 float a, float b, void* out_bf16, cgStream stream) {
  if (!x || !u || !sums || !out_bf16 || n <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_dragan_perturb: bad argument");
  dragan_perturb_kernel<<<grid_cap(n), 256, 0, (hipStream_t)stream>>>(x, u, sums, 1.0f / (float)n,
                                                                      a, b, n, (bf16_t*)out_bf16);
  CG_CHECK_LAUNCH("cg_dragan_perturb");
  return CG_OK;
}
