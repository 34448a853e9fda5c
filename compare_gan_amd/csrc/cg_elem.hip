// Element-wise, pooling and small reduction kernels of the G/D graphs (HBM-bound: 16-byte
// vector accesses, grid-stride loops capped at 2048 blocks).  Contracts: include/cgamd.h.
#include "cg_common.h"

namespace {

constexpr int kBlock = 256;
inline int grid_for(int64_t work_items) {
  int64_t b = (work_items + kBlock - 1) / kBlock;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

union V8 {
  uint4 q;
  bf16_t h[8];
};

// out[i] = f(a[i], b[i]) over bf16 arrays (b optional), 8 elements per lane when aligned.
template <typename F>
__global__ void ew2_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                           bf16_t* __restrict__ out, int64_t n, F f) {
  const bool aligned = ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)out)) & 15) == 0;
  const int64_t nv = aligned ? n / 8 : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    V8 va, vb, vo;
    va.q = reinterpret_cast<const uint4*>(a)[i];
    if (b) vb.q = reinterpret_cast<const uint4*>(b)[i];
#pragma unroll
    for (int e = 0; e < 8; ++e) vo.h[e] = f2bf(f(bf2f(va.h[e]), b ? bf2f(vb.h[e]) : 0.f));
    reinterpret_cast<uint4*>(out)[i] = vo.q;
  }
  for (int64_t i = nv * 8 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = f2bf(f(bf2f(a[i]), b ? bf2f(b[i]) : 0.f));
}

struct LreluF {
  float slope;
  __device__ float operator()(float x, float) const { return x > 0.f ? x : slope * x; }
};
struct LreluBwdF {
  float slope;
  __device__ float operator()(float x, float dy) const { return x > 0.f ? dy : slope * dy; }
};
struct AxpbyF {
  float alpha, beta;
  __device__ float operator()(float a, float b) const { return alpha * a + beta * b; }
};

// ---- pooling -------------------------------------------------------------------------------
// one thread handles 8 channels of one output pixel (C % 8 == 0) or 1 channel otherwise.
template <bool VEC, bool MAXP>
__global__ void pool2_kernel(const bf16_t* __restrict__ x, int N, int H, int W, int C,
                             bf16_t* __restrict__ y) {
  const int Ho = H / 2, Wo = W / 2;
  const int CV = VEC ? C / 8 : C;
  const int64_t total = (int64_t)N * Ho * Wo * CV;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int cv = (int)(i % CV);
    int64_t p = i / CV;
    const int ow = (int)(p % Wo);
    p /= Wo;
    const int oh = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const int64_t base = (((int64_t)n * H + oh * 2) * W + ow * 2) * C;
    if (VEC) {
      V8 v[4], o;
      const bf16_t* xp = x + base + cv * 8;
      v[0].q = *reinterpret_cast<const uint4*>(xp);
      v[1].q = *reinterpret_cast<const uint4*>(xp + C);
      v[2].q = *reinterpret_cast<const uint4*>(xp + (int64_t)W * C);
      v[3].q = *reinterpret_cast<const uint4*>(xp + (int64_t)W * C + C);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = bf2f(v[0].h[e]), b = bf2f(v[1].h[e]), c = bf2f(v[2].h[e]),
                    d = bf2f(v[3].h[e]);
        o.h[e] = MAXP ? f2bf(fmaxf(fmaxf(a, b), fmaxf(c, d))) : f2bf(0.25f * (a + b + c + d));
      }
      *reinterpret_cast<uint4*>(y + (((int64_t)n * Ho + oh) * Wo + ow) * C + cv * 8) = o.q;
    } else {
      const bf16_t* xp = x + base + cv;
      const float a = bf2f(xp[0]), b = bf2f(xp[C]), c = bf2f(xp[(int64_t)W * C]),
                  d = bf2f(xp[(int64_t)W * C + C]);
      y[(((int64_t)n * Ho + oh) * Wo + ow) * C + cv] =
          MAXP ? f2bf(fmaxf(fmaxf(a, b), fmaxf(c, d))) : f2bf(0.25f * (a + b + c + d));
    }
  }
}

// gradient: avg: dx = dy/4 broadcast.  max: route dy to the FIRST maximal tap (row-major order).
template <bool MAXP>
__global__ void pool2_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                 int N, int H, int W, int C, bf16_t* __restrict__ dx) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)N * Ho * Wo * C;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C);
    int64_t p = i / C;
    const int ow = (int)(p % Wo);
    p /= Wo;
    const int oh = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const int64_t base = (((int64_t)n * H + oh * 2) * W + ow * 2) * C + c;
    const int64_t o1 = C, o2 = (int64_t)W * C, o3 = (int64_t)W * C + C;
    const float g = bf2f(dy[i]);
    if (MAXP) {
      const float a = bf2f(x[base]), b = bf2f(x[base + o1]), cc = bf2f(x[base + o2]),
                  d = bf2f(x[base + o3]);
      const float m = fmaxf(fmaxf(a, b), fmaxf(cc, d));
      int sel = 3;
      if (a == m) sel = 0;
      else if (b == m) sel = 1;
      else if (cc == m) sel = 2;
      const bf16_t z = 0, gv = dy[i];
      dx[base] = sel == 0 ? gv : z;
      dx[base + o1] = sel == 1 ? gv : z;
      dx[base + o2] = sel == 2 ? gv : z;
      dx[base + o3] = sel == 3 ? gv : z;
    } else {
      const bf16_t q = f2bf(0.25f * g);
      dx[base] = q;
      dx[base + o1] = q;
      dx[base + o2] = q;
      dx[base + o3] = q;
    }
  }
}

// the same, 8 channels per thread (C % 8 == 0): one 16-byte gradient load, four 16-byte stores (and
// four 16-byte loads of x for the maximum): the scalar form moved 2 bytes per thread and instruction
// (2.1-2.8 TB/s algorithmic; profiles/r06_stream_rates.txt)
template <bool MAXP>
__global__ __launch_bounds__(256) void pool2_bwd_vec8_kernel(const bf16_t* __restrict__ x,
                                                             const bf16_t* __restrict__ dy, int N,
                                                             int H, int W, int C,
                                                             bf16_t* __restrict__ dx) {
  const int Ho = H / 2, Wo = W / 2, CV = C / 8;
  const int64_t total = (int64_t)N * Ho * Wo * CV;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int cv = (int)(i % CV);
    int64_t p = i / CV;
    const int ow = (int)(p % Wo);
    p /= Wo;
    const int oh = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const int64_t base = (((int64_t)n * H + oh * 2) * W + ow * 2) * C + cv * 8;
    const int64_t o1 = C, o2 = (int64_t)W * C, o3 = (int64_t)W * C + C;
    V8 g;
    g.q = *reinterpret_cast<const uint4*>(dy + i * 8);
    V8 r0, r1, r2, r3;
    if (MAXP) {
      V8 a, b, c, d;
      a.q = *reinterpret_cast<const uint4*>(x + base);
      b.q = *reinterpret_cast<const uint4*>(x + base + o1);
      c.q = *reinterpret_cast<const uint4*>(x + base + o2);
      d.q = *reinterpret_cast<const uint4*>(x + base + o3);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float fa = bf2f(a.h[e]), fb = bf2f(b.h[e]), fc = bf2f(c.h[e]), fd = bf2f(d.h[e]);
        const float m = fmaxf(fmaxf(fa, fb), fmaxf(fc, fd));
        int sel = 3;
        if (fa == m) sel = 0;
        else if (fb == m) sel = 1;
        else if (fc == m) sel = 2;
        const bf16_t z = 0, gv = g.h[e];
        r0.h[e] = sel == 0 ? gv : z;
        r1.h[e] = sel == 1 ? gv : z;
        r2.h[e] = sel == 2 ? gv : z;
        r3.h[e] = sel == 3 ? gv : z;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) r0.h[e] = f2bf(0.25f * bf2f(g.h[e]));
      r1 = r0; r2 = r0; r3 = r0;
    }
    *reinterpret_cast<uint4*>(dx + base) = r0.q;
    *reinterpret_cast<uint4*>(dx + base + o1) = r1.q;
    *reinterpret_cast<uint4*>(dx + base + o2) = r2.q;
    *reinterpret_cast<uint4*>(dx + base + o3) = r3.q;
  }
}

// ---- spatial reduce over HW per (n, c) -----------------------------------------------------
// block = 256 threads handles one n and 64 channels; 4 waves split HW; LDS combine.
__global__ void spatial_reduce_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gate,
                                      int HW, int C, float scale, bf16_t* __restrict__ out) {
  __shared__ float part[4][64];
  const int n = blockIdx.y;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C) {
    const bf16_t* xp = x + (int64_t)n * HW * C + c;
    const bf16_t* gp = gate ? gate + (int64_t)n * HW * C + c : nullptr;
    for (int p = w; p < HW; p += 4) {
      float v = bf2f(xp[(int64_t)p * C]);
      if (gp && !(bf2f(gp[(int64_t)p * C]) > 0.f)) v = 0.f;
      s += v;
    }
  }
  part[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < C) {
    const int l = threadIdx.x & 63;
    out[(int64_t)n * C + c] = f2bf(scale * (part[0][l] + part[1][l] + part[2][l] + part[3][l]));
  }
}

__global__ void spatial_reduce_bwd_kernel(const bf16_t* __restrict__ gate,
                                          const bf16_t* __restrict__ dout, int HW, int C,
                                          float scale, int64_t total,
                                          bf16_t* __restrict__ dx) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C);
    const int64_t n = i / ((int64_t)HW * C);
    float g = scale * bf2f(dout[n * C + c]);
    if (gate && !(bf2f(gate[i]) > 0.f)) g = 0.f;
    dx[i] = f2bf(g);
  }
}

// ---- heads ---------------------------------------------------------------------------------
__global__ void head_kernel(const float* __restrict__ x, int kind, float* __restrict__ y,
                            int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = x[i];
    y[i] = kind == 0 ? 1.f / (1.f + expf(-v)) : 0.5f * tanhf(v) + 0.5f;
  }
}
__global__ void head_bwd_kernel(const float* __restrict__ y, int kind, const void* dy,
                                int dy_is_f32, bf16_t* __restrict__ dx, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float yy = y[i];
    const float g = dy_is_f32 ? reinterpret_cast<const float*>(dy)[i]
                              : bf2f(reinterpret_cast<const bf16_t*>(dy)[i]);
    // sigmoid' = y(1-y);  (0.5 tanh + 0.5)' = 0.5 (1 - tanh^2) = 2 y (1 - y)
    const float d = kind == 0 ? yy * (1.f - yy) : 2.f * yy * (1.f - yy);
    dx[i] = f2bf(g * d);
  }
}

__global__ void cast_f2b_kernel(const float* __restrict__ x, float a, float b,
                                bf16_t* __restrict__ y, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = f2bf(x[i] * a + b);
}
__global__ void cast_b2f_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = bf2f(x[i]);
}

// ---- column sums: [rows, C] bf16 -> fp32 [C] -------------------------------------------------
// grid (ceil(C/64), splits): each block sums a row slab for 64 columns; partials -> ws; reduce.
__global__ void colsum_part_kernel(const bf16_t* __restrict__ x, int64_t rows, int C,
                                   int64_t rows_per_split, float* __restrict__ part) {
  __shared__ float sm[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_split;
  const int64_t r1 = min(rows, r0 + rows_per_split);
  float s = 0.f;
  if (c < C)
    for (int64_t r = r0 + w; r < r1; r += 4) s += bf2f(x[r * C + c]);
  sm[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < C) {
    const int l = threadIdx.x & 63;
    part[(int64_t)blockIdx.y * C + c] = sm[0][l] + sm[1][l] + sm[2][l] + sm[3][l];
  }
}
// one block per 64 columns: 16 split-lanes per column walk the partials (four loads in flight each;
// one lane per column walking up to 512 splits alone was 16 us of load latency), LDS combines them
// in a fixed order (deterministic)
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ part,
                                                            int splits, int C,
                                                            float* __restrict__ out) {
  __shared__ float sm[16][64];
  const int l = threadIdx.x & 63, zl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + l;
  float s = 0.f;
  if (c < C) {
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = zl;
    for (; z + 48 < splits; z += 64) {
      const float a0 = part[(int64_t)z * C + c], a1 = part[(int64_t)(z + 16) * C + c];
      const float a2 = part[(int64_t)(z + 32) * C + c], a3 = part[(int64_t)(z + 48) * C + c];
      s += a0; s1 += a1; s2 += a2; s3 += a3;
    }
    for (; z < splits; z += 16) s += part[(int64_t)z * C + c];
    s = (s + s1) + (s2 + s3);
  }
  sm[zl][l] = s;
  __syncthreads();
  if (zl == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += sm[r][l];
    out[c] = t;
  }
}
inline int colsum_splits(int64_t rows, int C) {
  const int ct = cdiv(C, 64);
  int s = cdiv(1024, ct);
  const int64_t maxs = rows / 64 > 0 ? rows / 64 : 1;
  if (s > maxs) s = (int)maxs;
  if (s < 1) s = 1;
  return s;
}

// ---- row dot ---------------------------------------------------------------------------------
__global__ void rowdot_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, int C,
                              float* __restrict__ out) {
  __shared__ float sm4[4];
  const int r = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    s += bf2f(a[(int64_t)r * C + c]) * bf2f(b[(int64_t)r * C + c]);
  s = block_sum_256(s, sm4);
  if (threadIdx.x == 0) out[r] = s;
}
__global__ void rowdot_bwd_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                  const float* __restrict__ dout, int C, int64_t total,
                                  bf16_t* __restrict__ da, bf16_t* __restrict__ db) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const float g = dout[i / C];
    if (da) da[i] = f2bf(g * bf2f(b[i]));
    if (db) db[i] = f2bf(g * bf2f(a[i]));
  }
}

__global__ void one_hot_kernel(const int32_t* __restrict__ labels, int K, int64_t total,
                               bf16_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int b = (int)(i / K), k = (int)(i - (int64_t)b * K);
    out[i] = labels[b] == k ? (bf16_t)0x3f80 : (bf16_t)0;
  }
}

// ---- general pooling (Inception graph) -----------------------------------------------------
__global__ void pool2d_kernel(const bf16_t* __restrict__ x, int N, int H, int W, int C, int k,
                              int s, int p, int kind, int Ho, int Wo, bf16_t* __restrict__ y) {
  const int64_t total = (int64_t)N * Ho * Wo * C;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C);
    int64_t q = i / C;
    const int ow = (int)(q % Wo);
    q /= Wo;
    const int oh = (int)(q % Ho);
    const int n = (int)(q / Ho);
    float acc = kind == 0 ? -3.4e38f : 0.f;
    int cnt = 0;
    for (int r = 0; r < k; ++r) {
      const int ih = oh * s - p + r;
      if (ih < 0 || ih >= H) continue;
      for (int t = 0; t < k; ++t) {
        const int iw = ow * s - p + t;
        if (iw < 0 || iw >= W) continue;
        const float v = bf2f(x[(((int64_t)n * H + ih) * W + iw) * C + c]);
        acc = kind == 0 ? fmaxf(acc, v) : acc + v;
        ++cnt;
      }
    }
    y[i] = f2bf(kind == 0 ? acc : acc / (float)(cnt > 0 ? cnt : 1));
  }
}

// 8 channels per thread (16-byte loads / stores), C % 8 == 0: the Inception stages pool 64 ... 2048
// channels and are pure HBM streams; the scalar form above ran at a tenth of the bandwidth
// (1 ms per call on 512 x 35 x 35 x 288: as much time as all convolutions of the network).
// x_ld / y_ld: elements between consecutive pixels of x / y (C when dense; wider for channel slices of
// a concatenation, cg_pool2d_ld); bias (fp32 [C] or NULL) and relu finish the pooled value.
__global__ __launch_bounds__(256) void pool2d_vec8_kernel(const bf16_t* __restrict__ x, int N,
                                                          int H, int W, int C8, int k, int s,
                                                          int p, int kind, int Ho, int Wo,
                                                          bf16_t* __restrict__ y, int x_ld, int y_ld,
                                                          const float* __restrict__ bias,
                                                          int relu) {
  const int64_t total = (int64_t)N * Ho * Wo * C8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c8 = (int)(i % C8);
    int64_t q = i / C8;
    const int ow = (int)(q % Wo);
    q /= Wo;
    const int oh = (int)(q % Ho);
    const int n = (int)(q / Ho);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = kind == 0 ? -3.4e38f : 0.f;
    int cnt = 0;
    if (k == 3) {
      // the 3x3 windows of the Inception stages: nine unconditional loads from clamped positions,
      // selected afterwards (guarded loads in the tap loop are nine serial round trips)
      uint4 raw[9];
      bool ok[9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int ih = oh * s - p + r, iw = ow * s - p + t;
          ok[r * 3 + t] = ih >= 0 && ih < H && iw >= 0 && iw < W;
          const int ihc = min(max(ih, 0), H - 1), iwc = min(max(iw, 0), W - 1);
          raw[r * 3 + t] = *reinterpret_cast<const uint4*>(
              x + (((int64_t)n * H + ihc) * W + iwc) * x_ld + c8 * 8);
        }
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        float v[8];
        unpack8_bf16(raw[j], v);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          acc[e] = kind == 0 ? (ok[j] ? fmaxf(acc[e], v[e]) : acc[e]) : acc[e] + (ok[j] ? v[e] : 0.f);
        cnt += ok[j] ? 1 : 0;
      }
    } else
    for (int r = 0; r < k; ++r) {
      const int ih = oh * s - p + r;
      if (ih < 0 || ih >= H) continue;
      for (int t = 0; t < k; ++t) {
        const int iw = ow * s - p + t;
        if (iw < 0 || iw >= W) continue;
        const uint4 raw = *reinterpret_cast<const uint4*>(
            x + (((int64_t)n * H + ih) * W + iw) * x_ld + c8 * 8);
        float v[8];
        unpack8_bf16(raw, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = kind == 0 ? fmaxf(acc[e], v[e]) : acc[e] + v[e];
        ++cnt;
      }
    }
    if (kind != 0) {
      const float d = (float)(cnt > 0 ? cnt : 1);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = acc[e] / d;
    }
    if (bias) {   // kernel-uniform
      const float4 b0 = *reinterpret_cast<const float4*>(bias + c8 * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(bias + c8 * 8 + 4);
      acc[0] += b0.x; acc[1] += b0.y; acc[2] += b0.z; acc[3] += b0.w;
      acc[4] += b1.x; acc[5] += b1.y; acc[6] += b1.z; acc[7] += b1.w;
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaxf(acc[e], 0.f);
    }
    *reinterpret_cast<uint4*>(y + (i / C8) * y_ld + c8 * 8) = pack8_bf16(acc);
  }
}

// TF1 legacy bilinear resize (align_corners=False, half_pixel_centers=False):
//   src = dst * (in / out);  lo = floor(src), hi = min(lo+1, in-1), frac = src - lo.
// A thread produces 8 consecutive output elements (one 16-byte store; the index decode -- five 64-bit
// divisions -- once per 8 and incremented in between): the element-per-thread form wrote the 275 MB of a
// 512-image batch with 2-byte stores at 0.46 TB/s (0.6 ms per batch).  Same arithmetic per element.
__device__ __forceinline__ float inception_resize_at(const float* __restrict__ x, int H, int W, int C,
                                                     float sh, float sw, int n, int oh, int ow,
                                                     int c) {
  const float fy = oh * sh, fx = ow * sw;
  const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
  const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
  const float wy = fy - y0, wx = fx - x0;
  const float* xp = x + (int64_t)n * H * W * C + c;
  const float tl = xp[((int64_t)y0 * W + x0) * C], tr = xp[((int64_t)y0 * W + x1) * C];
  const float bl = xp[((int64_t)y1 * W + x0) * C], br = xp[((int64_t)y1 * W + x1) * C];
  const float top = tl + (tr - tl) * wx, bot = bl + (br - bl) * wx;
  const float v = top + (bot - top) * wy;
  return (v - 128.f) / 128.f;
}
__global__ void inception_preprocess_kernel(const float* __restrict__ x, int N, int H, int W,
                                            int C, int Ho, int Wo, bf16_t* __restrict__ y) {
  const int64_t total = (int64_t)N * Ho * Wo * C;
  const int64_t total8 = total / 8;
  const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i8 = gid; i8 < total8; i8 += stride) {
    const int64_t i = i8 * 8;
    int c = (int)(i % C);
    int64_t q = i / C;
    int ow = (int)(q % Wo);
    q /= Wo;
    int oh = (int)(q % Ho);
    int n = (int)(q / Ho);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = inception_resize_at(x, H, W, C, sh, sw, n, oh, ow, c);
      if (++c == C) {
        c = 0;
        if (++ow == Wo) {
          ow = 0;
          if (++oh == Ho) {
            oh = 0;
            ++n;
          }
        }
      }
    }
    *reinterpret_cast<uint4*>(y + i) = pack8_bf16(v);
  }
  // the last total % 8 elements
  const int64_t i = total8 * 8 + gid;
  if (i < total) {
    const int c = (int)(i % C);
    int64_t q = i / C;
    const int ow = (int)(q % Wo);
    q /= Wo;
    const int oh = (int)(q % Ho);
    const int n = (int)(q / Ho);
    y[i] = f2bf(inception_resize_at(x, H, W, C, sh, sw, n, oh, ow, c));
  }
}

__global__ void axpby_f32_kernel(const float* __restrict__ a, float alpha,
                                 const float* __restrict__ b, float beta, float* __restrict__ out,
                                 int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}
// out = x * scale; *nan_count += NaNs of x (one atomic per wave that saw one)
__global__ __launch_bounds__(256) void scale_count_nan_kernel(const float* __restrict__ x,
                                                              float scale, float* __restrict__ out,
                                                              int64_t n, int* __restrict__ nan_count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int bad = 0;
  if ((((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      float4 v = reinterpret_cast<const float4*>(x)[i];
      bad += (v.x != v.x) + (v.y != v.y) + (v.z != v.z) + (v.w != v.w);
      v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
      reinterpret_cast<float4*>(out)[i] = v;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
      const float v = x[i];
      bad += v != v;
      out[i] = v * scale;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
      const float v = x[i];
      bad += v != v;
      out[i] = v * scale;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(nan_count, bad);
}
__global__ void axpy_dev_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ o,
                                const float* __restrict__ sigma, bf16_t* __restrict__ out,
                                int64_t n) {
  const float s = *sigma;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = f2bf((x ? bf2f(x[i]) : 0.f) + s * bf2f(o[i]));
}
__global__ __launch_bounds__(256) void dot_bf16_part_kernel(const bf16_t* __restrict__ a,
                                                            const bf16_t* __restrict__ b,
                                                            int64_t n, float* __restrict__ part) {
  __shared__ float sm4[4];
  float s = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    s += bf2f(a[i]) * bf2f(b[i]);
  s = block_sum_256(s, sm4);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void dot_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
  float s = 0.f;
  for (int i = 0; i < nb; ++i) s += part[i];
  *out = s;
}
inline int dot_blocks(int64_t n) {
  int64_t b = (n + 2047) / 2048;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

#define CG_NONNEG(n, who) \
  if ((n) < 0) CG_FAIL(CG_ERR_BAD_ARG, who ": negative size")

extern "C" int cg_lrelu(const void* x, float slope, void* y, int64_t n, cgStream stream) {
  CG_NONNEG(n, "cg_lrelu");
  if (n == 0) return CG_OK;
  if (!x || !y) CG_FAIL(CG_ERR_BAD_ARG, "cg_lrelu: null pointer");
  ew2_kernel<<<grid_for(n / 8 + 1), kBlock, 0, (hipStream_t)stream>>>(
      (const bf16_t*)x, (const bf16_t*)nullptr, (bf16_t*)y, n, LreluF{slope});
  CG_CHECK_LAUNCH("cg_lrelu");
  return CG_OK;
}
extern "C" int cg_lrelu_bwd(const void* x, const void* dy, float slope, void* dx, int64_t n,
                            cgStream stream) {
  CG_NONNEG(n, "cg_lrelu_bwd");
  if (n == 0) return CG_OK;
  if (!x || !dy || !dx) CG_FAIL(CG_ERR_BAD_ARG, "cg_lrelu_bwd: null pointer");
  ew2_kernel<<<grid_for(n / 8 + 1), kBlock, 0, (hipStream_t)stream>>>(
      (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n, LreluBwdF{slope});
  CG_CHECK_LAUNCH("cg_lrelu_bwd");
  return CG_OK;
}
extern "C" int cg_axpby(const void* a, float alpha, const void* b, float beta, void* out,
                        int64_t n, cgStream stream) {
  CG_NONNEG(n, "cg_axpby");
  if (n == 0) return CG_OK;
  if (!a || !out) CG_FAIL(CG_ERR_BAD_ARG, "cg_axpby: null pointer");
  ew2_kernel<<<grid_for(n / 8 + 1), kBlock, 0, (hipStream_t)stream>>>(
      (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n, AxpbyF{alpha, b ? beta : 0.f});
  CG_CHECK_LAUNCH("cg_axpby");
  return CG_OK;
}

// out = a + b + c (+ d): the gradient contributions of a tensor with three / four consumers (the input
// of the self-attention block, arch_ops.py:709-758: three 1x1 projections and the residual path), summed
// in fp32 with ONE rounding -- autograd's accumulation was a chain of bf16 torch adds, each a pass of
// 2 reads + 1 write over the [N, H, W, C] map.
__global__ __launch_bounds__(256) void sum4_kernel(const bf16_t* __restrict__ a,
                                                   const bf16_t* __restrict__ b,
                                                   const bf16_t* __restrict__ c,
                                                   const bf16_t* __restrict__ d,
                                                   bf16_t* __restrict__ out, int64_t n) {
  const bool aligned = ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d) |
                         ((uintptr_t)out)) & 15) == 0;
  const int64_t nv = aligned ? n / 8 : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    V8 va, vb, vc, vd, vo;
    va.q = reinterpret_cast<const uint4*>(a)[i];
    vb.q = reinterpret_cast<const uint4*>(b)[i];
    vc.q = reinterpret_cast<const uint4*>(c)[i];
    if (d) vd.q = reinterpret_cast<const uint4*>(d)[i];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      vo.h[e] = f2bf(((bf2f(va.h[e]) + bf2f(vb.h[e])) + bf2f(vc.h[e])) + (d ? bf2f(vd.h[e]) : 0.f));
    reinterpret_cast<uint4*>(out)[i] = vo.q;
  }
  for (int64_t i = nv * 8 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = f2bf(((bf2f(a[i]) + bf2f(b[i])) + bf2f(c[i])) + (d ? bf2f(d[i]) : 0.f));
}
extern "C" int cg_sum4(const void* a, const void* b, const void* c, const void* d, void* out,
                       int64_t n, cgStream stream) {
  CG_NONNEG(n, "cg_sum4");
  if (n == 0) return CG_OK;
  if (!a || !b || !c || !out) CG_FAIL(CG_ERR_BAD_ARG, "cg_sum4: null pointer");
  sum4_kernel<<<grid_for(n / 8 + 1), 256, 0, (hipStream_t)stream>>>(
      (const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)c, (const bf16_t*)d, (bf16_t*)out, n);
  CG_CHECK_LAUNCH("cg_sum4");
  return CG_OK;
}

// ---- zero-insertion upsampling (resnet_ops.unpool) on its own --------------------------------------
// The convolutions take the upsampling as a launch parameter (U = 2, no zero is ever stored); the
// BigGAN-deep generator also upsamples its shortcut branch WITHOUT a convolution
// (resnet_biggan_deep.py:102-103).  out[n, 2i, 2j, :] = x[n, i, j, :] (+ residual), every other pixel
// = residual (or 0): one thread per 8 channels (C % 8 == 0) or per channel of one OUTPUT pixel.
template <bool VEC>
__global__ void unpool2_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res, int N,
                               int H, int W, int C, bf16_t* __restrict__ y) {
  const int Ho = H * 2, Wo = W * 2;
  const int CV = VEC ? C / 8 : C;
  const int64_t total = (int64_t)N * Ho * Wo * CV;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int cv = (int)(i % CV);
    int64_t p = i / CV;
    const int ow = (int)(p % Wo);
    p /= Wo;
    const int oh = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const bool hit = !(oh & 1) && !(ow & 1);
    const int64_t xo = (((int64_t)n * H + (oh >> 1)) * W + (ow >> 1)) * C;
    const int64_t yo = (((int64_t)n * Ho + oh) * Wo + ow) * C;
    if (VEC) {
      V8 a, r, o;
      a.q = make_uint4(0, 0, 0, 0);
      r.q = make_uint4(0, 0, 0, 0);
      if (hit) a.q = *reinterpret_cast<const uint4*>(x + xo + cv * 8);
      if (res) r.q = *reinterpret_cast<const uint4*>(res + yo + cv * 8);
      if (res && hit) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o.h[e] = f2bf(bf2f(a.h[e]) + bf2f(r.h[e]));
      } else {
        o.q = hit ? a.q : r.q;
      }
      *reinterpret_cast<uint4*>(y + yo + cv * 8) = o.q;
    } else {
      const float a = hit ? bf2f(x[xo + cv]) : 0.f;
      const float r = res ? bf2f(res[yo + cv]) : 0.f;
      y[yo + cv] = f2bf(a + r);
    }
  }
}
// gradient with respect to x: dx[n, i, j, :] = dy[n, 2i, 2j, :] (the residual's gradient is dy itself)
template <bool VEC>
__global__ void unpool2_bwd_kernel(const bf16_t* __restrict__ dy, int N, int H, int W, int C,
                                   bf16_t* __restrict__ dx) {
  const int Ho = H * 2, Wo = W * 2;
  const int CV = VEC ? C / 8 : C;
  const int64_t total = (int64_t)N * H * W * CV;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int cv = (int)(i % CV);
    int64_t p = i / CV;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const int n = (int)(p / H);
    const int64_t so = (((int64_t)n * Ho + 2 * h) * Wo + 2 * w) * C;
    const int64_t dst_o = (((int64_t)n * H + h) * W + w) * C;
    if (VEC)
      *reinterpret_cast<uint4*>(dx + dst_o + cv * 8) = *reinterpret_cast<const uint4*>(dy + so + cv * 8);
    else
      dx[dst_o + cv] = dy[so + cv];
  }
}

static int check_pool(const void* x, int N, int H, int W, int C, const void* y, const char* who) {
  if (!x || !y) CG_FAIL(CG_ERR_BAD_ARG, "%s: null pointer", who);
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (H & 1) || (W & 1))
    CG_FAIL(CG_ERR_BAD_ARG, "%s: bad shape [%d,%d,%d,%d] (H, W must be even)", who, N, H, W, C);
  return CG_OK;
}
template <bool MAXP>
static int pool2_launch(const void* x, int N, int H, int W, int C, void* y, hipStream_t st) {
  const bool vec = (C % 8) == 0;
  const int64_t work = (int64_t)N * (H / 2) * (W / 2) * (vec ? C / 8 : C);
  if (vec)
    pool2_kernel<true, MAXP><<<grid_for(work), kBlock, 0, st>>>((const bf16_t*)x, N, H, W, C,
                                                                 (bf16_t*)y);
  else
    pool2_kernel<false, MAXP><<<grid_for(work), kBlock, 0, st>>>((const bf16_t*)x, N, H, W, C,
                                                                  (bf16_t*)y);
  return CG_OK;
}
extern "C" int cg_avgpool2(const void* x, int N, int H, int W, int C, void* y, cgStream stream) {
  int rc = check_pool(x, N, H, W, C, y, "cg_avgpool2");
  if (rc) return rc;
  pool2_launch<false>(x, N, H, W, C, y, (hipStream_t)stream);
  CG_CHECK_LAUNCH("cg_avgpool2");
  return CG_OK;
}
extern "C" int cg_maxpool2(const void* x, int N, int H, int W, int C, void* y, cgStream stream) {
  int rc = check_pool(x, N, H, W, C, y, "cg_maxpool2");
  if (rc) return rc;
  pool2_launch<true>(x, N, H, W, C, y, (hipStream_t)stream);
  CG_CHECK_LAUNCH("cg_maxpool2");
  return CG_OK;
}
extern "C" int cg_avgpool2_bwd(const void* dy, int N, int H, int W, int C, void* dx,
                               cgStream stream) {
  int rc = check_pool(dy, N, H, W, C, dx, "cg_avgpool2_bwd");
  if (rc) return rc;
  const int64_t work = (int64_t)N * (H / 2) * (W / 2) * C;
  if ((C % 8) == 0 && ((((uintptr_t)dy) | ((uintptr_t)dx)) & 15) == 0)
    pool2_bwd_vec8_kernel<false><<<grid_for(work / 8), 256, 0, (hipStream_t)stream>>>(
        nullptr, (const bf16_t*)dy, N, H, W, C, (bf16_t*)dx);
  else
  pool2_bwd_kernel<false><<<grid_for(work), kBlock, 0, (hipStream_t)stream>>>(
      nullptr, (const bf16_t*)dy, N, H, W, C, (bf16_t*)dx);
  CG_CHECK_LAUNCH("cg_avgpool2_bwd");
  return CG_OK;
}
extern "C" int cg_unpool2(const void* x, const void* residual, int N, int H, int W, int C,
                          void* y, cgStream stream) {
  if (!x || !y) CG_FAIL(CG_ERR_BAD_ARG, "cg_unpool2: null pointer");
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_unpool2: bad shape [%d,%d,%d,%d]", N, H, W, C);
  const bool vec = (C % 8) == 0;
  const int64_t work = (int64_t)N * H * 2 * W * 2 * (vec ? C / 8 : C);
  if (vec)
    unpool2_kernel<true><<<grid_for(work), kBlock, 0, (hipStream_t)stream>>>(
        (const bf16_t*)x, (const bf16_t*)residual, N, H, W, C, (bf16_t*)y);
  else
    unpool2_kernel<false><<<grid_for(work), kBlock, 0, (hipStream_t)stream>>>(
        (const bf16_t*)x, (const bf16_t*)residual, N, H, W, C, (bf16_t*)y);
  CG_CHECK_LAUNCH("cg_unpool2");
  return CG_OK;
}
extern "C" int cg_unpool2_bwd(const void* dy, int N, int H, int W, int C, void* dx,
                              cgStream stream) {
  if (!dy || !dx) CG_FAIL(CG_ERR_BAD_ARG, "cg_unpool2_bwd: null pointer");
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_unpool2_bwd: bad shape [%d,%d,%d,%d]", N, H, W, C);
  const bool vec = (C % 8) == 0;
  const int64_t work = (int64_t)N * H * W * (vec ? C / 8 : C);
  if (vec)
    unpool2_bwd_kernel<true><<<grid_for(work), kBlock, 0, (hipStream_t)stream>>>(
        (const bf16_t*)dy, N, H, W, C, (bf16_t*)dx);
  else
    unpool2_bwd_kernel<false><<<grid_for(work), kBlock, 0, (hipStream_t)stream>>>(
        (const bf16_t*)dy, N, H, W, C, (bf16_t*)dx);
  CG_CHECK_LAUNCH("cg_unpool2_bwd");
  return CG_OK;
}
extern "C" int cg_maxpool2_bwd(const void* x, const void* dy, int N, int H, int W, int C,
                               void* dx, cgStream stream) {
  int rc = check_pool(x, N, H, W, C, dx, "cg_maxpool2_bwd");
  if (rc) return rc;
  if (!dy) CG_FAIL(CG_ERR_BAD_ARG, "cg_maxpool2_bwd: null dy");
  const int64_t work = (int64_t)N * (H / 2) * (W / 2) * C;
  if ((C % 8) == 0 && ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx)) & 15) == 0)
    pool2_bwd_vec8_kernel<true><<<grid_for(work / 8), 256, 0, (hipStream_t)stream>>>(
        (const bf16_t*)x, (const bf16_t*)dy, N, H, W, C, (bf16_t*)dx);
  else
  pool2_bwd_kernel<true><<<grid_for(work), kBlock, 0, (hipStream_t)stream>>>(
      (const bf16_t*)x, (const bf16_t*)dy, N, H, W, C, (bf16_t*)dx);
  CG_CHECK_LAUNCH("cg_maxpool2_bwd");
  return CG_OK;
}

extern "C" int cg_spatial_reduce(const void* x, const void* gate, int N, int HW, int C,
                                 float scale, void* out, cgStream stream) {
  if (!x || !out || N <= 0 || HW <= 0 || C <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_spatial_reduce: bad argument");
  dim3 grid(cdiv(C, 64), N);
  spatial_reduce_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(
      (const bf16_t*)x, (const bf16_t*)gate, HW, C, scale, (bf16_t*)out);
  CG_CHECK_LAUNCH("cg_spatial_reduce");
  return CG_OK;
}
extern "C" int cg_spatial_reduce_bwd(const void* gate, const void* dout, int N, int HW, int C,
                                     float scale, void* dx, cgStream stream) {
  if (!dout || !dx || N <= 0 || HW <= 0 || C <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_spatial_reduce_bwd: bad argument");
  const int64_t total = (int64_t)N * HW * C;
  spatial_reduce_bwd_kernel<<<grid_for(total), kBlock, 0, (hipStream_t)stream>>>(
      (const bf16_t*)gate, (const bf16_t*)dout, HW, C, scale, total, (bf16_t*)dx);
  CG_CHECK_LAUNCH("cg_spatial_reduce_bwd");
  return CG_OK;
}

extern "C" int cg_head(const float* x, int kind, float* y, int64_t n, cgStream stream) {
  CG_NONNEG(n, "cg_head");
  if (n == 0) return CG_OK;
  if (!x || !y || (kind != 0 && kind != 1)) CG_FAIL(CG_ERR_BAD_ARG, "cg_head: bad argument");
  head_kernel<<<grid_for(n), kBlock, 0, (hipStream_t)stream>>>(x, kind, y, n);
  CG_CHECK_LAUNCH("cg_head");
  return CG_OK;
}
extern "C" int cg_head_bwd(const float* y, int kind, const void* dy, int dy_is_f32, void* dx,
                           int64_t n, cgStream stream) {
  CG_NONNEG(n, "cg_head_bwd");
  if (n == 0) return CG_OK;
  if (!y || !dy || !dx || (kind != 0 && kind != 1))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_head_bwd: bad argument");
  head_bwd_kernel<<<grid_for(n), kBlock, 0, (hipStream_t)stream>>>(y, kind, dy, dy_is_f32,
                                                                  (bf16_t*)dx, n);
  CG_CHECK_LAUNCH("cg_head_bwd");
  return CG_OK;
}

extern "C" int cg_cast_f32_to_bf16(const float* x, void* y, int64_t n, cgStream stream) {
  CG_NONNEG(n, "cg_cast_f32_to_bf16");
  if (n == 0) return CG_OK;
  if (!x || !y) CG_FAIL(CG_ERR_BAD_ARG, "cg_cast_f32_to_bf16: null pointer");
  cast_f2b_kernel<<<grid_for(n), kBlock, 0, (hipStream_t)stream>>>(x, 1.f, 0.f, (bf16_t*)y, n);
  CG_CHECK_LAUNCH("cg_cast_f32_to_bf16");
  return CG_OK;
}
extern "C" int cg_affine_f32_to_bf16(const float* x, float a, float b, void* y, int64_t n,
                                     cgStream stream) {
  CG_NONNEG(n, "cg_affine_f32_to_bf16");
  if (n == 0) return CG_OK;
  if (!x || !y) CG_FAIL(CG_ERR_BAD_ARG, "cg_affine_f32_to_bf16: null pointer");
  cast_f2b_kernel<<<grid_for(n), kBlock, 0, (hipStream_t)stream>>>(x, a, b, (bf16_t*)y, n);
  CG_CHECK_LAUNCH("cg_affine_f32_to_bf16");
  return CG_OK;
}
extern "C" int cg_cast_bf16_to_f32(const void* x, float* y, int64_t n, cgStream stream) {
  CG_NONNEG(n, "cg_cast_bf16_to_f32");
  if (n == 0) return CG_OK;
  if (!x || !y) CG_FAIL(CG_ERR_BAD_ARG, "cg_cast_bf16_to_f32: null pointer");
  cast_b2f_kernel<<<grid_for(n), kBlock, 0, (hipStream_t)stream>>>((const bf16_t*)x, y, n);
  CG_CHECK_LAUNCH("cg_cast_bf16_to_f32");
  return CG_OK;
}

extern "C" size_t cg_colsum_workspace_bytes(int64_t rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  return align_up((size_t)colsum_splits(rows, C) * C * sizeof(float), 256);
}
extern "C" int cg_colsum(const void* x, int64_t rows, int C, float* out, void* ws,
                         size_t ws_bytes, cgStream stream) {
  if (!x || !out || rows <= 0 || C <= 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_colsum: bad argument");
  if (!ws || ws_bytes < cg_colsum_workspace_bytes(rows, C))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_colsum: workspace too small");
  const int splits = colsum_splits(rows, C);
  const int64_t rps = (rows + splits - 1) / splits;
  dim3 grid(cdiv(C, 64), splits);
  hipStream_t st = (hipStream_t)stream;
  colsum_part_kernel<<<grid, 256, 0, st>>>((const bf16_t*)x, rows, C, rps, (float*)ws);
  CG_CHECK_LAUNCH("cg_colsum(part)");
  colsum_final_kernel<<<cdiv(C, 64), 1024, 0, st>>>((const float*)ws, splits, C, out);
  CG_CHECK_LAUNCH("cg_colsum(final)");
  return CG_OK;
}

extern "C" int cg_rowdot(const void* a, const void* b, int B, int C, float* out,
                         cgStream stream) {
  if (!a || !b || !out || B <= 0 || C <= 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_rowdot: bad argument");
  rowdot_kernel<<<B, 256, 0, (hipStream_t)stream>>>((const bf16_t*)a, (const bf16_t*)b, C, out);
  CG_CHECK_LAUNCH("cg_rowdot");
  return CG_OK;
}
extern "C" int cg_rowdot_bwd(const void* a, const void* b, const float* dout, int B, int C,
                             void* da, void* db, cgStream stream) {
  if (!a || !b || !dout || B <= 0 || C <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_rowdot_bwd: bad argument");
  const int64_t total = (int64_t)B * C;
  rowdot_bwd_kernel<<<grid_for(total), kBlock, 0, (hipStream_t)stream>>>(
      (const bf16_t*)a, (const bf16_t*)b, dout, C, total, (bf16_t*)da, (bf16_t*)db);
  CG_CHECK_LAUNCH("cg_rowdot_bwd");
  return CG_OK;
}

extern "C" int cg_one_hot(const int32_t* labels, int B, int K, void* out, cgStream stream) {
  if (!labels || !out || B <= 0 || K <= 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_one_hot: bad argument");
  const int64_t total = (int64_t)B * K;
  one_hot_kernel<<<grid_for(total), kBlock, 0, (hipStream_t)stream>>>(labels, K, total,
                                                                      (bf16_t*)out);
  CG_CHECK_LAUNCH("cg_one_hot");
  return CG_OK;
}

extern "C" int cg_pool2d(const void* x, int N, int H, int W, int C, int k, int s, int p, int kind,
                         int Ho, int Wo, void* y, cgStream stream) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || s <= 0 || p < 0 || Ho <= 0 ||
      Wo <= 0 || (kind != 0 && kind != 1))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_pool2d: bad argument");
  const int64_t total = (int64_t)N * Ho * Wo * C;
  if (C % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
    const int64_t units = total / 8;
    int64_t blocks = (units + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    pool2d_vec8_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(
        (const bf16_t*)x, N, H, W, C / 8, k, s, p, kind, Ho, Wo, (bf16_t*)y, C, C, nullptr, 0);
    CG_CHECK_LAUNCH("cg_pool2d");
    return CG_OK;
  }
  pool2d_kernel<<<grid_for(total), kBlock, 0, (hipStream_t)stream>>>(
      (const bf16_t*)x, N, H, W, C, k, s, p, kind, Ho, Wo, (bf16_t*)y);
  CG_CHECK_LAUNCH("cg_pool2d");
  return CG_OK;
}

extern "C" int cg_pool2d_ld(const void* x, int x_ld, int N, int H, int W, int C, int k, int s, int p,
                            int kind, int Ho, int Wo, void* y, int y_ld, const float* bias, int relu,
                            cgStream stream) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || s <= 0 || p < 0 || Ho <= 0 ||
      Wo <= 0 || (kind != 0 && kind != 1))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_pool2d_ld: bad argument");
  if ((C & 7) || (x_ld & 7) || (y_ld & 7) || x_ld < C || y_ld < C || ((uintptr_t)x & 15) ||
      ((uintptr_t)y & 15) || (bias && ((uintptr_t)bias & 15)))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_pool2d_ld: channels and pitches must be multiples of 8 (>= C), "
                            "slices 16-byte aligned");
  const int64_t units = (int64_t)N * Ho * Wo * (C / 8);
  int64_t blocks = (units + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  pool2d_vec8_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(
      (const bf16_t*)x, N, H, W, C / 8, k, s, p, kind, Ho, Wo, (bf16_t*)y, x_ld, y_ld, bias, relu);
  CG_CHECK_LAUNCH("cg_pool2d_ld");
  return CG_OK;
}

extern "C" int cg_inception_preprocess(const float* x, int N, int H, int W, int C, int Ho, int Wo,
                                       void* y, cgStream stream) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_inception_preprocess: bad argument");
  const int64_t total = (int64_t)N * Ho * Wo * C;
  inception_preprocess_kernel<<<grid_for(total / 8 + 8), kBlock, 0, (hipStream_t)stream>>>(
      x, N, H, W, C, Ho, Wo, (bf16_t*)y);
  CG_CHECK_LAUNCH("cg_inception_preprocess");
  return CG_OK;
}

extern "C" int cg_axpby_f32(const float* a, float alpha, const float* b, float beta, float* out,
                            int64_t n, cgStream stream) {
  CG_NONNEG(n, "cg_axpby_f32");
  if (n == 0) return CG_OK;
  if (!a || !out) CG_FAIL(CG_ERR_BAD_ARG, "cg_axpby_f32: null pointer");
  axpby_f32_kernel<<<grid_for(n), kBlock, 0, (hipStream_t)stream>>>(a, alpha, b, beta, out, n);
  CG_CHECK_LAUNCH("cg_axpby_f32");
  return CG_OK;
}

extern "C" int cg_scale_count_nan_f32(const float* x, float scale, float* out, int64_t n,
                                      int32_t* nan_count, cgStream stream) {
  CG_NONNEG(n, "cg_scale_count_nan_f32");
  if (n == 0) return CG_OK;
  if (!x || !out || !nan_count) CG_FAIL(CG_ERR_BAD_ARG, "cg_scale_count_nan_f32: null pointer");
  int64_t b = (n / 4 + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  scale_count_nan_kernel<<<(int)b, 256, 0, (hipStream_t)stream>>>(x, scale, out, n, nan_count);
  CG_CHECK_LAUNCH("cg_scale_count_nan_f32");
  return CG_OK;
}

extern "C" int cg_axpy_dev(const void* x, const void* o, const float* sigma, void* out, int64_t n,
                           cgStream stream) {
  CG_NONNEG(n, "cg_axpy_dev");
  if (n == 0) return CG_OK;
  if (!o || !sigma || !out) CG_FAIL(CG_ERR_BAD_ARG, "cg_axpy_dev: null pointer");
  axpy_dev_kernel<<<grid_for(n), kBlock, 0, (hipStream_t)stream>>>(
      (const bf16_t*)x, (const bf16_t*)o, sigma, (bf16_t*)out, n);
  CG_CHECK_LAUNCH("cg_axpy_dev");
  return CG_OK;
}

extern "C" size_t cg_dot_bf16_workspace_bytes(int64_t n) {
  return n > 0 ? align_up((size_t)dot_blocks(n) * sizeof(float), 256) : 0;
}
extern "C" int cg_dot_bf16(const void* a, const void* b, int64_t n, float* out, void* ws,
                           size_t ws_bytes, cgStream stream) {
  if (!a || !b || !out || n <= 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_dot_bf16: bad argument");
  if (!ws || ws_bytes < cg_dot_bf16_workspace_bytes(n))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_dot_bf16: workspace too small");
  const int nb = dot_blocks(n);
  hipStream_t st = (hipStream_t)stream;
  dot_bf16_part_kernel<<<nb, 256, 0, st>>>((const bf16_t*)a, (const bf16_t*)b, n, (float*)ws);
  CG_CHECK_LAUNCH("cg_dot_bf16(part)");
  dot_final_kernel<<<1, 1, 0, st>>>((const float*)ws, nb, out);
  CG_CHECK_LAUNCH("cg_dot_bf16(final)");
  return CG_OK;
}
