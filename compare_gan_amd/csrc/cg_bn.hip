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

// Thread -> (channel vector, row lane) of the vector kernels.  A block covers the channel vectors
// [32 bx, 32 bx + LW), LW = min(32, CV - 32 bx), with RL = 256 / LW row lanes: every lane of the block
// works for any channel count (round 5 fixed LW = 32: C = 64 left 24 of every 32 lanes idle, C = 96 20
// -- bn_stats 1.1-1.5 TB/s, bn_apply 2.7 TB/s on the 128 x 128 layers, profiles/r06_stream_rates.txt).
struct BnLane {
  int lw, rl_n, cg, rl;
  bool on;
};
__device__ __forceinline__ BnLane bn_lane(int CV, int bx, int tid) {
  BnLane m;
  m.lw = min(32, CV - bx * 32);
  m.rl_n = 256 / m.lw;
  m.rl = tid / m.lw;
  m.cg = tid - m.rl * m.lw;
  m.on = m.rl < m.rl_n;
  return m;
}
// sum of the block's row lanes per (channel vector, element): sm[tid][0:16] -> 256 threads as (cg2 =
// tid >> 3, e2 = tid & 7), fixed order
__device__ __forceinline__ void bn_block_reduce(const float (*sm)[17], const BnLane& m, int tid,
                                                float& a, float& b, int& cg2, int& e2) {
  cg2 = tid >> 3;
  e2 = tid & 7;
  a = 0.f;
  b = 0.f;
  if (cg2 < m.lw) {
    for (int r = 0; r < m.rl_n; ++r) {
      a += sm[r * m.lw + cg2][e2];
      b += sm[r * m.lw + cg2][8 + e2];
    }
  }
}

// ---- statistics ------------------------------------------------------------------------------
// grid (ceil(CV/32), splits); thread (cg = tid%32, rl = tid/32) accumulates rows rl, rl+8, ...
// Statistics groups (cg_bn_stats_groups): the rows form `groups` consecutive blocks of group_rows
// rows with independent statistics; every group is cut into spg splits (groups = 1: spg = splits,
// group_rows = rows).
__global__ __launch_bounds__(256) void bn_stats_part_kernel(const bf16_t* __restrict__ x,
                                                            int64_t group_rows, int spg, int C,
                                                            int64_t rows_per_split,
                                                            float* __restrict__ part) {
  __shared__ float sm[256][17];
  const int CV = C / 8;
  const BnLane m = bn_lane(CV, blockIdx.x, threadIdx.x);
  const int cv = blockIdx.x * 32 + m.cg;
  const int grp = blockIdx.y / spg, sl = blockIdx.y - grp * spg;
  const int64_t r0 = grp * group_rows + (int64_t)sl * rows_per_split;
  const int64_t r1 = min((grp + 1) * group_rows, r0 + rows_per_split);
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
  if (m.on) {
    const bf16_t* xp = x + (int64_t)cv * 8;
    int64_t r = r0 + m.rl;
    // two rows in flight per lane
    for (; r + m.rl_n < r1; r += 2 * m.rl_n) {
      V8 v0, v1;
      v0.q = *reinterpret_cast<const uint4*>(xp + r * C);
      v1.q = *reinterpret_cast<const uint4*>(xp + (r + m.rl_n) * C);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = bf2f(v0.h[e]);
        s[e] += f;
        q[e] += f * f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = bf2f(v1.h[e]);
        s[e] += f;
        q[e] += f * f;
      }
    }
    for (; r < r1; r += m.rl_n) {
      V8 v;
      v.q = *reinterpret_cast<const uint4*>(xp + r * C);
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
    sm[threadIdx.x][e] = s[e];
    sm[threadIdx.x][8 + e] = q[e];
  }
  __syncthreads();
  float ss, qq;
  int cg2, e2;
  bn_block_reduce(sm, m, threadIdx.x, ss, qq, cg2, e2);
  if (cg2 < m.lw) {
    const int cv2 = blockIdx.x * 32 + cg2;
    float* p = part + (int64_t)blockIdx.y * 2 * C;
    p[cv2 * 8 + e2] = ss;
    p[C + cv2 * 8 + e2] = qq;
  }
}
// scalar fallback (C % 8 != 0): grid (ceil(C/64), splits), 4 waves split rows.
__global__ __launch_bounds__(256) void bn_stats_part_scalar_kernel(const bf16_t* __restrict__ x,
                                                                   int64_t group_rows, int spg,
                                                                   int C, int64_t rows_per_split,
                                                                   float* __restrict__ part) {
  __shared__ float sm[2][4][64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + l;
  const int grp = blockIdx.y / spg, sl = blockIdx.y - grp * spg;
  const int64_t r0 = grp * group_rows + (int64_t)sl * rows_per_split;
  const int64_t r1 = min((grp + 1) * group_rows, r0 + rows_per_split);
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
// one block per BN_CL channels: BN_ZL split-lanes per channel sum the partials (4 loads in flight
// each: a short serial walk, the finalize is pure load latency), LDS combines them.  16 channels x 64
// lanes: a 256-channel layer runs on 16 workgroups with 4-8 trips per lane (32 x 32 left it on 8).
// Optionally folds the moving-average update m <- m - (1-decay)(m - batch) (arch_ops.py:105-114).
constexpr int BN_CL = 16, BN_ZL = 64;
__global__ __launch_bounds__(BN_CL * BN_ZL) void bn_stats_final_kernel(const float* __restrict__ part,
                                                             int splits, int C, float inv_rows,
                                                             float* __restrict__ mean,
                                                             float* __restrict__ var,
                                                             float* __restrict__ mm,
                                                             float* __restrict__ mv, float decay) {
  __shared__ float sm[2][BN_ZL][BN_CL + 1];
  const int cl = threadIdx.x & (BN_CL - 1), zl = threadIdx.x / BN_CL;
  const int c = blockIdx.x * BN_CL + cl;
  float s = 0.f, q = 0.f;
  if (c < C) {
#pragma unroll 4
    for (int z = zl; z < splits; z += BN_ZL) {
      s += part[(int64_t)z * 2 * C + c];
      q += part[(int64_t)z * 2 * C + C + c];
    }
  }
  sm[0][zl][cl] = s;
  sm[1][zl][cl] = q;
  __syncthreads();
  if (zl == 0 && c < C) {
    float ss = 0.f, qq = 0.f;
#pragma unroll
    for (int r = 0; r < BN_ZL; ++r) {
      ss += sm[0][r][cl];
      qq += sm[1][r][cl];
    }
    const float m = ss * inv_rows;
    const float v = qq * inv_rows - m * m;  // tf.nn.normalize_moments(shift=None)
    mean[c] = m;
    var[c] = v;
    if (mm) {
      mm[c] -= (1.f - decay) * (mm[c] - m);
      mv[c] -= (1.f - decay) * (mv[c] - v);
    }
  }
}
// The same finalisation for `groups` independent statistics groups: partial row (ph, t) = ph * T + t
// with T = rows / phases rows per phase (the U*U output phases of an up-sampling producer conv,
// cg_gconv_fused stats_out) and group g owning t in [g * T / groups, (g + 1) * T / groups).
// mean / var are [groups][C].  The moving averages take the groups' updates in order, as the
// separate network calls they stand for would have applied them (arch_ops.py:105-114).
// Up to BN_GMAX groups are summed at once -- all their loads in flight, ONE barrier pair -- instead of
// a load / barrier / combine round per group (5 groups: 13 us of serial latency per call).  Lanes
// and summation order are those of bn_stats_final_kernel: a group's statistics are bit for bit the
// ones of the separate call it stands for.
constexpr int BN_GMAX = 6;
__global__ __launch_bounds__(BN_CL * BN_ZL) void bn_stats_final_groups_kernel(
    const float* __restrict__ part, int rows, int C, int groups, int phases, float inv_count,
    float* __restrict__ mean, float* __restrict__ var, float* __restrict__ mm,
    float* __restrict__ mv, float decay) {
  __shared__ float sm[2][BN_GMAX][BN_ZL][BN_CL + 1];
  __shared__ float mo[2][BN_GMAX][BN_CL];
  const int cl = threadIdx.x & (BN_CL - 1), zl = threadIdx.x / BN_CL;
  const int c = blockIdx.x * BN_CL + cl;
  const int T = rows / phases, tg = T / groups, per_group = tg * phases;
  float mmc = 0.f, mvc = 0.f;
  if (mm && zl == 0 && c < C) { mmc = mm[c]; mvc = mv[c]; }
  for (int g0 = 0; g0 < groups; g0 += BN_GMAX) {
    const int ng = min(BN_GMAX, groups - g0);
    float s[BN_GMAX], q[BN_GMAX];
#pragma unroll
    for (int j = 0; j < BN_GMAX; ++j) s[j] = q[j] = 0.f;
    if (c < C) {
      for (int z = zl; z < per_group; z += BN_ZL) {
        const int ph = z / tg, t = z - ph * tg;
#pragma unroll
        for (int j = 0; j < BN_GMAX; ++j)
          if (j < ng) {
            const int64_t row = (int64_t)ph * T + (g0 + j) * tg + t;
            s[j] += part[row * 2 * C + c];
            q[j] += part[row * 2 * C + C + c];
          }
      }
    }
    __syncthreads();   // (the previous chunk's readers are done)
#pragma unroll
    for (int j = 0; j < BN_GMAX; ++j) {
      sm[0][j][zl][cl] = s[j];
      sm[1][j][zl][cl] = q[j];
    }
    __syncthreads();
    if (zl < ng && c < C) {   // split-lane j finishes group g0 + j
      float ss = 0.f, qq = 0.f;
#pragma unroll
      for (int r = 0; r < BN_ZL; ++r) {
        ss += sm[0][zl][r][cl];
        qq += sm[1][zl][r][cl];
      }
      const float m = ss * inv_count;
      const float v = qq * inv_count - m * m;
      mean[(int64_t)(g0 + zl) * C + c] = m;
      var[(int64_t)(g0 + zl) * C + c] = v;
      mo[0][zl][cl] = m;
      mo[1][zl][cl] = v;
    }
    if (mm) {   // (kernel-uniform) the moving averages take the groups' updates in order
      __syncthreads();
      if (zl == 0 && c < C)
        for (int j = 0; j < ng; ++j) {
          mmc -= (1.f - decay) * (mmc - mo[0][j][cl]);
          mvc -= (1.f - decay) * (mvc - mo[1][j][cl]);
        }
    }
  }
  if (mm && zl == 0 && c < C) { mm[c] = mmc; mv[c] = mvc; }
}
inline int stats_splits(int64_t rows, int C) {
  const int ct = (C % 8 == 0) ? cdiv(C / 8, 32) : cdiv(C, 64);
  int s = cdiv(1024, ct);
  const int64_t maxs = rows / 32 > 0 ? rows / 32 : 1;
  if (s > maxs) s = (int)maxs;
  if (s < 1) s = 1;
  if (s > 512) s = 512;
  return s;
}

// ---- apply -----------------------------------------------------------------------------------
// grid (ceil(CV/32), hw chunks, N); thread (cvl = tid & 31, rl = tid >> 5) owns 8 channels and
// walks the rows rl, rl+8, ... of its chunk: the affine (scale, shift) is computed once.
__global__ __launch_bounds__(256) void bn_apply_vec_kernel(
    const bf16_t* __restrict__ x, int HW, int C, int hw_per_chunk, const float* __restrict__ mean,
    const float* __restrict__ var, float eps, const float* __restrict__ gamma,
    const float* __restrict__ beta, int per_sample, int stat_group, int relu,
    bf16_t* __restrict__ y) {
  const BnLane m = bn_lane(C / 8, blockIdx.x, threadIdx.x);
  if (!m.on) return;
  const int cv = blockIdx.x * 32 + m.cg;
  const int rl = m.rl, RL = m.rl_n;
  const int n = blockIdx.z;
  const int c0 = cv * 8;
  // stat_group > 0: mean / var are [N / stat_group][C], one set per stat_group consecutive samples
  const int64_t sb = stat_group > 0 ? (int64_t)(n / stat_group) * C : 0;
  // the reference's operation order (x - mean) * rstd [* gamma] [+ beta] is kept: folding it into
  // one multiply-add cancels catastrophically where x ~ mean (tiny batches, arch_ops.py:306-312)
  float mu[8], rs[8], gm[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu[e] = mean[sb + c0 + e];
    rs[e] = rsqrtf(var[sb + c0 + e] + eps);
    const int64_t pidx = per_sample ? (int64_t)n * C + c0 + e : c0 + e;
    gm[e] = gamma ? gamma[pidx] : 1.f;
    bt[e] = beta ? beta[pidx] : 0.f;
  }
  const int h0 = blockIdx.y * hw_per_chunk, h1 = min(HW, h0 + hw_per_chunk);
  const bf16_t* xp = x + (int64_t)n * HW * C + c0;
  bf16_t* yp = y + (int64_t)n * HW * C + c0;
  int p = h0 + rl;
  for (; p + RL < h1; p += 2 * RL) {   // two rows in flight per lane
    V8 v0, v1, o0, o1;
    v0.q = *reinterpret_cast<const uint4*>(xp + (int64_t)p * C);
    v1.q = *reinterpret_cast<const uint4*>(xp + (int64_t)(p + RL) * C);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = (bf2f(v0.h[e]) - mu[e]) * rs[e];
      t = t * gm[e] + bt[e];
      if (relu) t = fmaxf(t, 0.f);
      o0.h[e] = f2bf(t);
      float u = (bf2f(v1.h[e]) - mu[e]) * rs[e];
      u = u * gm[e] + bt[e];
      if (relu) u = fmaxf(u, 0.f);
      o1.h[e] = f2bf(u);
    }
    *reinterpret_cast<uint4*>(yp + (int64_t)p * C) = o0.q;
    *reinterpret_cast<uint4*>(yp + (int64_t)(p + RL) * C) = o1.q;
  }
  for (; p < h1; p += RL) {
    V8 v, o;
    v.q = *reinterpret_cast<const uint4*>(xp + (int64_t)p * C);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = (bf2f(v.h[e]) - mu[e]) * rs[e];
      t = t * gm[e] + bt[e];
      if (relu) t = fmaxf(t, 0.f);
      o.h[e] = f2bf(t);
    }
    *reinterpret_cast<uint4*>(yp + (int64_t)p * C) = o.q;
  }
}
// scalar fallback (C % 8 != 0)
__global__ void bn_apply_scalar_kernel(const bf16_t* __restrict__ x, int HW, int C,
                                       int64_t total, const float* __restrict__ mean,
                                       const float* __restrict__ var, float eps,
                                       const float* __restrict__ gamma,
                                       const float* __restrict__ beta, int per_sample,
                                       int stat_group, int relu, bf16_t* __restrict__ y) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C);
    const int64_t n = i / ((int64_t)HW * C);
    const int64_t pidx = per_sample ? n * C + c : c;
    const int64_t sidx = stat_group > 0 ? (n / stat_group) * C + c : c;
    const float rstd = rsqrtf(var[sidx] + eps);
    float t = (bf2f(x[i]) - mean[sidx]) * rstd;
    if (gamma) t *= gamma[pidx];
    if (beta) t += beta[pidx];
    if (relu) t = fmaxf(t, 0.f);
    y[i] = f2bf(t);
  }
}

// ---- backward --------------------------------------------------------------------------------
// pass 1: S1[n,c] = sum_hw dz, S2[n,c] = sum_hw dz*xhat, dz = dy * (y>0 if relu).
// vector form: grid (ceil(CV/32), hsplits, N); partial [hs][2][N][C].
__global__ __launch_bounds__(256) void bn_bwd_sums_vec_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ y, const bf16_t* __restrict__ dy,
    int HW, int C, int hw_per_split, const float* __restrict__ mean,
    const float* __restrict__ var, float eps, int relu, float* __restrict__ part) {
  __shared__ float sm[256][17];
  const BnLane m = bn_lane(C / 8, blockIdx.x, threadIdx.x);
  const int cv = blockIdx.x * 32 + m.cg;
  const int rl = m.rl, RL = m.rl_n;
  const int n = blockIdx.z, N = gridDim.z;
  const int h0 = blockIdx.y * hw_per_split, h1 = min(HW, h0 + hw_per_split);
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  if (m.on) {
    const int c0 = cv * 8;
    float mu[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mu[e] = mean[c0 + e];
      rs[e] = rsqrtf(var[c0 + e] + eps);
    }
    const int64_t base = (int64_t)n * HW * C + c0;
    for (int p = h0 + rl; p < h1; p += RL) {
      const int64_t o = base + (int64_t)p * C;
      V8 vx, vy, vg;
      vx.q = *reinterpret_cast<const uint4*>(x + o);
      vg.q = *reinterpret_cast<const uint4*>(dy + o);
      if (relu) vy.q = *reinterpret_cast<const uint4*>(y + o);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float g = bf2f(vg.h[e]);
        if (relu && !(bf2f(vy.h[e]) > 0.f)) g = 0.f;
        s1[e] += g;
        s2[e] += g * (bf2f(vx.h[e]) - mu[e]) * rs[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sm[threadIdx.x][e] = s1[e];
    sm[threadIdx.x][8 + e] = s2[e];
  }
  __syncthreads();
  float a, b;
  int cg2, e2;
  bn_block_reduce(sm, m, threadIdx.x, a, b, cg2, e2);
  const int cv2 = blockIdx.x * 32 + cg2;
  if (cg2 < m.lw) {
    float* p = part + (int64_t)blockIdx.y * 2 * N * C;
    p[(int64_t)n * C + cv2 * 8 + e2] = a;
    p[(int64_t)N * C + (int64_t)n * C + cv2 * 8 + e2] = b;
  }
}
// scalar form (C % 8 != 0): grid (ceil(C/64), N, hsplits): 4 waves split hw.
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
// finalize: block = 32 channels x BN_BZL lanes.  per-sample: lanes split the samples (each sample's
// dgamma/dbeta is complete after the hw-split sum); otherwise lanes split (n, z) jointly.
constexpr int BN_BZL = 32;   // split-lanes of the backward finaliser (32 channels x 32 lanes)
__global__ __launch_bounds__(32 * BN_BZL) void bn_bwd_final_kernel(
    const float* __restrict__ part, int hsplits, int N, int C, float inv_rows,
    const float* __restrict__ gamma, int per_sample, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ m12) {
  __shared__ float sm[2][BN_BZL][33];
  const int cl = threadIdx.x & 31, zl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float a1 = 0.f, a2 = 0.f;
  if (c < C) {
    for (int n = zl; n < N; n += BN_BZL) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
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
        a1 += s1;
        a2 += s2;
      }
    }
  }
  sm[0][zl][cl] = a1;
  sm[1][zl][cl] = a2;
  __syncthreads();
  if (zl == 0 && c < C) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int r = 0; r < BN_BZL; ++r) {
      t1 += sm[0][r][cl];
      t2 += sm[1][r][cl];
    }
    if (!per_sample) {
      const float g = gamma ? gamma[c] : 1.f;
      if (dgamma) dgamma[c] = t2;
      if (dbeta) dbeta[c] = t1;
      t1 *= g;
      t2 *= g;
    }
    m12[c] = t1 * inv_rows;
    m12[C + c] = t2 * inv_rows;
  }
}
// pass 2, vector form: grid (ceil(CV/32), hw chunks, N)
__global__ __launch_bounds__(256) void bn_bwd_dx_vec_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ y, const bf16_t* __restrict__ dy,
    int HW, int C, int hw_per_chunk, const float* __restrict__ mean,
    const float* __restrict__ var, float eps, const float* __restrict__ gamma, int per_sample,
    int relu, int batch_stats, const float* __restrict__ m12, bf16_t* __restrict__ dx) {
  const BnLane m = bn_lane(C / 8, blockIdx.x, threadIdx.x);
  if (!m.on) return;
  const int cv = blockIdx.x * 32 + m.cg;
  const int rl = m.rl, RL = m.rl_n;
  const int n = blockIdx.z;
  const int c0 = cv * 8;
  float mu[8], rs[8], gm[8], k1[8], k2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu[e] = mean[c0 + e];
    rs[e] = rsqrtf(var[c0 + e] + eps);
    gm[e] = gamma ? gamma[per_sample ? (int64_t)n * C + c0 + e : c0 + e] : 1.f;
    k1[e] = batch_stats ? m12[c0 + e] : 0.f;
    k2[e] = batch_stats ? m12[C + c0 + e] : 0.f;
  }
  const int h0 = blockIdx.y * hw_per_chunk, h1 = min(HW, h0 + hw_per_chunk);
  const int64_t base = (int64_t)n * HW * C + c0;
  for (int p = h0 + rl; p < h1; p += RL) {
    const int64_t o = base + (int64_t)p * C;
    V8 vx, vy, vg, vo;
    vg.q = *reinterpret_cast<const uint4*>(dy + o);
    if (batch_stats) vx.q = *reinterpret_cast<const uint4*>(x + o);
    if (relu) vy.q = *reinterpret_cast<const uint4*>(y + o);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float g = bf2f(vg.h[e]);
      if (relu && !(bf2f(vy.h[e]) > 0.f)) g = 0.f;
      float d = gm[e] * g;
      if (batch_stats) {
        const float xhat = (bf2f(vx.h[e]) - mu[e]) * rs[e];
        d = d - k1[e] - xhat * k2[e];
      }
      vo.h[e] = f2bf(d * rs[e]);
    }
    *reinterpret_cast<uint4*>(dx + o) = vo.q;
  }
}
// pass 2, scalar form
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
  const int blocks = ((C % 8 == 0) ? cdiv(C / 8, 32) : cdiv(C, 64)) * N;
  int s = cdiv(1024, blocks);
  const int maxs = HW / 16 > 0 ? HW / 16 : 1;
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  return s;
}
// hw chunks so that (channel blocks) x chunks x N ~ 2048 blocks, >= 32 rows per chunk
inline int hw_chunks(int N, int HW, int C) {
  const int64_t blocks = (int64_t)cdiv(C / 8, 32) * N;
  int64_t ch = (2048 + blocks - 1) / blocks;
  const int maxc = HW / 32 > 0 ? HW / 32 : 1;
  if (ch > maxc) ch = maxc;
  if (ch < 1) ch = 1;
  return (int)ch;
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

extern "C" int cg_bn_stats(const void* x, int64_t rows, int C, float* mean, float* var,
                           float* moving_mean, float* moving_var, float decay, void* ws,
                           size_t ws_bytes, cgStream stream) {
  if (!x || !mean || !var || rows <= 0 || C <= 0 || ((moving_mean == nullptr) != (moving_var == nullptr)))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_bn_stats: bad argument");
  if (!ws || ws_bytes < cg_bn_stats_workspace_bytes(rows, C))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_bn_stats: workspace too small");
  const int splits = stats_splits(rows, C);
  const int64_t rps = (rows + splits - 1) / splits;
  hipStream_t st = (hipStream_t)stream;
  if (C % 8 == 0) {
    dim3 grid(cdiv(C / 8, 32), splits);
    bn_stats_part_kernel<<<grid, 256, 0, st>>>((const bf16_t*)x, rows, splits, C, rps, (float*)ws);
  } else {
    dim3 grid(cdiv(C, 64), splits);
    bn_stats_part_scalar_kernel<<<grid, 256, 0, st>>>((const bf16_t*)x, rows, splits, C, rps,
                                                      (float*)ws);
  }
  CG_CHECK_LAUNCH("cg_bn_stats(part)");
  bn_stats_final_kernel<<<cdiv(C, BN_CL), BN_CL * BN_ZL, 0, st>>>((const float*)ws, splits, C,
                                                     1.0f / (float)rows, mean, var, moving_mean,
                                                     moving_var, decay);
  CG_CHECK_LAUNCH("cg_bn_stats(final)");
  return CG_OK;
}

// `groups` consecutive blocks of rows / groups rows with independent statistics: the batch norm of
// several network calls run as one batched call (modular_gan.py:464-467: the generator forwards of
// all sub-steps share the generator's weights).  mean / var [groups][C].
extern "C" size_t cg_bn_stats_groups_workspace_bytes(int64_t rows, int C, int groups) {
  if (rows <= 0 || C <= 0 || groups <= 0 || rows % groups) return 0;
  return align_up((size_t)groups * stats_splits(rows / groups, C) * 2 * C * sizeof(float), 256);
}

extern "C" int cg_bn_stats_groups(const void* x, int64_t rows, int C, int groups, float* mean,
                                  float* var, float* moving_mean, float* moving_var, float decay,
                                  void* ws, size_t ws_bytes, cgStream stream) {
  if (!x || !mean || !var || rows <= 0 || C <= 0 || groups <= 0 || rows % groups ||
      ((moving_mean == nullptr) != (moving_var == nullptr)))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_bn_stats_groups: bad argument");
  if (!ws || ws_bytes < cg_bn_stats_groups_workspace_bytes(rows, C, groups))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_bn_stats_groups: workspace too small");
  const int64_t grows = rows / groups;
  const int spg = stats_splits(grows, C);
  const int64_t rps = (grows + spg - 1) / spg;
  hipStream_t st = (hipStream_t)stream;
  if (C % 8 == 0) {
    dim3 grid(cdiv(C / 8, 32), groups * spg);
    bn_stats_part_kernel<<<grid, 256, 0, st>>>((const bf16_t*)x, grows, spg, C, rps, (float*)ws);
  } else {
    dim3 grid(cdiv(C, 64), groups * spg);
    bn_stats_part_scalar_kernel<<<grid, 256, 0, st>>>((const bf16_t*)x, grows, spg, C, rps,
                                                      (float*)ws);
  }
  CG_CHECK_LAUNCH("cg_bn_stats_groups(part)");
  bn_stats_final_groups_kernel<<<cdiv(C, BN_CL), BN_CL * BN_ZL, 0, st>>>(
      (const float*)ws, groups * spg, C, groups, 1, 1.0f / (float)grows, mean, var, moving_mean,
      moving_var, decay);
  CG_CHECK_LAUNCH("cg_bn_stats_groups(final)");
  return CG_OK;
}

extern "C" int cg_bn_finalize_groups(const float* partials, int rows, int C, int64_t count,
                                     int groups, int phases, float* mean, float* var,
                                     float* moving_mean, float* moving_var, float decay,
                                     cgStream stream) {
  if (!partials || !mean || !var || rows <= 0 || C <= 0 || count <= 0 || groups <= 0 ||
      phases <= 0 || rows % phases || (rows / phases) % groups ||
      ((moving_mean == nullptr) != (moving_var == nullptr)))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_bn_finalize_groups: bad argument");
  hipStream_t st = (hipStream_t)stream;
  bn_stats_final_groups_kernel<<<cdiv(C, BN_CL), BN_CL * BN_ZL, 0, st>>>(
      partials, rows, C, groups, phases, 1.0f / (float)count, mean, var, moving_mean, moving_var,
      decay);
  CG_CHECK_LAUNCH("cg_bn_finalize_groups");
  return CG_OK;
}

// accumulator statistics (arch_ops.py:122-191): accu += batch moments, counter += 1; and their read
// side accu / counter -- all on the device, no host round trip per batch norm call
__global__ void bn_accu_add_kernel(float* __restrict__ am, float* __restrict__ av,
                                   float* __restrict__ cnt, const float* __restrict__ mean,
                                   const float* __restrict__ var, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    am[c] += mean[c];
    av[c] += var[c];
  }
  if (c == 0) *cnt += 1.f;
}
__global__ void bn_accu_read_kernel(const float* __restrict__ am, const float* __restrict__ av,
                                    const float* __restrict__ cnt, float* __restrict__ mean,
                                    float* __restrict__ var, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float n = *cnt;
    mean[c] = am[c] / n;
    var[c] = av[c] / n;
  }
}

extern "C" int cg_bn_accumulate(float* accu_mean, float* accu_var, float* accu_counter,
                                const float* mean, const float* var, int C, cgStream stream) {
  if (!accu_mean || !accu_var || !accu_counter || !mean || !var || C <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_bn_accumulate: bad argument");
  bn_accu_add_kernel<<<cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(accu_mean, accu_var,
                                                                    accu_counter, mean, var, C);
  CG_CHECK_LAUNCH("cg_bn_accumulate");
  return CG_OK;
}

extern "C" int cg_bn_accumulated_moments(const float* accu_mean, const float* accu_var,
                                         const float* accu_counter, float* mean, float* var,
                                         int C, cgStream stream) {
  if (!accu_mean || !accu_var || !accu_counter || !mean || !var || C <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_bn_accumulated_moments: bad argument");
  bn_accu_read_kernel<<<cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(accu_mean, accu_var,
                                                                     accu_counter, mean, var, C);
  CG_CHECK_LAUNCH("cg_bn_accumulated_moments");
  return CG_OK;
}

extern "C" int cg_bn_finalize(const float* partials, int rows, int C, int64_t count, float* mean,
                              float* var, float* moving_mean, float* moving_var, float decay,
                              cgStream stream) {
  if (!partials || !mean || !var || rows <= 0 || C <= 0 || count <= 0 ||
      ((moving_mean == nullptr) != (moving_var == nullptr)))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_bn_finalize: bad argument");
  hipStream_t st = (hipStream_t)stream;
  bn_stats_final_kernel<<<cdiv(C, BN_CL), BN_CL * BN_ZL, 0, st>>>(partials, rows, C, 1.0f / (float)count,
                                                             mean, var, moving_mean, moving_var,
                                                             decay);
  CG_CHECK_LAUNCH("cg_bn_finalize");
  return CG_OK;
}

extern "C" int cg_bn_apply_groups(const void* x, int N, int HW, int C, const float* mean,
                                  const float* var, float eps, const float* gamma,
                                  const float* beta, int per_sample, int stat_group, int relu,
                                  void* y, cgStream stream);

extern "C" int cg_bn_apply(const void* x, int N, int HW, int C, const float* mean,
                           const float* var, float eps, const float* gamma, const float* beta,
                           int per_sample, int relu, void* y, cgStream stream) {
  return cg_bn_apply_groups(x, N, HW, C, mean, var, eps, gamma, beta, per_sample, 0, relu, y,
                            stream);
}

// stat_group > 0: mean / var are [N / stat_group][C] (cg_bn_stats_groups); 0: [C]
extern "C" int cg_bn_apply_groups(const void* x, int N, int HW, int C, const float* mean,
                                  const float* var, float eps, const float* gamma,
                                  const float* beta, int per_sample, int stat_group, int relu,
                                  void* y, cgStream stream) {
  if (!x || !y || !mean || !var || N <= 0 || HW <= 0 || C <= 0 || stat_group < 0 ||
      (stat_group > 0 && N % stat_group))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_bn_apply: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (C % 8 == 0) {
    const int ch = hw_chunks(N, HW, C);
    dim3 grid(cdiv(C / 8, 32), ch, N);
    bn_apply_vec_kernel<<<grid, 256, 0, st>>>((const bf16_t*)x, HW, C, cdiv(HW, ch), mean, var,
                                              eps, gamma, beta, per_sample, stat_group, relu,
                                              (bf16_t*)y);
  } else {
    const int64_t units = (int64_t)N * HW * C;
    bn_apply_scalar_kernel<<<grid_cap(units), 256, 0, st>>>((const bf16_t*)x, HW, C, units, mean,
                                                            var, eps, gamma, beta, per_sample,
                                                            stat_group, relu, (bf16_t*)y);
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
  if (C % 8 == 0) {
    dim3 grid(cdiv(C / 8, 32), hs, N);
    bn_bwd_sums_vec_kernel<<<grid, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)y,
                                                 (const bf16_t*)dy, HW, C, hps, mean, var, eps,
                                                 relu, part);
  } else {
    dim3 grid(cdiv(C, 64), N, hs);
    bn_bwd_sums_kernel<<<grid, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)y,
                                             (const bf16_t*)dy, HW, C, hps, mean, var, eps, relu,
                                             part);
  }
  CG_CHECK_LAUNCH("cg_bn_backward_reduce(sums)");
  bn_bwd_final_kernel<<<cdiv(C, 32), 32 * BN_BZL, 0, st>>>(part, hs, N, C,
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
  hipStream_t st = (hipStream_t)stream;
  if (C % 8 == 0) {
    const int ch = hw_chunks(N, HW, C);
    dim3 grid(cdiv(C / 8, 32), ch, N);
    bn_bwd_dx_vec_kernel<<<grid, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)y,
                                               (const bf16_t*)dy, HW, C, cdiv(HW, ch), mean, var,
                                               eps, gamma, per_sample, relu, batch_stats, m12,
                                               (bf16_t*)dx);
  } else {
    const int64_t total = (int64_t)N * HW * C;
    bn_bwd_dx_kernel<<<grid_cap(total), 256, 0, st>>>(
        (const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, HW, C, total, mean, var, eps,
        gamma, per_sample, relu, batch_stats, m12, (bf16_t*)dx);
  }
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
