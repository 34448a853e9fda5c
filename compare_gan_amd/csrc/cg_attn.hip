// Self-attention core of arch_ops.non_local_block (arch_ops.py:744-753):
//   attn = softmax(theta phi^T), out = attn g   per image, never materialising [B, Lq, Lk].
// v1: streaming online-softmax on the vector ALU (Dk = C/8 is 12..48, too thin for MFMA K);
// four lanes cooperate on one query (forward, dtheta) or one key (dphi, dg), each owning a
// quarter of the value / key channels; K/V (resp. Q/dO) tiles of 64 rows are staged in LDS.
#include <stdlib.h>

#include "cg_common.h"

namespace {

constexpr int TILE = 64;    // rows staged per LDS tile
constexpr int DKMAX = 64;   // max key channels
constexpr int DVQMAX = 32;  // max value channels per lane (Dv <= 128)
constexpr int DKQMAX = 16;  // max key channels per lane for gradients (Dk <= 64)

__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}

// grid (Lq/64, B); block 256: thread t -> query t/4, quarter t%4.
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ theta,
                                                       const bf16_t* __restrict__ phi,
                                                       const bf16_t* __restrict__ g, int Lq,
                                                       int Lk, int Dk, int Dv,
                                                       bf16_t* __restrict__ out,
                                                       float* __restrict__ lse) {
  __shared__ float sphi[TILE][DKMAX + 1];
  __shared__ float sg[TILE][4 * DVQMAX + 1];
  const int b = blockIdx.y;
  const int q = blockIdx.x * 64 + (threadIdx.x >> 2);
  const int part = threadIdx.x & 3;
  const int dvq = Dv / 4;
  const bool qok = q < Lq;
  float th[DKMAX];
#pragma unroll
  for (int d = 0; d < DKMAX; ++d)
    th[d] = (qok && d < Dk) ? bf2f(theta[((int64_t)b * Lq + q) * Dk + d]) : 0.f;
  float acc[DVQMAX];
#pragma unroll
  for (int j = 0; j < DVQMAX; ++j) acc[j] = 0.f;
  float m = -3.0e38f, l = 0.f;
  for (int k0 = 0; k0 < Lk; k0 += TILE) {
    __syncthreads();
    for (int i = threadIdx.x; i < TILE * Dk; i += 256) {
      const int r = i / Dk, d = i - r * Dk;
      sphi[r][d] = (k0 + r < Lk) ? bf2f(phi[((int64_t)b * Lk + k0 + r) * Dk + d]) : 0.f;
    }
    for (int i = threadIdx.x; i < TILE * Dv; i += 256) {
      const int r = i / Dv, d = i - r * Dv;
      sg[r][d] = (k0 + r < Lk) ? bf2f(g[((int64_t)b * Lk + k0 + r) * Dv + d]) : 0.f;
    }
    __syncthreads();
    const int kn = min(TILE, Lk - k0);
    for (int r = 0; r < kn; ++r) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < DKMAX; ++d)
        if (d < Dk) s += th[d] * sphi[r][d];
      if (s > m) {
        const float corr = __expf(m - s);
        l *= corr;
#pragma unroll
        for (int j = 0; j < DVQMAX; ++j) acc[j] *= corr;
        m = s;
      }
      const float p = __expf(s - m);
      l += p;
#pragma unroll
      for (int j = 0; j < DVQMAX; ++j)
        if (j < dvq) acc[j] += p * sg[r][part * dvq + j];
    }
  }
  if (qok) {
    const float inv = 1.f / l;
#pragma unroll
    for (int j = 0; j < DVQMAX; ++j)
      if (j < dvq) out[((int64_t)b * Lq + q) * Dv + part * dvq + j] = f2bf(acc[j] * inv);
    if (part == 0) lse[(int64_t)b * Lq + q] = m + __logf(l);
  }
}

// delta[b,q] = sum_j dout * out
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ out,
                                                         const bf16_t* __restrict__ dout,
                                                         int64_t rows, int Dv,
                                                         float* __restrict__ delta) {
  const int64_t r = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 2);
  const int part = threadIdx.x & 3;
  float s = 0.f;
  if (r < rows)
    for (int j = part; j < Dv; j += 4) s += bf2f(out[r * Dv + j]) * bf2f(dout[r * Dv + j]);
  s = quad_sum(s);
  if (r < rows && part == 0) delta[r] = s;
}

// dtheta: grid (Lq/64, B); thread -> query t/4, quarter t%4 (of Dv for dp, of Dk for dtheta).
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(
    const bf16_t* __restrict__ theta, const bf16_t* __restrict__ phi, const bf16_t* __restrict__ g,
    const bf16_t* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta, int Lq, int Lk, int Dk, int Dv,
    bf16_t* __restrict__ dtheta) {
  __shared__ float sphi[TILE][DKMAX + 1];
  __shared__ float sg[TILE][4 * DVQMAX + 1];
  const int b = blockIdx.y;
  const int q = blockIdx.x * 64 + (threadIdx.x >> 2);
  const int part = threadIdx.x & 3;
  const int dvq = Dv / 4, dkq = Dk / 4;
  const bool qok = q < Lq;
  float th[DKMAX], dob[DVQMAX], dth[DKQMAX];
#pragma unroll
  for (int d = 0; d < DKMAX; ++d)
    th[d] = (qok && d < Dk) ? bf2f(theta[((int64_t)b * Lq + q) * Dk + d]) : 0.f;
#pragma unroll
  for (int j = 0; j < DVQMAX; ++j)
    dob[j] = (qok && j < dvq) ? bf2f(dout[((int64_t)b * Lq + q) * Dv + part * dvq + j]) : 0.f;
#pragma unroll
  for (int d = 0; d < DKQMAX; ++d) dth[d] = 0.f;
  const float ls = qok ? lse[(int64_t)b * Lq + q] : 0.f;
  const float dl = qok ? delta[(int64_t)b * Lq + q] : 0.f;
  for (int k0 = 0; k0 < Lk; k0 += TILE) {
    __syncthreads();
    for (int i = threadIdx.x; i < TILE * Dk; i += 256) {
      const int r = i / Dk, d = i - r * Dk;
      sphi[r][d] = (k0 + r < Lk) ? bf2f(phi[((int64_t)b * Lk + k0 + r) * Dk + d]) : 0.f;
    }
    for (int i = threadIdx.x; i < TILE * Dv; i += 256) {
      const int r = i / Dv, d = i - r * Dv;
      sg[r][d] = (k0 + r < Lk) ? bf2f(g[((int64_t)b * Lk + k0 + r) * Dv + d]) : 0.f;
    }
    __syncthreads();
    const int kn = min(TILE, Lk - k0);
    for (int r = 0; r < kn; ++r) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < DKMAX; ++d)
        if (d < Dk) s += th[d] * sphi[r][d];
      const float p = __expf(s - ls);
      float dp = 0.f;
#pragma unroll
      for (int j = 0; j < DVQMAX; ++j)
        if (j < dvq) dp += dob[j] * sg[r][part * dvq + j];
      dp = quad_sum(dp);
      const float ds = p * (dp - dl);
#pragma unroll
      for (int d = 0; d < DKQMAX; ++d)
        if (d < dkq) dth[d] += ds * sphi[r][part * dkq + d];
    }
  }
  if (qok) {
#pragma unroll
    for (int d = 0; d < DKQMAX; ++d)
      if (d < dkq) dtheta[((int64_t)b * Lq + q) * Dk + part * dkq + d] = f2bf(dth[d]);
  }
}

// dphi, dg: grid (Lk/64, B); thread -> key t/4, quarter t%4; loops over all queries.
__global__ __launch_bounds__(256) void attn_bwd_k_kernel(
    const bf16_t* __restrict__ theta, const bf16_t* __restrict__ phi, const bf16_t* __restrict__ g,
    const bf16_t* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta, int Lq, int Lk, int Dk, int Dv, bf16_t* __restrict__ dphi,
    bf16_t* __restrict__ dg) {
  __shared__ float sth[TILE][DKMAX + 1];
  __shared__ float sdo[TILE][4 * DVQMAX + 1];
  __shared__ float sls[TILE], sdl[TILE];
  const int b = blockIdx.y;
  const int k = blockIdx.x * 64 + (threadIdx.x >> 2);
  const int part = threadIdx.x & 3;
  const int dvq = Dv / 4, dkq = Dk / 4;
  const bool kok = k < Lk;
  float ph[DKMAX], gq[DVQMAX], dph[DKQMAX], dgq[DVQMAX];
#pragma unroll
  for (int d = 0; d < DKMAX; ++d)
    ph[d] = (kok && d < Dk) ? bf2f(phi[((int64_t)b * Lk + k) * Dk + d]) : 0.f;
#pragma unroll
  for (int j = 0; j < DVQMAX; ++j) {
    gq[j] = (kok && j < dvq) ? bf2f(g[((int64_t)b * Lk + k) * Dv + part * dvq + j]) : 0.f;
    dgq[j] = 0.f;
  }
#pragma unroll
  for (int d = 0; d < DKQMAX; ++d) dph[d] = 0.f;
  for (int q0 = 0; q0 < Lq; q0 += TILE) {
    __syncthreads();
    for (int i = threadIdx.x; i < TILE * Dk; i += 256) {
      const int r = i / Dk, d = i - r * Dk;
      sth[r][d] = (q0 + r < Lq) ? bf2f(theta[((int64_t)b * Lq + q0 + r) * Dk + d]) : 0.f;
    }
    for (int i = threadIdx.x; i < TILE * Dv; i += 256) {
      const int r = i / Dv, d = i - r * Dv;
      sdo[r][d] = (q0 + r < Lq) ? bf2f(dout[((int64_t)b * Lq + q0 + r) * Dv + d]) : 0.f;
    }
    if (threadIdx.x < TILE) {
      const bool ok = q0 + threadIdx.x < Lq;
      sls[threadIdx.x] = ok ? lse[(int64_t)b * Lq + q0 + threadIdx.x] : 0.f;
      sdl[threadIdx.x] = ok ? delta[(int64_t)b * Lq + q0 + threadIdx.x] : 0.f;
    }
    __syncthreads();
    const int qn = min(TILE, Lq - q0);
    for (int r = 0; r < qn; ++r) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < DKMAX; ++d)
        if (d < Dk) s += ph[d] * sth[r][d];
      const float p = __expf(s - sls[r]);
      float dp = 0.f;
#pragma unroll
      for (int j = 0; j < DVQMAX; ++j)
        if (j < dvq) dp += gq[j] * sdo[r][part * dvq + j];
      dp = quad_sum(dp);
      const float ds = p * (dp - sdl[r]);
#pragma unroll
      for (int d = 0; d < DKQMAX; ++d)
        if (d < dkq) dph[d] += ds * sth[r][part * dkq + d];
#pragma unroll
      for (int j = 0; j < DVQMAX; ++j)
        if (j < dvq) dgq[j] += p * sdo[r][part * dvq + j];
    }
  }
  if (kok) {
#pragma unroll
    for (int d = 0; d < DKQMAX; ++d)
      if (d < dkq) dphi[((int64_t)b * Lk + k) * Dk + part * dkq + d] = f2bf(dph[d]);
#pragma unroll
    for (int j = 0; j < DVQMAX; ++j)
      if (j < dvq) dg[((int64_t)b * Lk + k) * Dv + part * dvq + j] = f2bf(dgq[j]);
  }
}

// -------------------------------------------------------------------------------------------
// v2: MFMA (v_mfma_f32_32x32x16_bf16) flash-style kernels for Lq % 128 == 0, Lk % 64 == 0,
// Dk <= 64, Dv <= 128.  Score tiles are computed TRANSPOSED (S^T[key][query]) so that a lane owns one
// query column and 16 keys of it: softmax statistics are per-lane (plus one lane^32 exchange), and
// the probabilities feed the second MFMA straight from the accumulator registers -- the key order
// inside a 16-wide MFMA K step is a permutation, which is harmless as long as the other operand
// (V^T / K^T / dO^T / Q^T, staged transposed in LDS) uses the same one:
//   element e of lane half h of K-step ks  <->  row 16*ks + 4*h + (e & 3) + 8*(e >> 2).
// Dk / Dv are zero-padded to DKP / DVP in LDS.
// -------------------------------------------------------------------------------------------
constexpr int AT = 64;          // keys (fwd, bwd_q) / queries (bwd_k) per LDS stage
constexpr int ATP = AT + 4;     // padded row of the transposed tiles: 34 dwords -> the 2-byte
                                // transposed stores of a wave hit 32 distinct banks, the 8-byte
                                // fragment reads are 2-way (their minimum)

__device__ __forceinline__ bf16x8_t pack_bf16x8(const float* v) {
  s16x8_t r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (short)f2bf(v[e]);
  return __builtin_bit_cast(bf16x8_t, r);
}
// two 4-element (8-byte) pieces of a transposed LDS row: rows base + 4h + {0..3} and + 8
__device__ __forceinline__ bf16x8_t read_perm(const bf16_t* row, int off) {
  const uint2 lo = *reinterpret_cast<const uint2*>(row + off);
  const uint2 hi = *reinterpret_cast<const uint2*>(row + off + 8);
  uint4 q = make_uint4(lo.x, lo.y, hi.x, hi.y);
  return __builtin_bit_cast(bf16x8_t, q);
}
// Staging of a [AT rows][D] global tile (row-major, D % 4 == 0) goes through REGISTERS: the 8-byte
// chunks of the NEXT tile are loaded right after the barrier that publishes the current one and
// stay in flight while it is multiplied (the round-2 kernels loaded and stored element by element
// between two barriers: 24 dependent 2-byte round trips per tile, 10x the MFMA time of the tile).
// DP: padded width (multiple of 16): chunk c of thread t is row (t + 256 c) / (DP/4), channels
// 4 ((t + 256 c) % (DP/4)) ..+3, zero beyond D.
template <int DP, int NT>
struct TileRegs {
  static constexpr int CH = DP / 4;
  static constexpr int TOTAL = AT * CH;
  static constexpr int N = (TOTAL + NT - 1) / NT;
  uint2 r[N];
};
template <int DP, int NT>
__device__ __forceinline__ void tile_load(TileRegs<DP, NT>& t, const bf16_t* __restrict__ src, int D,
                                          int64_t row0) {
  using T = TileRegs<DP, NT>;
#pragma unroll
  for (int c = 0; c < T::N; ++c) {
    const int i = threadIdx.x + c * NT;
    const int row = i / T::CH, d = (i - row * T::CH) * 4;
    t.r[c] = (d < D && (T::TOTAL % NT == 0 || i < T::TOTAL))
                 ? *reinterpret_cast<const uint2*>(src + (row0 + row) * D + d) : make_uint2(0u, 0u);
  }
}
// row-major LDS image, pitch DS + 8 elements; only channels < DS are kept (DS <= DP)
template <int DP, int DS, int NT>
__device__ __forceinline__ void tile_store_rows(const TileRegs<DP, NT>& t, bf16_t* dst) {
  using T = TileRegs<DP, NT>;
#pragma unroll
  for (int c = 0; c < T::N; ++c) {
    const int i = threadIdx.x + c * NT;
    const int row = i / T::CH, d = (i - row * T::CH) * 4;
    if ((DS == DP || d < DS) && (T::TOTAL % NT == 0 || i < T::TOTAL))
      *reinterpret_cast<uint2*>(dst + row * (DS + 8) + d) = t.r[c];
  }
}
// transposed LDS image dst[d][row], pitch ATP
template <int DP, int NT>
__device__ __forceinline__ void tile_store_cols(const TileRegs<DP, NT>& t, bf16_t* dst) {
  using T = TileRegs<DP, NT>;
#pragma unroll
  for (int c = 0; c < T::N; ++c) {
    const int i = threadIdx.x + c * NT;
    const int row = i / T::CH, d = (i - row * T::CH) * 4;
    if (T::TOTAL % NT != 0 && i >= T::TOTAL) continue;
    dst[(d + 0) * ATP + row] = (bf16_t)(t.r[c].x & 0xffffu);
    dst[(d + 1) * ATP + row] = (bf16_t)(t.r[c].x >> 16);
    dst[(d + 2) * ATP + row] = (bf16_t)(t.r[c].y & 0xffffu);
    dst[(d + 3) * ATP + row] = (bf16_t)(t.r[c].y >> 16);
  }
}
// register fragment of a global row-major matrix: row = lane & 31, columns kd*16 + (lane>>5)*8 + e
__device__ __forceinline__ bf16x8_t load_frag(const bf16_t* __restrict__ src, int D, int64_t row,
                                              int kd, int half) {
  s16x8_t r;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int d = kd * 16 + half * 8 + e;
    r[e] = d < D ? (short)src[row * D + d] : (short)0;
  }
  return __builtin_bit_cast(bf16x8_t, r);
}

// forward: grid (Lq / (32 NW), B); wave -> 32 queries.  NW = 8 waves halve the staging work and the
// barriers per query.
template <int DKP, int DVP, int NW>
__global__ __launch_bounds__(NW * 64) void attn_fwd_mfma_kernel(
    const bf16_t* __restrict__ theta, const bf16_t* __restrict__ phi, const bf16_t* __restrict__ g,
    int Lq, int Lk, int Dk, int Dv, bf16_t* __restrict__ out, float* __restrict__ lse) {
  constexpr int KD = DKP / 16, VT = DVP / 32;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[AT * (DKP + 8)];
  __shared__ __attribute__((aligned(16))) bf16_t Vt[DVP * ATP];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int64_t q = (int64_t)b * Lq + blockIdx.x * (32 * NW) + wave * 32 + col;
  bf16x8_t qf[KD];
#pragma unroll
  for (int kd = 0; kd < KD; ++kd) qf[kd] = load_frag(theta, Dk, q, kd, half);
  f32x16_t o[VT];
#pragma unroll
  for (int t = 0; t < VT; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) o[t][v] = 0.f;
  float m = -3.0e38f, l = 0.f;
  TileRegs<DKP, NW * 64> kreg;
  TileRegs<DVP, NW * 64> vreg;
  tile_load<DKP, NW * 64>(kreg, phi, Dk, (int64_t)b * Lk);
  tile_load<DVP, NW * 64>(vreg, g, Dv, (int64_t)b * Lk);
  for (int k0 = 0; k0 < Lk; k0 += AT) {
    __syncthreads();   // every wave is done with the previous tile
    tile_store_rows<DKP, DKP, NW * 64>(kreg, Ks);
    tile_store_cols<DVP, NW * 64>(vreg, Vt);
    __syncthreads();
    if (k0 + AT < Lk) {
      tile_load<DKP, NW * 64>(kreg, phi, Dk, (int64_t)b * Lk + k0 + AT);
      tile_load<DVP, NW * 64>(vreg, g, Dv, (int64_t)b * Lk + k0 + AT);
    }
#pragma unroll
    for (int sub = 0; sub < AT / 32; ++sub) {
      f32x16_t s;
#pragma unroll
      for (int v = 0; v < 16; ++v) s[v] = 0.f;
#pragma unroll
      for (int kd = 0; kd < KD; ++kd) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(
            Ks + (sub * 32 + col) * (DKP + 8) + kd * 16 + half * 8);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kd], s, 0, 0, 0);
      }
      float mx = s[0];
#pragma unroll
      for (int v = 1; v < 16; ++v) mx = fmaxf(mx, s[v]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m, mx);
      const float corr = __expf(m - mn);
      m = mn;
      l *= corr;
#pragma unroll
      for (int t = 0; t < VT; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) o[t][v] *= corr;
      float p[16];
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        p[v] = __expf(s[v] - mn);
        l += p[v];
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8_t pf = pack_bf16x8(p + 8 * ks);
#pragma unroll
        for (int t = 0; t < VT; ++t) {
          const bf16x8_t vf = read_perm(Vt + (t * 32 + col) * ATP, sub * 32 + 16 * ks + 4 * half);
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[t], 0, 0, 0);
        }
      }
    }
  }
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.f / l;
#pragma unroll
  for (int t = 0; t < VT; ++t)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int dv = t * 32 + qd * 8 + 4 * half;
      if (dv < Dv) {   // Dv % 4 == 0
        uint2 w;
        w.x = (uint32_t)f2bf(o[t][qd * 4 + 0] * inv) | ((uint32_t)f2bf(o[t][qd * 4 + 1] * inv) << 16);
        w.y = (uint32_t)f2bf(o[t][qd * 4 + 2] * inv) | ((uint32_t)f2bf(o[t][qd * 4 + 3] * inv) << 16);
        *reinterpret_cast<uint2*>(out + q * Dv + dv) = w;
      }
    }
  if (half == 0) lse[q] = m + __logf(l);
}

// dtheta: grid (Lq / (32 NW), B); wave -> 32 queries, loops over the keys.
template <int DKP, int DVP, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_q_mfma_kernel(
    const bf16_t* __restrict__ theta, const bf16_t* __restrict__ phi, const bf16_t* __restrict__ g,
    const bf16_t* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta, int Lq, int Lk, int Dk, int Dv,
    bf16_t* __restrict__ dtheta) {
  constexpr int KD = DKP / 16, VD = DVP / 16, KT = (DKP + 31) / 32;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[AT * (DKP + 8)];
  __shared__ __attribute__((aligned(16))) bf16_t Kt[KT * 32 * ATP];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[AT * (DVP + 8)];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int64_t q = (int64_t)b * Lq + blockIdx.x * (32 * NW) + wave * 32 + col;
  bf16x8_t qf[KD], dof[VD];
#pragma unroll
  for (int kd = 0; kd < KD; ++kd) qf[kd] = load_frag(theta, Dk, q, kd, half);
#pragma unroll
  for (int vd = 0; vd < VD; ++vd) dof[vd] = load_frag(dout, Dv, q, vd, half);
  const float ls = lse[q], dl = delta[q];
  f32x16_t dq[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) dq[t][v] = 0.f;
  TileRegs<KT * 32, NW * 64> kreg;
  TileRegs<DVP, NW * 64> vreg;
  tile_load<KT * 32, NW * 64>(kreg, phi, Dk, (int64_t)b * Lk);
  tile_load<DVP, NW * 64>(vreg, g, Dv, (int64_t)b * Lk);
  for (int k0 = 0; k0 < Lk; k0 += AT) {
    __syncthreads();
    tile_store_rows<KT * 32, DKP, NW * 64>(kreg, Ks);
    tile_store_cols<KT * 32, NW * 64>(kreg, Kt);
    tile_store_rows<DVP, DVP, NW * 64>(vreg, Vs);
    __syncthreads();
    if (k0 + AT < Lk) {
      tile_load<KT * 32, NW * 64>(kreg, phi, Dk, (int64_t)b * Lk + k0 + AT);
      tile_load<DVP, NW * 64>(vreg, g, Dv, (int64_t)b * Lk + k0 + AT);
    }
#pragma unroll
    for (int sub = 0; sub < AT / 32; ++sub) {
      f32x16_t s, dp;
#pragma unroll
      for (int v = 0; v < 16; ++v) s[v] = dp[v] = 0.f;
#pragma unroll
      for (int kd = 0; kd < KD; ++kd) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(
            Ks + (sub * 32 + col) * (DKP + 8) + kd * 16 + half * 8);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kd], s, 0, 0, 0);
      }
#pragma unroll
      for (int vd = 0; vd < VD; ++vd) {
        const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(
            Vs + (sub * 32 + col) * (DVP + 8) + vd * 16 + half * 8);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[vd], dp, 0, 0, 0);
      }
      float ds[16];
#pragma unroll
      for (int v = 0; v < 16; ++v) ds[v] = __expf(s[v] - ls) * (dp[v] - dl);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8_t dsf = pack_bf16x8(ds + 8 * ks);
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          const bf16x8_t kf = read_perm(Kt + (t * 32 + col) * ATP, sub * 32 + 16 * ks + 4 * half);
          dq[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, dsf, dq[t], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int d = t * 32 + qd * 8 + 4 * half;
      if (d < Dk) {   // Dk % 4 == 0
        uint2 w;
        w.x = (uint32_t)f2bf(dq[t][qd * 4 + 0]) | ((uint32_t)f2bf(dq[t][qd * 4 + 1]) << 16);
        w.y = (uint32_t)f2bf(dq[t][qd * 4 + 2]) | ((uint32_t)f2bf(dq[t][qd * 4 + 3]) << 16);
        *reinterpret_cast<uint2*>(dtheta + q * Dk + d) = w;
      }
    }
}

// dphi, dg: grid (Lk / (32 NW), B); wave -> 32 keys, loops over the queries.
template <int DKP, int DVP, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_k_mfma_kernel(
    const bf16_t* __restrict__ theta, const bf16_t* __restrict__ phi, const bf16_t* __restrict__ g,
    const bf16_t* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta, int Lq, int Lk, int Dk, int Dv, bf16_t* __restrict__ dphi,
    bf16_t* __restrict__ dg) {
  constexpr int KD = DKP / 16, VD = DVP / 16, KT = (DKP + 31) / 32, VT = DVP / 32;
  __shared__ __attribute__((aligned(16))) bf16_t Qs[AT * (DKP + 8)];
  __shared__ __attribute__((aligned(16))) bf16_t Qt[KT * 32 * ATP];
  __shared__ __attribute__((aligned(16))) bf16_t Os[AT * (DVP + 8)];
  __shared__ __attribute__((aligned(16))) bf16_t Ot[DVP * ATP];
  __shared__ float sls[AT], sdl[AT];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int64_t k = (int64_t)b * Lk + blockIdx.x * (32 * NW) + wave * 32 + col;
  bf16x8_t kf[KD], vf[VD];
#pragma unroll
  for (int kd = 0; kd < KD; ++kd) kf[kd] = load_frag(phi, Dk, k, kd, half);
#pragma unroll
  for (int vd = 0; vd < VD; ++vd) vf[vd] = load_frag(g, Dv, k, vd, half);
  f32x16_t dk[KT], dv[VT];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) dk[t][v] = 0.f;
#pragma unroll
  for (int t = 0; t < VT; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) dv[t][v] = 0.f;
  TileRegs<KT * 32, NW * 64> qreg;
  TileRegs<DVP, NW * 64> oreg;
  float lreg = 0.f, dreg = 0.f;
  tile_load<KT * 32, NW * 64>(qreg, theta, Dk, (int64_t)b * Lq);
  tile_load<DVP, NW * 64>(oreg, dout, Dv, (int64_t)b * Lq);
  if (threadIdx.x < AT) {
    lreg = lse[(int64_t)b * Lq + threadIdx.x];
    dreg = delta[(int64_t)b * Lq + threadIdx.x];
  }
  for (int q0 = 0; q0 < Lq; q0 += AT) {
    __syncthreads();
    tile_store_rows<KT * 32, DKP, NW * 64>(qreg, Qs);
    tile_store_cols<KT * 32, NW * 64>(qreg, Qt);
    tile_store_rows<DVP, DVP, NW * 64>(oreg, Os);
    tile_store_cols<DVP, NW * 64>(oreg, Ot);
    if (threadIdx.x < AT) {
      sls[threadIdx.x] = lreg;
      sdl[threadIdx.x] = dreg;
    }
    __syncthreads();
    if (q0 + AT < Lq) {
      tile_load<KT * 32, NW * 64>(qreg, theta, Dk, (int64_t)b * Lq + q0 + AT);
      tile_load<DVP, NW * 64>(oreg, dout, Dv, (int64_t)b * Lq + q0 + AT);
      if (threadIdx.x < AT) {
        lreg = lse[(int64_t)b * Lq + q0 + AT + threadIdx.x];
        dreg = delta[(int64_t)b * Lq + q0 + AT + threadIdx.x];
      }
    }
#pragma unroll
    for (int sub = 0; sub < AT / 32; ++sub) {
      // S[q][key], dP[q][key]: lane owns key `col`, queries 4*half + (v & 3) + 8*(v >> 2)
      f32x16_t s, dp;
#pragma unroll
      for (int v = 0; v < 16; ++v) s[v] = dp[v] = 0.f;
#pragma unroll
      for (int kd = 0; kd < KD; ++kd) {
        const bf16x8_t qf = *reinterpret_cast<const bf16x8_t*>(
            Qs + (sub * 32 + col) * (DKP + 8) + kd * 16 + half * 8);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[kd], s, 0, 0, 0);
      }
#pragma unroll
      for (int vd = 0; vd < VD; ++vd) {
        const bf16x8_t of = *reinterpret_cast<const bf16x8_t*>(
            Os + (sub * 32 + col) * (DVP + 8) + vd * 16 + half * 8);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of, vf[vd], dp, 0, 0, 0);
      }
      float p[16], ds[16];
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int qi = sub * 32 + 4 * half + (v & 3) + 8 * (v >> 2);
        p[v] = __expf(s[v] - sls[qi]);
        ds[v] = p[v] * (dp[v] - sdl[qi]);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8_t pf = pack_bf16x8(p + 8 * ks);
        const bf16x8_t dsf = pack_bf16x8(ds + 8 * ks);
        const int off = sub * 32 + 16 * ks + 4 * half;
#pragma unroll
        for (int t = 0; t < VT; ++t) {
          const bf16x8_t of = read_perm(Ot + (t * 32 + col) * ATP, off);
          dv[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of, pf, dv[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          const bf16x8_t qf = read_perm(Qt + (t * 32 + col) * ATP, off);
          dk[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, dsf, dk[t], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int d = t * 32 + qd * 8 + 4 * half;
      if (d < Dk) {
        uint2 w;
        w.x = (uint32_t)f2bf(dk[t][qd * 4 + 0]) | ((uint32_t)f2bf(dk[t][qd * 4 + 1]) << 16);
        w.y = (uint32_t)f2bf(dk[t][qd * 4 + 2]) | ((uint32_t)f2bf(dk[t][qd * 4 + 3]) << 16);
        *reinterpret_cast<uint2*>(dphi + k * Dk + d) = w;
      }
    }
#pragma unroll
  for (int t = 0; t < VT; ++t)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int d = t * 32 + qd * 8 + 4 * half;
      if (d < Dv) {
        uint2 w;
        w.x = (uint32_t)f2bf(dv[t][qd * 4 + 0]) | ((uint32_t)f2bf(dv[t][qd * 4 + 1]) << 16);
        w.y = (uint32_t)f2bf(dv[t][qd * 4 + 2]) | ((uint32_t)f2bf(dv[t][qd * 4 + 3]) << 16);
        *reinterpret_cast<uint2*>(dg + k * Dv + d) = w;
      }
    }
}

bool attn_mfma_ok(int Lq, int Lk, int Dk, int Dv) {
  return (Lq % 128) == 0 && (Lk % 128) == 0 && Dk <= 64 && Dv <= 128 && (Dk % 4) == 0 &&
         (Dv % 4) == 0;
}
// 8-wave workgroups (256 rows) when the rows divide and the grid still fills the chip twice over
// (CGAMD_ATTN_NW8=0/1 forces)
bool attn_nw8(int L, int B) {
  const char* e = getenv("CGAMD_ATTN_NW8");   // read per call: the tests toggle it in-process
  const int env = e ? atoi(e) : -1;
  if (L % 256) return false;
  if (env >= 0) return env != 0;
  return (int64_t)(L / 256) * B >= 512;
}
// padded sizes: DKP in {16, 32, 64}; DVP in {32, 64, 96, 128}
#define CG_ATTN_DISPATCH(KERNEL, GRID, ...)                                                   \
  do {                                                                                        \
    if (nw8) CG_ATTN_DISPATCH_NW(KERNEL, 8, GRID, __VA_ARGS__);                               \
    else CG_ATTN_DISPATCH_NW(KERNEL, 4, GRID, __VA_ARGS__);                                   \
  } while (0)
#define CG_ATTN_DISPATCH_NW(KERNEL, NW, GRID, ...)                                            \
  do {                                                                                        \
    const int dkp_ = Dk <= 16 ? 16 : (Dk <= 32 ? 32 : 64);                                    \
    const int dvp_ = (Dv + 31) / 32 * 32;                                                     \
    if (dkp_ == 16 && dvp_ == 32) KERNEL<16, 32, NW><<<GRID, NW * 64, 0, st>>>(__VA_ARGS__);          \
    else if (dkp_ == 16 && dvp_ == 64) KERNEL<16, 64, NW><<<GRID, NW * 64, 0, st>>>(__VA_ARGS__);     \
    else if (dkp_ == 16 && dvp_ == 96) KERNEL<16, 96, NW><<<GRID, NW * 64, 0, st>>>(__VA_ARGS__);     \
    else if (dkp_ == 16) KERNEL<16, 128, NW><<<GRID, NW * 64, 0, st>>>(__VA_ARGS__);                  \
    else if (dkp_ == 32 && dvp_ == 32) KERNEL<32, 32, NW><<<GRID, NW * 64, 0, st>>>(__VA_ARGS__);     \
    else if (dkp_ == 32 && dvp_ == 64) KERNEL<32, 64, NW><<<GRID, NW * 64, 0, st>>>(__VA_ARGS__);     \
    else if (dkp_ == 32 && dvp_ == 96) KERNEL<32, 96, NW><<<GRID, NW * 64, 0, st>>>(__VA_ARGS__);     \
    else if (dkp_ == 32) KERNEL<32, 128, NW><<<GRID, NW * 64, 0, st>>>(__VA_ARGS__);                  \
    else if (dvp_ == 32) KERNEL<64, 32, NW><<<GRID, NW * 64, 0, st>>>(__VA_ARGS__);                   \
    else if (dvp_ == 64) KERNEL<64, 64, NW><<<GRID, NW * 64, 0, st>>>(__VA_ARGS__);                   \
    else if (dvp_ == 96) KERNEL<64, 96, NW><<<GRID, NW * 64, 0, st>>>(__VA_ARGS__);                   \
    else KERNEL<64, 128, NW><<<GRID, NW * 64, 0, st>>>(__VA_ARGS__);                                  \
  } while (0)

int check_attn(int B, int Lq, int Lk, int Dk, int Dv, const char* who) {
  if (B <= 0 || Lq <= 0 || Lk <= 0 || Dk <= 0 || Dv <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "%s: non-positive dimension", who);
  if (Dk > DKMAX || Dv > 4 * DVQMAX || (Dv % 4) || (Dk % 4))
    CG_FAIL(CG_ERR_UNSUPPORTED, "%s: Dk=%d (<=%d, %%4) Dv=%d (<=%d, %%4) unsupported", who, Dk,
            DKMAX, Dv, 4 * DVQMAX);
  return CG_OK;
}

}  // namespace

extern "C" int cg_attention_fwd(const void* theta, const void* phi, const void* g, int B, int Lq,
                                int Lk, int Dk, int Dv, void* out, float* lse, cgStream stream) {
  int rc = check_attn(B, Lq, Lk, Dk, Dv, "cg_attention_fwd");
  if (rc) return rc;
  if (!theta || !phi || !g || !out || !lse) CG_FAIL(CG_ERR_BAD_ARG, "cg_attention_fwd: null");
  hipStream_t st = (hipStream_t)stream;
  if (attn_mfma_ok(Lq, Lk, Dk, Dv)) {
    const bool nw8 = attn_nw8(Lq, B);
    dim3 mgrid(Lq / (nw8 ? 256 : 128), B);
    CG_ATTN_DISPATCH(attn_fwd_mfma_kernel, mgrid, (const bf16_t*)theta, (const bf16_t*)phi,
                     (const bf16_t*)g, Lq, Lk, Dk, Dv, (bf16_t*)out, lse);
    CG_CHECK_LAUNCH("cg_attention_fwd(mfma)");
    return CG_OK;
  }
  dim3 grid(cdiv(Lq, 64), B);
  attn_fwd_kernel<<<grid, 256, 0, st>>>((const bf16_t*)theta, (const bf16_t*)phi,
                                        (const bf16_t*)g, Lq, Lk, Dk, Dv, (bf16_t*)out, lse);
  CG_CHECK_LAUNCH("cg_attention_fwd");
  return CG_OK;
}

extern "C" size_t cg_attention_bwd_workspace_bytes(int B, int Lq, int Lk, int Dk, int Dv) {
  (void)Lk; (void)Dk; (void)Dv;
  if (B <= 0 || Lq <= 0) return 0;
  return align_up((size_t)B * Lq * sizeof(float), 256);
}

extern "C" int cg_attention_bwd(const void* theta, const void* phi, const void* g, const void* out,
                                const float* lse, const void* dout, int B, int Lq, int Lk, int Dk,
                                int Dv, void* dtheta, void* dphi, void* dg, void* ws,
                                size_t ws_bytes, cgStream stream) {
  int rc = check_attn(B, Lq, Lk, Dk, Dv, "cg_attention_bwd");
  if (rc) return rc;
  if (!theta || !phi || !g || !out || !lse || !dout || !dtheta || !dphi || !dg)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_attention_bwd: null");
  if (!ws || ws_bytes < cg_attention_bwd_workspace_bytes(B, Lq, Lk, Dk, Dv))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_attention_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* delta = (float*)ws;
  const int64_t rows = (int64_t)B * Lq;
  attn_delta_kernel<<<cdiv(rows, 64), 256, 0, st>>>((const bf16_t*)out, (const bf16_t*)dout, rows,
                                                    Dv, delta);
  CG_CHECK_LAUNCH("cg_attention_bwd(delta)");
  if (attn_mfma_ok(Lq, Lk, Dk, Dv)) {
    bool nw8 = attn_nw8(Lq, B);
    dim3 mq(Lq / (nw8 ? 256 : 128), B);
    CG_ATTN_DISPATCH(attn_bwd_q_mfma_kernel, mq, (const bf16_t*)theta, (const bf16_t*)phi,
                     (const bf16_t*)g, (const bf16_t*)dout, lse, delta, Lq, Lk, Dk, Dv,
                     (bf16_t*)dtheta);
    nw8 = attn_nw8(Lk, B);
    dim3 mk(Lk / (nw8 ? 256 : 128), B);
    CG_ATTN_DISPATCH(attn_bwd_k_mfma_kernel, mk, (const bf16_t*)theta, (const bf16_t*)phi,
                     (const bf16_t*)g, (const bf16_t*)dout, lse, delta, Lq, Lk, Dk, Dv,
                     (bf16_t*)dphi, (bf16_t*)dg);
    CG_CHECK_LAUNCH("cg_attention_bwd(mfma)");
    return CG_OK;
  }
  dim3 gq(cdiv(Lq, 64), B);
  attn_bwd_q_kernel<<<gq, 256, 0, st>>>((const bf16_t*)theta, (const bf16_t*)phi, (const bf16_t*)g,
                                        (const bf16_t*)dout, lse, delta, Lq, Lk, Dk, Dv,
                                        (bf16_t*)dtheta);
  CG_CHECK_LAUNCH("cg_attention_bwd(q)");
  dim3 gk(cdiv(Lk, 64), B);
  attn_bwd_k_kernel<<<gk, 256, 0, st>>>((const bf16_t*)theta, (const bf16_t*)phi, (const bf16_t*)g,
                                        (const bf16_t*)dout, lse, delta, Lq, Lk, Dk, Dv,
                                        (bf16_t*)dphi, (bf16_t*)dg);
  CG_CHECK_LAUNCH("cg_attention_bwd(k)");
  return CG_OK;
}
