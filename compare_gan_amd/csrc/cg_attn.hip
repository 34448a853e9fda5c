// Self-attention core of arch_ops.non_local_block (arch_ops.py:744-753):
//   attn = softmax(theta phi^T), out = attn g   per image, never materialising [B, Lq, Lk].
// v1: streaming online-softmax on the vector ALU (Dk = C/8 is 12..48, too thin for MFMA K);
// four lanes cooperate on one query (forward, dtheta) or one key (dphi, dg), each owning a
// quarter of the value / key channels; K/V (resp. Q/dO) tiles of 64 rows are staged in LDS.
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
  dim3 grid(cdiv(Lq, 64), B);
  attn_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)theta, (const bf16_t*)phi,
                                                        (const bf16_t*)g, Lq, Lk, Dk, Dv,
                                                        (bf16_t*)out, lse);
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
