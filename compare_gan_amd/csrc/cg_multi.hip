// Multi-tensor forms of the per-weight bookkeeping of a network call: spectral normalisation of ALL
// spectrally-normalised weights (arch_ops.py:453-535), its backward, and the fp32 -> bf16 MFMA
// operand preparation of ALL convolution / linear weights -- a handful of launches per network call
// instead of several per weight (the small configs are launch-bound: SURVEY.md section 7).
//
// Per-tensor pointers travel BY VALUE in the kernel arguments (chunks of up to CG_MULTI_MAX tensors
// per launch): nothing is uploaded, so the launches are hipGraph-capturable even though the
// gradient / output tensors change from call to call.
#include "cg_conv_fast.h"

#include <vector>

namespace {

constexpr int MAXT = CG_MULTI_MAX;

// ---------------------------------------------------------------------------------------------
// spectral norm: two mat-vec passes per weight.
//   mode 0 (left,  u in R^K ): pass A = column reduction  t = W^T u -> v = l2n(t)
//                              pass B = row dots          s = W v   -> u' = l2n(s), sigma = |s|^2 rs
//   mode 1 (right, u in R^Co): pass A = row dots          t = W u   -> v = l2n(t)
//                              pass B = column reduction  s = W^T v -> u' = l2n(s), sigma likewise
// ---------------------------------------------------------------------------------------------
struct SNChunk {
  const float* w[MAXT];
  float* u[MAXT];        // persisted vector (read in pass A, rewritten by finalise B)
  float* u_out[MAXT];    // per-call copy of u'
  float* v_out[MAXT];    // per-call v
  float* sigma[MAXT];    // [2] sigma, 1 / sigma
  float* part[MAXT];     // workspace: column-reduction partials [splits][Co], then row-dot output [K]
  int K[MAXT], Co[MAXT], mode[MAXT], splits[MAXT];
  int blk[MAXT + 1];     // block prefix of the current mat-vec launch
  int n;
};

__device__ __forceinline__ int find_item(const int* blk, int n, int b) {
  int i = 0;
  while (i + 1 < n && blk[i + 1] <= b) ++i;
  return i;
}

// colred: block = 64 columns x one row split; rowdot: block = 4 rows (one wave each).
// `second` selects pass B (input vector v_out, operation flipped).
__global__ __launch_bounds__(256) void sn_matvec_kernel(SNChunk c, int second) {
  __shared__ float sm[4][64];
  const int it = find_item(c.blk, c.n, blockIdx.x);
  const int b = blockIdx.x - c.blk[it];
  const float* __restrict__ w = c.w[it];
  const int K = c.K[it], Co = c.Co[it];
  const bool colred = (c.mode[it] == 0) != (second != 0);
  const float* __restrict__ vec = second ? c.v_out[it] : c.u[it];
  float* part = c.part[it];
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (colred) {
    const int ct = (Co + 63) / 64;
    const int cblk = b % ct, z = b / ct;
    const int splits = c.splits[it];
    const int kps = (K + splits - 1) / splits;
    const int k0 = z * kps, k1 = min(K, k0 + kps);
    const int co = cblk * 64 + l;
    float s = 0.f;
    if (co < Co) {
      // four independent loads in flight per lane (the loop is a chain of dependent adds otherwise)
      int k = k0 + wv;
      float s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (; k + 12 < k1; k += 16) {
        const float w0 = w[(int64_t)k * Co + co], w1 = w[(int64_t)(k + 4) * Co + co];
        const float w2 = w[(int64_t)(k + 8) * Co + co], w3 = w[(int64_t)(k + 12) * Co + co];
        s += vec[k] * w0;
        s1 += vec[k + 4] * w1;
        s2 += vec[k + 8] * w2;
        s3 += vec[k + 12] * w3;
      }
      for (; k < k1; k += 4) s += vec[k] * w[(int64_t)k * Co + co];
      s = (s + s1) + (s2 + s3);
    }
    sm[wv][l] = s;
    __syncthreads();
    if (wv == 0 && co < Co) part[(int64_t)z * Co + co] = sm[0][l] + sm[1][l] + sm[2][l] + sm[3][l];
  } else {
    const int k = b * 4 + wv;
    if (k >= K) return;
    float s = 0.f;
    const float* __restrict__ wr = w + (int64_t)k * Co;
    if ((Co & 3) == 0 && (((uintptr_t)wr | (uintptr_t)vec) & 15) == 0) {
      // 16-byte loads, two in flight
      float s1 = 0.f;
      int co = l * 4;
      for (; co + 256 < Co; co += 512) {
        const float4 a0 = *reinterpret_cast<const float4*>(wr + co);
        const float4 a1 = *reinterpret_cast<const float4*>(wr + co + 256);
        const float4 v0 = *reinterpret_cast<const float4*>(vec + co);
        const float4 v1 = *reinterpret_cast<const float4*>(vec + co + 256);
        s += (a0.x * v0.x + a0.y * v0.y) + (a0.z * v0.z + a0.w * v0.w);
        s1 += (a1.x * v1.x + a1.y * v1.y) + (a1.z * v1.z + a1.w * v1.w);
      }
      for (; co < Co; co += 256) {
        const float4 a0 = *reinterpret_cast<const float4*>(wr + co);
        const float4 v0 = *reinterpret_cast<const float4*>(vec + co);
        s += (a0.x * v0.x + a0.y * v0.y) + (a0.z * v0.z + a0.w * v0.w);
      }
      s += s1;
    } else {
      for (int co = l; co < Co; co += 64) s += wr[co] * vec[co];
    }
    s = wave_sum(s);
    if (l == 0) part[(int64_t)c.splits[it] * Co + k] = s;   // row-dot area follows the partials
  }
}

// one block per weight: raw = sum of partials (or the row-dot output), out = l2n(raw); pass B also
// writes sigma and the persisted u.
__global__ __launch_bounds__(1024) void sn_finalize_kernel(SNChunk c, int second, float eps) {
  __shared__ float sm[16];
  const int it = blockIdx.x;
  const int K = c.K[it], Co = c.Co[it];
  const bool colred = (c.mode[it] == 0) != (second != 0);
  const int n = colred ? Co : K;
  const int splits = colred ? c.splits[it] : 1;
  const float* part = colred ? c.part[it] : c.part[it] + (int64_t)c.splits[it] * Co;
  float* out = second ? c.u_out[it] : c.v_out[it];
  float ss = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float r = 0.f;
    for (int z = 0; z < splits; ++z) r += part[(int64_t)z * n + i];
    out[i] = r;
    ss += r * r;
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) tot += sm[i];
  const float rs = rsqrtf(fmaxf(tot, eps));
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float v = out[i] * rs;
    out[i] = v;
    if (second) c.u[it][i] = v;   // arch_ops.py:516: u is persisted after the round
  }
  if (second && threadIdx.x == 0) {
    const float s = tot * rs;       // sigma = u'^T W v = |W v|^2 * rsqrt(...)
    c.sigma[it][0] = s;
    c.sigma[it][1] = 1.f / s;
  }
}

// ---------------------------------------------------------------------------------------------
// elementwise multi-tensor passes: wbar = w / sigma ; SN backward ; operand preparation
// ---------------------------------------------------------------------------------------------
struct ScaleChunk {
  const float* w[MAXT];
  const float* sigma[MAXT];   // [2], uses [1] = 1 / sigma
  float* out[MAXT];
  int64_t n[MAXT];
  int blk[MAXT + 1];
  int cnt;
};
// Elements per block.  A block of 256 threads moves its chunk as 16-byte accesses, four in flight per
// lane (the round-3 form -- 8192 elements per block, 4-byte accesses, 32 dependent trips -- ran a
// 1 M-parameter network on 134 workgroups at 11-19 us per pass: profiles/r04_cifar_kernel_stats.csv)
constexpr int EW_CHUNK = 4096;

__device__ __forceinline__ bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

__global__ __launch_bounds__(256) void scale_multi_kernel(ScaleChunk c) {
  const int it = find_item(c.blk, c.cnt, blockIdx.x);
  const int64_t base = (int64_t)(blockIdx.x - c.blk[it]) * EW_CHUNK;
  const int64_t end = min(c.n[it], base + EW_CHUNK);
  const float s = c.sigma[it][1];
  const float* __restrict__ w = c.w[it];
  float* __restrict__ o = c.out[it];
  if (aligned16(w) && aligned16(o)) {
    const int64_t vend = base + ((end - base) & ~3ll);
#pragma unroll 4
    for (int64_t i = base + threadIdx.x * 4; i < vend; i += 1024) {
      float4 v = *reinterpret_cast<const float4*>(w + i);
      v.x *= s; v.y *= s; v.z *= s; v.w *= s;
      *reinterpret_cast<float4*>(o + i) = v;
    }
    for (int64_t i = vend + threadIdx.x; i < end; i += 256) o[i] = w[i] * s;
    return;
  }
  for (int64_t i = base + threadIdx.x; i < end; i += 256) o[i] = w[i] * s;
}

struct SNBwdChunk {
  const float* dwbar[MAXT];
  const float* w[MAXT];
  const float* a_k[MAXT];
  const float* b_co[MAXT];
  const float* sigma[MAXT];
  float* dw[MAXT];
  float* part[MAXT];    // dot partials, one per block of this item
  int64_t n[MAXT];
  int Co[MAXT];
  int blk[MAXT + 1];
  int cnt;
};

__global__ __launch_bounds__(256) void sn_bwd_dot_multi_kernel(SNBwdChunk c) {
  __shared__ float sm4[4];
  const int it = find_item(c.blk, c.cnt, blockIdx.x);
  const int b = blockIdx.x - c.blk[it];
  const int64_t base = (int64_t)b * EW_CHUNK, end = min(c.n[it], base + EW_CHUNK);
  const float* __restrict__ x = c.dwbar[it];
  const float* __restrict__ w = c.w[it];
  float s = 0.f;
  if (aligned16(x) && aligned16(w)) {
    const int64_t vend = base + ((end - base) & ~3ll);
#pragma unroll 4
    for (int64_t i = base + threadIdx.x * 4; i < vend; i += 1024) {
      const float4 a = *reinterpret_cast<const float4*>(x + i);
      const float4 q = *reinterpret_cast<const float4*>(w + i);
      s += (a.x * q.x + a.y * q.y) + (a.z * q.z + a.w * q.w);
    }
    for (int64_t i = vend + threadIdx.x; i < end; i += 256) s += x[i] * w[i];
  } else {
    for (int64_t i = base + threadIdx.x; i < end; i += 256) s += x[i] * w[i];
  }
  s = block_sum_256(s, sm4);
  if (threadIdx.x == 0) c.part[it][b] = s;
}

__global__ __launch_bounds__(256) void sn_bwd_apply_multi_kernel(SNBwdChunk c) {
  __shared__ float sm4[4];
  const int it = find_item(c.blk, c.cnt, blockIdx.x);
  const int b = blockIdx.x - c.blk[it];
  const int nb = c.blk[it + 1] - c.blk[it];
  const float inv = 1.f / c.sigma[it][0];
  // <dwbar, w>: every block sums the item's partials in the same fixed order (deterministic)
  float dot = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) dot += c.part[it][i];
  dot = block_sum_256(dot, sm4);
  const float coef = dot * inv;   // <dwbar, w_bar>
  const int Co = c.Co[it];
  const int64_t base = (int64_t)b * EW_CHUNK, end = min(c.n[it], base + EW_CHUNK);
  const float* __restrict__ x = c.dwbar[it];
  const float* __restrict__ ak = c.a_k[it];
  const float* __restrict__ bc = c.b_co[it];
  float* __restrict__ dw = c.dw[it];
  if ((Co & 3) == 0 && aligned16(x) && aligned16(dw) && aligned16(bc)) {
    // four consecutive elements share their row k (Co % 4 == 0; n = K * Co)
#pragma unroll 4
    for (int64_t i = base + threadIdx.x * 4; i < end; i += 1024) {
      const int64_t k = i / Co;
      const int co = (int)(i - k * Co);
      const float4 a = *reinterpret_cast<const float4*>(x + i);
      const float4 q = *reinterpret_cast<const float4*>(bc + co);
      const float ck = coef * ak[k];
      float4 r;
      r.x = (a.x - ck * q.x) * inv;
      r.y = (a.y - ck * q.y) * inv;
      r.z = (a.z - ck * q.z) * inv;
      r.w = (a.w - ck * q.w) * inv;
      *reinterpret_cast<float4*>(dw + i) = r;
    }
    return;
  }
  for (int64_t i = base + threadIdx.x; i < end; i += 256) {
    const int64_t k = i / Co;
    const int co = (int)(i - k * Co);
    dw[i] = (x[i] - coef * ak[k] * bc[co]) * inv;
  }
}

struct PrepChunk {
  const float* w[MAXT];
  bf16_t* bt_fwd[MAXT];   // [Co][Kp] or NULL
  bf16_t* bt_bwd[MAXT];   // [Ci][Kbp] or NULL
  int T[MAXT], Ci[MAXT], Co[MAXT];
  int blk_f[MAXT + 1];    // 32x32 transpose tiles of the forward image
  int blk_b[MAXT + 1];    // EW_CHUNK element blocks of the backward image
  int cnt;
};

__global__ __launch_bounds__(256) void prep_fwd_multi_kernel(PrepChunk c) {
  __shared__ float tile[32][33];
  const int it = find_item(c.blk_f, c.cnt, blockIdx.x);
  const int b = blockIdx.x - c.blk_f[it];
  const int Co = c.Co[it], K = c.T[it] * c.Ci[it], Kp = (K + 7) & ~7;
  const int kt = (Kp + 31) / 32;
  const int kb = (b % kt) * 32, cb = (b / kt) * 32;
  const float* __restrict__ w = c.w[it];
  bf16_t* __restrict__ bt = c.bt_fwd[it];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int k = kb + r, co = cb + tx;
    tile[r][tx] = (k < K && co < Co) ? w[(int64_t)k * Co + co] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int co = cb + r, k = kb + tx;
    if (co < Co && k < Kp) bt[(int64_t)co * Kp + k] = f2bf(tile[tx][r]);
  }
}

// backward image bt[ci][(T-1-t) * Co + co] = w[t][ci][co] (row pitch Kbp): the rows of the image are
// runs of Co consecutive source elements, so a thread converts 8 consecutive outputs from two
// 16-byte loads and stores them as one (Co % 8 == 0: the 8 share their tap; one division per 8)
__global__ __launch_bounds__(256) void prep_bwd_multi_kernel(PrepChunk c) {
  const int it = find_item(c.blk_b, c.cnt, blockIdx.x);
  const int b = blockIdx.x - c.blk_b[it];
  const int T = c.T[it], Ci = c.Ci[it], Co = c.Co[it];
  const int Kbp = (T * Co + 7) & ~7;
  const int64_t total = (int64_t)Ci * Kbp;
  const int64_t base = (int64_t)b * EW_CHUNK, end = min(total, base + EW_CHUNK);
  const float* __restrict__ w = c.w[it];
  bf16_t* __restrict__ bt = c.bt_bwd[it];
  if ((Co & 7) == 0 && aligned16(w) && aligned16(bt)) {   // (then Kbp = T * Co, no padding columns)
#pragma unroll 2
    for (int64_t i = base + threadIdx.x * 8; i < end; i += 2048) {
      const int ci = (int)(i / Kbp);
      const int kk = (int)(i - (int64_t)ci * Kbp);
      const int tt = kk / Co, co = kk - tt * Co;
      const float* src = w + ((int64_t)(T - 1 - tt) * Ci + ci) * Co + co;
      const float4 lo = *reinterpret_cast<const float4*>(src);
      const float4 hi = *reinterpret_cast<const float4*>(src + 4);
      const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      *reinterpret_cast<uint4*>(bt + i) = pack8_bf16(v);
    }
    return;
  }
  for (int64_t i = base + threadIdx.x; i < end; i += 256) {
    const int ci = (int)(i / Kbp);
    const int kk = (int)(i - (int64_t)ci * Kbp);
    float v = 0.f;
    if (kk < T * Co) {
      const int tt = kk / Co, co = kk - tt * Co;
      v = w[((int64_t)(T - 1 - tt) * Ci + ci) * Co + co];
    }
    bt[i] = f2bf(v);
  }
}

struct FlatChunk {
  const float* src[MAXT];
  int64_t off[MAXT];   // element offset of the tensor inside the flat buffer
  int64_t n[MAXT];
  int blk[MAXT + 1];
  int cnt;
};

__global__ __launch_bounds__(256) void flatten_multi_kernel(FlatChunk c, float* __restrict__ flat) {
  const int it = find_item(c.blk, c.cnt, blockIdx.x);
  const int64_t base = (int64_t)(blockIdx.x - c.blk[it]) * EW_CHUNK;
  const int64_t end = min(c.n[it], base + EW_CHUNK);
  const float* __restrict__ s = c.src[it];
  float* __restrict__ d = flat + c.off[it];
  if (aligned16(s) && aligned16(d)) {
    const int64_t vend = base + ((end - base) & ~3ll);
#pragma unroll 4
    for (int64_t i = base + threadIdx.x * 4; i < vend; i += 1024)
      *reinterpret_cast<float4*>(d + i) = *reinterpret_cast<const float4*>(s + i);
    for (int64_t i = vend + threadIdx.x; i < end; i += 256) d[i] = s[i];
    return;
  }
  for (int64_t i = base + threadIdx.x; i < end; i += 256) d[i] = s[i];
}

inline int sn_splits(int K, int Co) {
  const int ct = cdiv(Co, 64);
  int s = cdiv(128, ct);
  const int maxs = K / 64 > 0 ? K / 64 : 1;
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  return s;
}

}  // namespace

extern "C" size_t cg_spectral_norm_multi_workspace_floats(int K, int Co) {
  if (K <= 0 || Co <= 0) return 0;
  return (size_t)sn_splits(K, Co) * Co + (size_t)K;
}

extern "C" int cg_spectral_norm_multi(const cgSNItem* items, int n, float eps, cgStream stream) {
  if (!items || n <= 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_spectral_norm_multi: bad argument");
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < n; i0 += MAXT) {
    const int cnt = (n - i0) < MAXT ? (n - i0) : MAXT;
    SNChunk c;
    ScaleChunk sc;
    c.n = cnt;
    sc.cnt = 0;
    int blkA = 0, blkB = 0, blkS = 0;
    int prefA[MAXT + 1], prefB[MAXT + 1];
    for (int i = 0; i < cnt; ++i) {
      const cgSNItem& t = items[i0 + i];
      if (!t.w || !t.u || !t.u_out || !t.v_out || !t.sigma || !t.ws || t.K <= 0 || t.Co <= 0 ||
          (t.mode != 0 && t.mode != 1))
        CG_FAIL(CG_ERR_BAD_ARG, "cg_spectral_norm_multi: bad item %d", i0 + i);
      c.w[i] = t.w; c.u[i] = t.u; c.u_out[i] = t.u_out; c.v_out[i] = t.v_out;
      c.sigma[i] = t.sigma; c.part[i] = t.ws;
      c.K[i] = t.K; c.Co[i] = t.Co; c.mode[i] = t.mode;
      c.splits[i] = sn_splits(t.K, t.Co);
      const int col_blocks = cdiv(t.Co, 64) * c.splits[i], row_blocks = cdiv(t.K, 4);
      prefA[i] = blkA; prefB[i] = blkB;
      blkA += (t.mode == 0) ? col_blocks : row_blocks;
      blkB += (t.mode == 0) ? row_blocks : col_blocks;
      if (t.wbar) {
        const int j = sc.cnt++;
        sc.w[j] = t.w; sc.sigma[j] = t.sigma; sc.out[j] = t.wbar;
        sc.n[j] = (int64_t)t.K * t.Co;
        sc.blk[j] = blkS;
        blkS += cdiv(sc.n[j], EW_CHUNK);
      }
    }
    prefA[cnt] = blkA; prefB[cnt] = blkB;
    sc.blk[sc.cnt] = blkS;
    for (int i = 0; i <= cnt; ++i) c.blk[i] = prefA[i];
    sn_matvec_kernel<<<blkA, 256, 0, st>>>(c, 0);
    sn_finalize_kernel<<<cnt, 1024, 0, st>>>(c, 0, eps);
    for (int i = 0; i <= cnt; ++i) c.blk[i] = prefB[i];
    sn_matvec_kernel<<<blkB, 256, 0, st>>>(c, 1);
    sn_finalize_kernel<<<cnt, 1024, 0, st>>>(c, 1, eps);
    if (sc.cnt > 0) scale_multi_kernel<<<blkS, 256, 0, st>>>(sc);
    CG_CHECK_LAUNCH("cg_spectral_norm_multi");
  }
  return CG_OK;
}

extern "C" size_t cg_sn_backward_multi_workspace_floats(int K, int Co) {
  if (K <= 0 || Co <= 0) return 0;
  return (size_t)cdiv((int64_t)K * Co, EW_CHUNK);
}

extern "C" int cg_sn_backward_multi(const cgSNBwdItem* items, int n, cgStream stream) {
  if (!items || n <= 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_sn_backward_multi: bad argument");
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < n; i0 += MAXT) {
    const int cnt = (n - i0) < MAXT ? (n - i0) : MAXT;
    SNBwdChunk c;
    c.cnt = cnt;
    int blk = 0;
    for (int i = 0; i < cnt; ++i) {
      const cgSNBwdItem& t = items[i0 + i];
      if (!t.dwbar || !t.w || !t.a_k || !t.b_co || !t.sigma || !t.dw || !t.ws || t.K <= 0 ||
          t.Co <= 0)
        CG_FAIL(CG_ERR_BAD_ARG, "cg_sn_backward_multi: bad item %d", i0 + i);
      c.dwbar[i] = t.dwbar; c.w[i] = t.w; c.a_k[i] = t.a_k; c.b_co[i] = t.b_co;
      c.sigma[i] = t.sigma; c.dw[i] = t.dw; c.part[i] = t.ws;
      c.n[i] = (int64_t)t.K * t.Co; c.Co[i] = t.Co;
      c.blk[i] = blk;
      blk += cdiv(c.n[i], EW_CHUNK);
    }
    c.blk[cnt] = blk;
    sn_bwd_dot_multi_kernel<<<blk, 256, 0, st>>>(c);
    sn_bwd_apply_multi_kernel<<<blk, 256, 0, st>>>(c);
    CG_CHECK_LAUNCH("cg_sn_backward_multi");
  }
  return CG_OK;
}

extern "C" int cg_weight_prep_multi(const cgPrepItem* items, int n, cgStream stream) {
  if (!items || n <= 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_weight_prep_multi: bad argument");
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < n; i0 += MAXT) {
    const int cnt = (n - i0) < MAXT ? (n - i0) : MAXT;
    PrepChunk c;
    c.cnt = cnt;
    int bf = 0, bb = 0;
    for (int i = 0; i < cnt; ++i) {
      const cgPrepItem& t = items[i0 + i];
      if (!t.w || t.T <= 0 || t.Ci <= 0 || t.Co <= 0)
        CG_FAIL(CG_ERR_BAD_ARG, "cg_weight_prep_multi: bad item %d", i0 + i);
      c.w[i] = t.w; c.bt_fwd[i] = (bf16_t*)t.bt_fwd; c.bt_bwd[i] = (bf16_t*)t.bt_bwd;
      c.T[i] = t.T; c.Ci[i] = t.Ci; c.Co[i] = t.Co;
      c.blk_f[i] = bf; c.blk_b[i] = bb;
      if (t.bt_fwd) bf += cdiv((t.T * t.Ci + 7) & ~7, 32) * cdiv(t.Co, 32);
      if (t.bt_bwd) bb += cdiv((int64_t)t.Ci * ((t.T * t.Co + 7) & ~7), EW_CHUNK);
    }
    c.blk_f[cnt] = bf; c.blk_b[cnt] = bb;
    if (bf > 0) prep_fwd_multi_kernel<<<bf, 256, 0, st>>>(c);
    if (bb > 0) prep_bwd_multi_kernel<<<bb, 256, 0, st>>>(c);
    CG_CHECK_LAUNCH("cg_weight_prep_multi");
  }
  return CG_OK;
}

extern "C" int cg_flatten_multi(const float* const* srcs, const int64_t* sizes, int n, float* flat,
                                cgStream stream) {
  if (!srcs || !sizes || !flat || n <= 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_flatten_multi: bad argument");
  hipStream_t st = (hipStream_t)stream;
  int64_t off = 0;
  for (int i0 = 0; i0 < n; i0 += MAXT) {
    const int cnt = (n - i0) < MAXT ? (n - i0) : MAXT;
    FlatChunk c;
    c.cnt = cnt;
    int blk = 0;
    for (int i = 0; i < cnt; ++i) {
      if (!srcs[i0 + i] || sizes[i0 + i] <= 0)
        CG_FAIL(CG_ERR_BAD_ARG, "cg_flatten_multi: bad item %d", i0 + i);
      c.src[i] = srcs[i0 + i];
      c.off[i] = off;
      c.n[i] = sizes[i0 + i];
      off += sizes[i0 + i];
      c.blk[i] = blk;
      blk += cdiv(c.n[i], EW_CHUNK);
    }
    c.blk[cnt] = blk;
    flatten_multi_kernel<<<blk, 256, 0, st>>>(c, flat);
    CG_CHECK_LAUNCH("cg_flatten_multi");
  }
  return CG_OK;
}
