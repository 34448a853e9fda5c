// Generalised NHWC convolution on MFMA (gfx950): forward / data-gradient / transposed conv /
// linear all map onto ONE im2col-free implicit-GEMM gather kernel (cg_gconv) and one
// weight-gradient kernel (cg_gwgrad).  See include/cgamd.h for the contract and the reference
// call sites (architectures/arch_ops.py:538-592, architectures/resnet_ops.py:35-56,112-134).
//
// Tiling (v1): 256 threads = 4 waves; block tile BM x BN output, BK = 64 reduction slice staged
// through LDS (register-staged double buffer, one barrier per slice); each wave owns a
// (BM/WM) x (BN/WN) sub-tile built from v_mfma_f32_32x32x16_bf16.  LDS rows are padded by 16 B
// (144 B stride) which makes every ds_read_b128 fragment read bank-conflict free.
#include "cg_common.h"
#include "cg_conv_fast.h"

namespace {

struct GConvArgs {
  const bf16_t* in;
  const bf16_t* bt;
  void* out;
  const float* bias;
  const bf16_t* gate_in;
  const bf16_t* gate_out;
  const bf16_t* residual;
  int N, Hin, Win, Ci, Ho, Wo, Co, kh, kw, S, ulog, pt, pl;
  int M;    // N*Ho*Wo
  int K;    // kh*kw*Ci
  int Kp;   // K rounded up to 8 (row stride of bt)
  int HinU, WinU;
  float slope_in, slope_out;
  int out_f32;
  int self_gate;  // gate_out == out
  FastDiv dWo, dHo, dCi, dKw;
};

constexpr int LDS_PAD = 8;  // bf16 elements (16 B)

__device__ __forceinline__ uint4 gate_apply(uint4 x, uint4 g, float slope) {
  union { uint4 q; bf16_t h[8]; } xv, gv, ov;
  xv.q = x;
  gv.q = g;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float gf = bf2f(gv.h[e]);
    const float xf = bf2f(xv.h[e]);
    ov.h[e] = gf > 0.f ? xv.h[e] : f2bf(xf * slope);
  }
  return ov.q;
}

// Loads 8 consecutive k (one tap, 8 channels) of im2col row described by (nbase, bh, bw).
template <bool VEC>
__device__ __forceinline__ uint4 load_a_chunk(const GConvArgs& a, bool row_ok, int nbase, int bh,
                                              int bw, int k) {
  uint4 z = make_uint4(0, 0, 0, 0);
  if (!row_ok) return z;
  if (VEC) {
    if (k >= a.K) return z;
    const uint32_t tap = fdiv((uint32_t)k, a.dCi);
    const int c = k - (int)tap * a.Ci;
    const int r = (int)fdiv(tap, a.dKw);
    const int s = (int)tap - r * a.kw;
    const int ihv = bh + r, iwv = bw + s;
    if (ihv < 0 || iwv < 0 || ihv >= a.HinU || iwv >= a.WinU) return z;
    const int um = (1 << a.ulog) - 1;
    if ((ihv & um) | (iwv & um)) return z;
    const int64_t off = ((int64_t)(nbase + (ihv >> a.ulog)) * a.Win + (iwv >> a.ulog)) * a.Ci + c;
    uint4 x = *reinterpret_cast<const uint4*>(a.in + off);
    if (a.gate_in) {
      uint4 g = (a.gate_in == a.in) ? x : *reinterpret_cast<const uint4*>(a.gate_in + off);
      x = gate_apply(x, g, a.slope_in);
    }
    return x;
  } else {
    union { uint4 q; bf16_t h[8]; } ov;
    ov.q = z;
    const int um = (1 << a.ulog) - 1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ke = k + e;
      if (ke >= a.K) continue;
      const uint32_t tap = fdiv((uint32_t)ke, a.dCi);
      const int c = ke - (int)tap * a.Ci;
      const int r = (int)fdiv(tap, a.dKw);
      const int s = (int)tap - r * a.kw;
      const int ihv = bh + r, iwv = bw + s;
      if (ihv < 0 || iwv < 0 || ihv >= a.HinU || iwv >= a.WinU) continue;
      if ((ihv & um) | (iwv & um)) continue;
      const int64_t off =
          ((int64_t)(nbase + (ihv >> a.ulog)) * a.Win + (iwv >> a.ulog)) * a.Ci + c;
      bf16_t x = a.in[off];
      if (a.gate_in) {
        const float g = bf2f(a.gate_in[off]);
        if (!(g > 0.f)) x = f2bf(bf2f(x) * a.slope_in);
      }
      ov.h[e] = x;
    }
    return ov.q;
  }
}

template <int BM, int BN, int BK, int WM, int WN, bool VEC>
__global__ __launch_bounds__(256) void gconv_kernel(GConvArgs a) {
  constexpr int LD = BK + LDS_PAD;
  constexpr int CPR = BK / 8;               // 16-byte chunks per tile row
  constexpr int LA = BM * CPR / 256;        // A chunks per thread
  constexpr int LB = (BN * CPR + 255) / 256;  // B chunks per thread
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(BM * CPR % 256 == 0, "A tile must divide over the block");

  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * (BM + BN) * LD];
  auto As = [&](int buf) { return smem + buf * (BM + BN) * LD; };
  auto Bs = [&](int buf) { return smem + buf * (BM + BN) * LD + BM * LD; };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // ---- per-thread im2col row descriptors (fixed across the K loop) ----
  const int seg = tid % CPR;
  int a_nbase[LA], a_bh[LA], a_bw[LA];
  bool a_ok[LA];
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    const int row = tid / CPR + i * (256 / CPR);
    const int m = m0 + row;
    a_ok[i] = m < a.M;
    const uint32_t mm = a_ok[i] ? (uint32_t)m : 0u;
    const uint32_t t1 = fdiv(mm, a.dWo);
    const int ow = (int)(mm - t1 * a.Wo);
    const uint32_t n = fdiv(t1, a.dHo);
    const int oh = (int)(t1 - n * a.Ho);
    a_nbase[i] = (int)n * a.Hin;
    a_bh[i] = oh * a.S - a.pt;
    a_bw[i] = ow * a.S - a.pl;
  }
  int b_row[LB];
  bool b_ok[LB];
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    const int cidx = tid + i * 256;
    const int row = cidx / CPR;
    b_row[i] = row;
    b_ok[i] = (row < BN) && (n0 + row < a.Co);
  }

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  const int nk = (a.K + BK - 1) / BK;
  uint4 ra[LA], rb[LB];

  auto gload = [&](int it) {
    const int k = it * BK + seg * 8;
#pragma unroll
    for (int i = 0; i < LA; ++i)
      ra[i] = load_a_chunk<VEC>(a, a_ok[i], a_nbase[i], a_bh[i], a_bw[i], k);
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      rb[i] = make_uint4(0, 0, 0, 0);
      if (b_ok[i] && k < a.Kp)
        rb[i] = *reinterpret_cast<const uint4*>(a.bt + (int64_t)(n0 + b_row[i]) * a.Kp + k);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int row = tid / CPR + i * (256 / CPR);
      *reinterpret_cast<uint4*>(As(buf) + row * LD + seg * 8) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < LB; ++i)
      if (b_row[i] < BN) *reinterpret_cast<uint4*>(Bs(buf) + b_row[i] * LD + seg * 8) = rb[i];
  };

  gload(0);
  sstore(0);
  __syncthreads();

  const int arow0 = wm * (BM / WM) + (lane & 31);
  const int brow0 = wn * (BN / WN) + (lane & 31);
  const int koff = (lane >> 5) * 8;

  for (int it = 0; it < nk; ++it) {
    const int buf = it & 1;
    if (it + 1 < nk) gload(it + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8_t af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const bf16x8_t*>(As(buf) + (arow0 + i * 32) * LD + kk * 16 + koff);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8_t*>(Bs(buf) + (brow0 + j * 32) * LD + kk * 16 + koff);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (it + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias, output gate, residual, store ----
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int co = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
      if (co >= a.Co) continue;
      const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int row = wm * (BM / WM) + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
        const int m = m0 + row;
        if (m >= a.M) continue;
        const int64_t o = (int64_t)m * a.Co + co;
        float val = acc[i][j][v] + bias;
        if (a.self_gate) {
          if (!(val > 0.f)) val *= a.slope_out;
        } else if (a.gate_out) {
          const float g = bf2f(a.gate_out[o]);
          if (!(g > 0.f)) val *= a.slope_out;
        }
        if (a.residual) val += bf2f(a.residual[o]);
        if (a.out_f32)
          reinterpret_cast<float*>(a.out)[o] = val;
        else
          reinterpret_cast<bf16_t*>(a.out)[o] = f2bf(val);
      }
    }
  }
}

// -------------------------------------------------------------------------------------------
// VALU kernels for the 1x1 problems no MFMA tile fits (profiles/r03_dstep_launches.txt,
// r03_biggan_launches.txt: the generic kernel spent 35-210 us on each of them):
//  * "thin": at most 8 input channels (K <= 8) -- the data gradient of a discriminator's final
//    linear layer (dy [B, 1] x W^T, arch_ops.py:538-556 under tf.gradients), the 1x1 shortcut of
//    BigGAN's first discriminator block on the pooled RGB image (resnet_biggan.py:285-300);
//  * "small linear": linear layers on at most 512 rows whose K is not a multiple of 8 -- the
//    conditional-batch-norm projections of [z chunk, label embedding] (148 -> C, arch_ops.py:
//    362-372) and the label embedding itself (1000 -> 128).
// A thread owns 8 consecutive output channels; thin: of THIN_PIX pixels, with its 8 x K weights in
// registers; small linear: of one row, lanes of a wave share the channel group (weight loads are
// wave-uniform) and differ in the row.
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ void vconv_bias8(const GConvArgs& a, int co, float* bv) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float b = a.bias ? a.bias[min(co + e, a.Co - 1)] : 0.f;   // clamped, not guarded
    bv[e] = co + e < a.Co ? b : 0.f;
  }
}
__device__ __forceinline__ void vconv_store8(const GConvArgs& a, int64_t m, int co, float* val,
                                             const float* bv) {
  const int64_t o = m * a.Co + co;
  if (co + 8 <= a.Co && (a.Co & 7) == 0) {
    // whole group: gate / residual as one 16-byte load each, no per-element branches
    float gv[8], rv[8];
    if (a.gate_out) unpack8_bf16(*reinterpret_cast<const uint4*>(a.gate_out + o), gv);
    if (a.residual) unpack8_bf16(*reinterpret_cast<const uint4*>(a.residual + o), rv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = val[e] + bv[e];
      if (a.self_gate) v = v > 0.f ? v : v * a.slope_out;
      else if (a.gate_out) v = gv[e] > 0.f ? v : v * a.slope_out;
      if (a.residual) v += rv[e];
      val[e] = v;
    }
    if (a.out_f32) {
      float* op = reinterpret_cast<float*>(a.out) + o;
      *reinterpret_cast<float4*>(op) = make_float4(val[0], val[1], val[2], val[3]);
      *reinterpret_cast<float4*>(op + 4) = make_float4(val[4], val[5], val[6], val[7]);
    } else {
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.out) + o) = pack8_bf16(val);
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (co + e >= a.Co) continue;
    val[e] += bv[e];
    if (a.self_gate) {
      if (!(val[e] > 0.f)) val[e] *= a.slope_out;
    } else if (a.gate_out) {
      if (!(bf2f(a.gate_out[o + e]) > 0.f)) val[e] *= a.slope_out;
    }
    if (a.residual) val[e] += bf2f(a.residual[o + e]);
    if (a.out_f32) reinterpret_cast<float*>(a.out)[o + e] = val[e];
    else reinterpret_cast<bf16_t*>(a.out)[o + e] = f2bf(val[e]);
  }
}
// input element `off` (always a valid offset: callers clamp and select afterwards -- a guarded load
// compiles to a branch with a wait on the value behind it, i.e. one serial round trip per element)
__device__ __forceinline__ float vconv_in(const GConvArgs& a, int64_t off) {
  bf16_t v = a.in[off];
  if (a.gate_in) {   // kernel-uniform
    const float g = bf2f(a.gate_in[off]);
    v = g > 0.f ? v : f2bf(bf2f(v) * a.slope_in);   // rounded like the staged operand
  }
  return bf2f(v);
}

constexpr int THIN_PIX = 4;
// block = gpb channel groups x ppb pixels (gpb * ppb <= 256): thread -> (pixel slot t / gpb, group
// blockIdx.y * gpb + t % gpb); iteration i covers the ppb consecutive pixels blockIdx.x * THIN_PIX *
// ppb + i * ppb ..: the stores of a block iteration are one contiguous span of pixel rows
// `iters` spans of THIN_PIX * ppb pixels per block: the weights of a thread (8 x 16 bytes, 64 conversions)
// are set up once per block, which at one span cost as much as the span itself
__global__ __launch_bounds__(256) void thin_conv_kernel(GConvArgs a, int groups, int gpb, int ppb,
                                                        int iters) {
  const int gl = threadIdx.x % gpb, ps = threadIdx.x / gpb;
  const int g = blockIdx.y * gpb + gl;
  if (g >= groups || ps >= ppb) return;
  const int co = g * 8;
  float w[8][8], bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {   // clamped row, zeroed afterwards
    unpack8_bf16(*reinterpret_cast<const uint4*>(a.bt + (int64_t)min(co + e, a.Co - 1) * a.Kp), w[e]);
#pragma unroll
    for (int k = 0; k < 8; ++k) w[e][k] = co + e < a.Co ? w[e][k] : 0.f;
  }
  vconv_bias8(a, co, bv);
  for (int it = 0; it < iters; ++it) {
  const int64_t m0 = ((int64_t)blockIdx.x * iters + it) * THIN_PIX * ppb + ps;
  if (m0 >= a.M) break;
  // all inputs of the thread's pixels first: unconditional loads from clamped offsets, selected
  // afterwards (the weights of the padding k are zero anyway)
  float x[THIN_PIX][8];
  if (a.Ci == 3 && !a.gate_in) {   // kernel-uniform
    // RGB pixels (the 1x1 shortcut of BigGAN's first discriminator block, resnet_biggan.py:372-377): the
    // 6 bytes of a pixel out of the two aligned dwords that cover them instead of eight 2-byte loads
    // (342 us for a 402 MB output at batch 512: 1.2 TB/s); the tensor's very last pixel keeps the
    // element-wise loads (its second dword would end 2 bytes behind the tensor)
    const uint32_t* in32 = reinterpret_cast<const uint32_t*>(a.in);
#pragma unroll
    for (int i = 0; i < THIN_PIX; ++i) {
      const int64_t m = min(m0 + (int64_t)i * ppb, (int64_t)a.M - 2);
      const int64_t mm = m < 0 ? 0 : m;
      const int64_t d0 = (mm * 6) >> 2;
      const uint32_t lo = in32[d0], hi = in32[d0 + 1];
      const bool odd = ((mm * 6) & 2) != 0;   // pixel starts at the upper half of `lo`
      const uint32_t c0 = odd ? lo >> 16 : lo & 0xffffu;
      const uint32_t c1 = odd ? hi & 0xffffu : lo >> 16;
      const uint32_t c2 = odd ? hi >> 16 : hi & 0xffffu;
      x[i][0] = __uint_as_float(c0 << 16);
      x[i][1] = __uint_as_float(c1 << 16);
      x[i][2] = __uint_as_float(c2 << 16);
#pragma unroll
      for (int k = 3; k < 8; ++k) x[i][k] = 0.f;
    }
    if (m0 + (int64_t)(THIN_PIX - 1) * ppb >= (int64_t)a.M - 1 || a.M < 2) {
      // a block that touches the last pixel: element-wise for exactly that pixel
#pragma unroll
      for (int i = 0; i < THIN_PIX; ++i) {
        if (m0 + (int64_t)i * ppb == (int64_t)a.M - 1) {
#pragma unroll
          for (int k = 0; k < 3; ++k) x[i][k] = bf2f(a.in[((int64_t)a.M - 1) * 3 + k]);
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < THIN_PIX; ++i) {
      const int64_t m = min(m0 + (int64_t)i * ppb, (int64_t)a.M - 1);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float v = vconv_in(a, m * a.Ci + min(k, a.Ci - 1));
        x[i][k] = k < a.Ci ? v : 0.f;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < THIN_PIX; ++i) {
    const int64_t m = m0 + (int64_t)i * ppb;
    if (m >= a.M) continue;
    float val[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      val[e] = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) val[e] += x[i][k] * w[e][k];   // Kp == 8, padding is zero
    }
    vconv_store8(a, m, co, val, bv);
  }
  }
}

// Linear layers with very few outputs and a long K (the final `linear(h, 1)` of the DCGAN / SNDCGAN
// discriminators on a flattened 16 x 16 x 512 map: dcgan.py:125-128, sndcgan.py:123-126 -- K =
// 131,072, Co = 1): one 1024-thread block per (row, output) walks K with 16-byte loads and reduces
// in a fixed order (deterministic).  The MFMA tiles spend a 32-wide tile on one useful column and a
// single workgroup's K loop of 2,048 slices: 4.6 ms per call at batch 64; this is one pass over
// x (16.8 MB) -- HBM / latency bound, a few microseconds.
__global__ __launch_bounds__(1024) void rowdot_linear_kernel(GConvArgs a) {
  __shared__ float part[16];
  const int m = blockIdx.x, co = blockIdx.y;
  const bf16_t* __restrict__ x = a.in + (int64_t)m * a.Ci;
  const bf16_t* __restrict__ gx = a.gate_in ? a.gate_in + (int64_t)m * a.Ci : nullptr;
  const bf16_t* __restrict__ w = a.bt + (int64_t)co * a.Kp;
  float acc = 0.f;
  for (int k0 = threadIdx.x * 8; k0 < a.Ci; k0 += 1024 * 8) {   // Ci % 8 == 0
    float xv[8], wv[8];
    unpack8_bf16(*reinterpret_cast<const uint4*>(x + k0), xv);
    unpack8_bf16(*reinterpret_cast<const uint4*>(w + k0), wv);
    if (gx) {   // kernel-uniform; the gated operand is rounded to bf16 like the staged one (vconv_in)
      float gv[8];
      unpack8_bf16(*reinterpret_cast<const uint4*>(gx + k0), gv);
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = gv[e] > 0.f ? xv[e] : bf2f(f2bf(xv[e] * a.slope_in));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += xv[e] * wv[e];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v += part[i];
    if (a.bias) v += a.bias[co];
    const int64_t o = (int64_t)m * a.Co + co;
    if (a.self_gate) v = v > 0.f ? v : v * a.slope_out;
    else if (a.gate_out) v = bf2f(a.gate_out[o]) > 0.f ? v : v * a.slope_out;
    if (a.residual) v += bf2f(a.residual[o]);
    if (a.out_f32) reinterpret_cast<float*>(a.out)[o] = v;
    else reinterpret_cast<bf16_t*>(a.out)[o] = f2bf(v);
  }
}

// block = 64 rows x ONE 8-channel group; wave w takes the 8-element K pieces w, w + 4, ... (the
// weight loads of a wave are uniform), lane -> row; the four partial sums meet in LDS.  The 8 weight
// rows of a group are read once per 64 rows (a wave per output re-read them per row: 100 MB of L2
// traffic for a 0.5 MB matrix).  Round 6 tried the x tile staged in LDS with coalesced loads, four
// channel groups per block, one wave per group over the whole K: 178.1 / 178.5 -> 179.5 / 179.5 ms on the
// BigGAN bs-256 step (a quarter of the blocks, a serial K loop per wave) -- not kept.
__global__ __launch_bounds__(256) void small_linear_kernel(GConvArgs a) {
  __shared__ float part[4][64][9];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * 64 + lane;
  const int co = blockIdx.y * 8;
  const bool mok = m < a.M;
  float val[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) val[e] = 0.f;
  const int64_t xrow = (int64_t)(mok ? m : 0) * a.Ci;
#pragma unroll 2
  for (int k0 = wave * 8; k0 < a.Ci; k0 += 32) {
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {   // clamped offsets, selected afterwards (see vconv_in)
      const float v = vconv_in(a, xrow + min(k0 + k, a.Ci - 1));
      x[k] = (mok && k0 + k < a.Ci) ? v : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float w[8];
      unpack8_bf16(*reinterpret_cast<const uint4*>(a.bt + (int64_t)min(co + e, a.Co - 1) * a.Kp + k0), w);
#pragma unroll
      for (int k = 0; k < 8; ++k) val[e] += x[k] * w[k];   // bt rows are zero-padded to Kp
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[wave][lane][e] = val[e];
  __syncthreads();
  if (wave == 0 && mok) {
    float bv[8];
    vconv_bias8(a, co, bv);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      val[e] = (part[0][lane][e] + part[1][lane][e]) + (part[2][lane][e] + part[3][lane][e]);
    vconv_store8(a, m, co, val, bv);
  }
}

// -------------------------------------------------------------------------------------------
// Weight gradient.
// -------------------------------------------------------------------------------------------
struct GWgradArgs {
  const bf16_t* in;
  const bf16_t* gate_in;
  const bf16_t* dy;
  const bf16_t* gate_dy;
  float* out;       // dw (splits == 1) or workspace partials [splits][K*Co]
  float* bias_out;  // NULL, dbias (splits == 1) or workspace partials [splits][Co]
  int N, Hin, Win, Ci, Ho, Wo, Co, kh, kw, S, ulog, pt, pl;
  int M, K, HinU, WinU;
  int rows_per_split;  // multiple of 32
  int accumulate;      // only honoured when writing dw directly
  float slope_in, slope_dy;
  FastDiv dWo, dHo, dCi, dKw;
};

template <int TK, int TN, bool VX, bool VY>
__global__ __launch_bounds__(256) void gwgrad_kernel(GWgradArgs a) {
  constexpr int BMR = 32;
  constexpr int LDX = TK + LDS_PAD;
  constexpr int LDY = TN + LDS_PAD;
  constexpr int XC = TK / 8, YC = TN / 8;  // chunks per row
  constexpr int LX = (BMR * XC + 255) / 256;
  constexpr int LY = (BMR * YC + 255) / 256;
  constexpr int WK = TK >= 64 ? 2 : 1;  // wave grid
  constexpr int WN = 4 / WK;
  constexpr int FK = TK / WK / 32;  // 32x32 tiles per wave along k
  constexpr int FN = TN / WN / 32;
  static_assert(FK >= 1 && FN >= 1, "tile too small");

  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * BMR * (LDX + LDY)];
  auto Xs = [&](int buf) { return smem + buf * BMR * (LDX + LDY); };
  auto Ys = [&](int buf) { return smem + buf * BMR * (LDX + LDY) + BMR * LDX; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / WN, wn = wave % WN;
  const int k0 = blockIdx.x * TK;
  const int c0 = blockIdx.y * TN;
  const int mbeg = blockIdx.z * a.rows_per_split;
  const int mend = min(a.M, mbeg + a.rows_per_split);
  const int nit = (mend - mbeg + BMR - 1) / BMR;
  const int um = (1 << a.ulog) - 1;

  // per-thread fixed k decode for the X tile loads
  int x_row[LX], x_c[LX], x_r[LX], x_s[LX];
  bool x_kok[LX];
  constexpr bool vec = VX;
#pragma unroll
  for (int i = 0; i < LX; ++i) {
    const int cidx = tid + i * 256;
    x_row[i] = cidx / XC;
    const int k = k0 + (cidx % XC) * 8;
    x_kok[i] = (x_row[i] < BMR) && (k < a.K);
    const uint32_t tap = fdiv((uint32_t)(x_kok[i] ? k : 0), a.dCi);
    x_c[i] = k - (int)tap * a.Ci;
    x_r[i] = (int)fdiv(tap, a.dKw);
    x_s[i] = (int)tap - x_r[i] * a.kw;
  }
  int y_row[LY], y_col[LY];
  bool y_ok[LY];
#pragma unroll
  for (int i = 0; i < LY; ++i) {
    const int cidx = tid + i * 256;
    y_row[i] = cidx / YC;
    y_col[i] = c0 + (cidx % YC) * 8;
    y_ok[i] = (y_row[i] < BMR) && (y_col[i] < a.Co);
  }
  constexpr bool yvec = VY;

  f32x16_t acc[FK][FN];
#pragma unroll
  for (int i = 0; i < FK; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
  float bias_acc = 0.f;

  uint4 rx[LX], ry[LY];
  auto gload = [&](int it) {
    const int mb = mbeg + it * BMR;
#pragma unroll
    for (int i = 0; i < LX; ++i) {
      uint4 ovq = make_uint4(0, 0, 0, 0);
      const int m = mb + x_row[i];
      if (x_kok[i] && m < mend) {
        const uint32_t t1 = fdiv((uint32_t)m, a.dWo);
        const int ow = m - (int)t1 * a.Wo;
        const uint32_t n = fdiv(t1, a.dHo);
        const int oh = (int)t1 - (int)n * a.Ho;
        const int bh = oh * a.S - a.pt, bw = ow * a.S - a.pl;
        if constexpr (vec) {
          const int ihv = bh + x_r[i], iwv = bw + x_s[i];
          if (ihv >= 0 && iwv >= 0 && ihv < a.HinU && iwv < a.WinU && !((ihv & um) | (iwv & um))) {
            const int64_t off =
                ((int64_t)((int)n * a.Hin + (ihv >> a.ulog)) * a.Win + (iwv >> a.ulog)) * a.Ci +
                x_c[i];
            ovq = *reinterpret_cast<const uint4*>(a.in + off);
            if (a.gate_in) {
              uint4 g = (a.gate_in == a.in) ? ovq
                                             : *reinterpret_cast<const uint4*>(a.gate_in + off);
              ovq = gate_apply(ovq, g, a.slope_in);
            }
          }
        } else {
          s16x8_t hv = {0, 0, 0, 0, 0, 0, 0, 0};
          const int kb = k0 + ((tid + i * 256) % XC) * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int ke = kb + e;
            if (ke >= a.K) continue;
            const uint32_t tap = fdiv((uint32_t)ke, a.dCi);
            const int c = ke - (int)tap * a.Ci;
            const int r = (int)fdiv(tap, a.dKw);
            const int s = (int)tap - r * a.kw;
            const int ihv = bh + r, iwv = bw + s;
            if (ihv < 0 || iwv < 0 || ihv >= a.HinU || iwv >= a.WinU) continue;
            if ((ihv & um) | (iwv & um)) continue;
            const int64_t off =
                ((int64_t)((int)n * a.Hin + (ihv >> a.ulog)) * a.Win + (iwv >> a.ulog)) * a.Ci + c;
            bf16_t x = a.in[off];
            if (a.gate_in) {
              const float g = bf2f(a.gate_in[off]);
              if (!(g > 0.f)) x = f2bf(bf2f(x) * a.slope_in);
            }
            hv[e] = (short)x;
          }
          ovq = __builtin_bit_cast(uint4, hv);
        }
      }
      rx[i] = ovq;
    }
#pragma unroll
    for (int i = 0; i < LY; ++i) {
      uint4 ovq = make_uint4(0, 0, 0, 0);
      const int m = mb + y_row[i];
      if (y_ok[i] && m < mend) {
        const int64_t off = (int64_t)m * a.Co + y_col[i];
        if constexpr (yvec) {
          ovq = *reinterpret_cast<const uint4*>(a.dy + off);
          if (a.gate_dy) {
            uint4 g = *reinterpret_cast<const uint4*>(a.gate_dy + off);
            ovq = gate_apply(ovq, g, a.slope_dy);
          }
        } else {
          s16x8_t hv = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (y_col[i] + e >= a.Co) continue;
            bf16_t x = a.dy[off + e];
            if (a.gate_dy) {
              const float g = bf2f(a.gate_dy[off + e]);
              if (!(g > 0.f)) x = f2bf(bf2f(x) * a.slope_dy);
            }
            hv[e] = (short)x;
          }
          ovq = __builtin_bit_cast(uint4, hv);
        }
      }
      ry[i] = ovq;
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LX; ++i)
      if (x_row[i] < BMR)
        *reinterpret_cast<uint4*>(Xs(buf) + x_row[i] * LDX + ((tid + i * 256) % XC) * 8) = rx[i];
#pragma unroll
    for (int i = 0; i < LY; ++i)
      if (y_row[i] < BMR)
        *reinterpret_cast<uint4*>(Ys(buf) + y_row[i] * LDY + ((tid + i * 256) % YC) * 8) = ry[i];
  };

  if (nit > 0) {
    gload(0);
    sstore(0);
  }
  __syncthreads();

  const int xcol0 = wk * (TK / WK) + (lane & 31);
  const int ycol0 = wn * (TN / WN) + (lane & 31);
  const int mofs = (lane >> 5) * 8;
  const bool do_bias = (a.bias_out != nullptr) && (blockIdx.x == 0);

  for (int it = 0; it < nit; ++it) {
    const int buf = it & 1;
    if (it + 1 < nit) gload(it + 1);
#pragma unroll
    for (int kk = 0; kk < BMR / 16; ++kk) {
      bf16x8_t xf[FK], yf[FN];
      const int mm = kk * 16 + mofs;
#pragma unroll
      for (int i = 0; i < FK; ++i) {
        s16x8_t u;
        const bf16_t* xp = Xs(buf) + mm * LDX + xcol0 + i * 32;
#pragma unroll
        for (int e = 0; e < 8; ++e) u[e] = (short)xp[e * LDX];
        xf[i] = __builtin_bit_cast(bf16x8_t, u);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        s16x8_t u;
        const bf16_t* yp = Ys(buf) + mm * LDY + ycol0 + j * 32;
#pragma unroll
        for (int e = 0; e < 8; ++e) u[e] = (short)yp[e * LDY];
        yf[j] = __builtin_bit_cast(bf16x8_t, u);
      }
#pragma unroll
      for (int i = 0; i < FK; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[i], yf[j], acc[i][j], 0, 0, 0);
    }
    if (do_bias && tid < TN) {
#pragma unroll 8
      for (int r = 0; r < BMR; ++r) bias_acc += bf2f(Ys(buf)[r * LDY + tid]);
    }
    if (it + 1 < nit) sstore(buf ^ 1);
    __syncthreads();
  }

  const bool direct = (gridDim.z == 1);
  float* outp = a.out + (direct ? 0 : (int64_t)blockIdx.z * a.K * a.Co);
#pragma unroll
  for (int i = 0; i < FK; ++i) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int co = c0 + wn * (TN / WN) + j * 32 + (lane & 31);
      if (co >= a.Co) continue;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int k = k0 + wk * (TK / WK) + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
        if (k >= a.K) continue;
        const int64_t o = (int64_t)k * a.Co + co;
        if (direct && a.accumulate)
          outp[o] += acc[i][j][v];
        else
          outp[o] = acc[i][j][v];
      }
    }
  }
  if (do_bias && tid < TN && c0 + tid < a.Co) {
    float* bp = a.bias_out + (direct ? 0 : (int64_t)blockIdx.z * a.Co);
    if (direct && a.accumulate)
      bp[c0 + tid] += bias_acc;
    else
      bp[c0 + tid] = bias_acc;
  }
}

__global__ void split_reduce_kernel(const float* __restrict__ part, int splits, int64_t n,
                                    float* __restrict__ out, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += part[(int64_t)z * n + i];
  out[i] = accumulate ? out[i] + s : s;
}

// -------------------------------------------------------------------------------------------
// Weight preparation (fp32 HWIO -> bf16 operand images).
// -------------------------------------------------------------------------------------------
__global__ void prep_fwd_kernel(const float* __restrict__ w, int K, int Kp, int Co,
                                const float* __restrict__ scale, bf16_t* __restrict__ bt) {
  // transpose [K][Co] -> [Co][Kp] through a 32x33 LDS tile
  __shared__ float tile[32][33];
  const float sc = scale ? *scale : 1.f;
  const int kb = blockIdx.x * 32, cb = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int k = kb + r, c = cb + tx;
    tile[r][tx] = (k < K && c < Co) ? w[(int64_t)k * Co + c] * sc : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = cb + r, k = kb + tx;
    if (c < Co && k < Kp) bt[(int64_t)c * Kp + k] = f2bf(tile[tx][r]);
  }
}

__global__ void prep_bwd_kernel(const float* __restrict__ w, int T, int Ci, int Co, int Kbp,
                                const float* __restrict__ scale, bf16_t* __restrict__ bt) {
  // out[ci][ (T-1-t)*Co + co ] = w[t][ci][co]; pad columns [T*Co, Kbp) are zero.
  const float sc = scale ? *scale : 1.f;
  const int64_t total = (int64_t)Ci * Kbp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i / Kbp);
    const int kk = (int)(i - (int64_t)ci * Kbp);
    float v = 0.f;
    if (kk < T * Co) {
      const int tt = kk / Co, co = kk - tt * Co;
      const int t = T - 1 - tt;
      v = w[((int64_t)t * Ci + ci) * Co + co] * sc;
    }
    bt[i] = f2bf(v);
  }
}

int ilog2_exact(int x) {
  int l = 0;
  while ((1 << l) < x) ++l;
  return ((1 << l) == x) ? l : -1;
}

int check_geom(const cgConvGeom* g, const char* who) {
  if (!g) CG_FAIL(CG_ERR_BAD_ARG, "%s: null geometry", who);
  if (g->N <= 0 || g->Hin <= 0 || g->Win <= 0 || g->Ci <= 0 || g->Ho <= 0 || g->Wo <= 0 ||
      g->Co <= 0 || g->kh <= 0 || g->kw <= 0 || g->S <= 0 || g->U <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "%s: non-positive dimension", who);
  if (ilog2_exact(g->U) < 0) CG_FAIL(CG_ERR_UNSUPPORTED, "%s: U=%d is not a power of two", who, g->U);
  if ((int64_t)g->N * g->Ho * g->Wo >= (1ll << 31) ||
      (int64_t)g->N * g->Hin * g->Win >= (1ll << 31) || (int64_t)g->kh * g->kw * g->Ci >= (1ll << 30))
    CG_FAIL(CG_ERR_UNSUPPORTED, "%s: index space exceeds 2^31", who);
  return CG_OK;
}

}  // namespace
void cg_conv_algorithmic_cost(const cgConvGeom* g, double* flops, double* bytes) {
  const double m = (double)g->N * g->Ho * g->Wo;
  double taps = (double)g->kh * g->kw;
  if (g->U > 1) taps /= (double)g->U * g->U;
  *flops = 2.0 * m * taps * g->Ci * g->Co;
  *bytes = 2.0 * ((double)g->N * g->Hin * g->Win * g->Ci + m * g->Co +
                  (double)g->kh * g->kw * g->Ci * g->Co);
}

extern "C" size_t cg_weight_prep_elems(int kh, int kw, int Ci, int Co, int which) {
  if (kh <= 0 || kw <= 0 || Ci <= 0 || Co <= 0) return 0;
  const int T = kh * kw;
  const int Cin = which ? Co : Ci, R = which ? Ci : Co;
  return (size_t)R * ((T * Cin + 7) & ~7);
}

extern "C" int cg_weight_prep(const float* w, int kh, int kw, int Ci, int Co, const float* scale,
                              void* bt_fwd, void* bt_bwd, cgStream stream) {
  if (!w || kh <= 0 || kw <= 0 || Ci <= 0 || Co <= 0)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_weight_prep: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int T = kh * kw;
  if (bt_fwd) {
    const int K = T * Ci, Kp = (K + 7) & ~7;
    dim3 grid(cdiv(Kp, 32), cdiv(Co, 32));
    prep_fwd_kernel<<<grid, 256, 0, st>>>(w, K, Kp, Co, scale, (bf16_t*)bt_fwd);
    CG_CHECK_LAUNCH("cg_weight_prep(fwd)");
  }
  if (bt_bwd) {
    const int Kb = T * Co, Kbp = (Kb + 7) & ~7;
    const int64_t total = (int64_t)Ci * Kbp;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    prep_bwd_kernel<<<blocks, 256, 0, st>>>(w, T, Ci, Co, Kbp, scale, (bf16_t*)bt_bwd);
    CG_CHECK_LAUNCH("cg_weight_prep(bwd)");
  }
  return CG_OK;
}

template <int BM, int BN, int WM, int WN>
static void launch_gconv(const GConvArgs& a, bool vec, hipStream_t st) {
  dim3 grid(cdiv(a.M, BM), cdiv(a.Co, BN));
  if (vec)
    gconv_kernel<BM, BN, 64, WM, WN, true><<<grid, 256, 0, st>>>(a);
  else
    gconv_kernel<BM, BN, 64, WM, WN, false><<<grid, 256, 0, st>>>(a);
}

extern "C" int cg_gconv(const cgConvGeom* g, const void* in, const void* bt, void* out,
                        int out_is_f32, const float* bias, const void* gate_in, float slope_in,
                        const void* gate_out, float slope_out, const void* residual,
                        cgStream stream) {
  int rc = check_geom(g, "cg_gconv");
  if (rc) return rc;
  if (!in || !bt || !out) CG_FAIL(CG_ERR_BAD_ARG, "cg_gconv: null tensor");
  if (g->kh == 1 && g->kw == 1 && g->S == 1 && g->U == 1 && g->Hin == 1 && g->Win == 1 &&
      g->Co <= 4 && (g->Ci % 8) == 0 && g->Ci >= 4096 && (int64_t)g->N * g->Co <= 4096 &&
      (!gate_in || gate_in == in)) {
    // few outputs, long K: one pass over x per output instead of an MFMA tile with one useful column
    GConvArgs r;
    memset(&r, 0, sizeof(r));
    r.in = (const bf16_t*)in; r.bt = (const bf16_t*)bt; r.out = out; r.bias = bias;
    r.gate_in = (const bf16_t*)gate_in;
    r.self_gate = (gate_out != nullptr && gate_out == out);
    r.gate_out = r.self_gate ? nullptr : (const bf16_t*)gate_out;
    r.residual = (const bf16_t*)residual;
    r.Ci = g->Ci; r.Co = g->Co; r.M = g->N; r.K = g->Ci; r.Kp = (g->Ci + 7) & ~7;
    r.slope_in = slope_in; r.slope_out = slope_out; r.out_f32 = out_is_f32;
    dim3 grid(g->N, g->Co);
    hipStream_t fst = (hipStream_t)stream;
    CgProfScope prof(CG_PROF_GCONV_GENERIC, g, fst);
    rowdot_linear_kernel<<<grid, 1024, 0, fst>>>(r);
    CG_CHECK_LAUNCH("cg_gconv(row dot)");
    return CG_OK;
  }
  if (cg_wstem_conv_supported(g, in, out, gate_in, slope_in, gate_out, residual)) {
    hipStream_t fst = (hipStream_t)stream;
    cg_wstem_conv_launch(g, in, bt, out, out_is_f32, bias, gate_in, gate_out, slope_out, fst);
    CG_CHECK_LAUNCH("cg_gconv(wstem)");
    return CG_OK;
  }
  if (cg_stem_conv_supported(g, in, out, gate_in, slope_in, gate_out, residual)) {
    hipStream_t fst = (hipStream_t)stream;
    cg_stem_conv_launch(g, in, bt, out, out_is_f32, bias, gate_in, gate_out, slope_out, fst);
    CG_CHECK_LAUNCH("cg_gconv(stem)");
    return CG_OK;
  }
  if (cg_sconv_supported(g, in, gate_in, slope_in)) {
    hipStream_t fst = (hipStream_t)stream;
    cg_sconv_launch(g, in, bt, out, out_is_f32, bias, gate_in, gate_out, slope_out, residual, fst);
    CG_CHECK_LAUNCH("cg_gconv(small)");
    return CG_OK;
  }
  if (cg_hconv_rw_supported(g, in, gate_in, slope_in)) {
    hipStream_t fst = (hipStream_t)stream;
    cg_hconv_rw_launch(g, in, bt, out, out_is_f32, bias, gate_in, gate_out, slope_out, residual,
                       fst);
    CG_CHECK_LAUNCH("cg_gconv(halo-rw)");
    return CG_OK;
  }
  if (cg_hconv_supported(g, in, gate_in, slope_in) &&
      !(cg_hconv_narrow(g) && ((gate_out && gate_out != out) || residual))) {
    hipStream_t fst = (hipStream_t)stream;
    cg_hconv_launch(g, in, bt, out, out_is_f32, bias, gate_in, gate_out, slope_out, residual, fst);
    CG_CHECK_LAUNCH("cg_gconv(halo)");
    return CG_OK;
  }
  if (cg_fast_conv_supported(g, in, gate_in, slope_in)) {
    hipStream_t fst = (hipStream_t)stream;
    cg_fast_conv_launch(g, in, bt, out, out_is_f32, bias, gate_in, gate_out, slope_out, residual,
                        fst);
    CG_CHECK_LAUNCH("cg_gconv(fast)");
    return CG_OK;
  }
  GConvArgs a;
  a.in = (const bf16_t*)in;
  a.bt = (const bf16_t*)bt;
  a.out = out;
  a.bias = bias;
  a.gate_in = (const bf16_t*)gate_in;
  a.self_gate = (gate_out != nullptr && gate_out == out);
  a.gate_out = a.self_gate ? nullptr : (const bf16_t*)gate_out;
  a.residual = (const bf16_t*)residual;
  a.N = g->N; a.Hin = g->Hin; a.Win = g->Win; a.Ci = g->Ci;
  a.Ho = g->Ho; a.Wo = g->Wo; a.Co = g->Co; a.kh = g->kh; a.kw = g->kw;
  a.S = g->S; a.ulog = ilog2_exact(g->U); a.pt = g->pt; a.pl = g->pl;
  a.M = g->N * g->Ho * g->Wo;
  a.K = g->kh * g->kw * g->Ci;
  a.Kp = (a.K + 7) & ~7;
  a.HinU = g->Hin * g->U; a.WinU = g->Win * g->U;
  a.slope_in = slope_in; a.slope_out = slope_out;
  a.out_f32 = out_is_f32;
  a.dWo = make_fastdiv(g->Wo); a.dHo = make_fastdiv(g->Ho);
  a.dCi = make_fastdiv(g->Ci); a.dKw = make_fastdiv(g->kw);
  const bool vec = (g->Ci % 8) == 0;
  hipStream_t st = (hipStream_t)stream;
  CgProfScope prof(CG_PROF_GCONV_GENERIC, g, st);
  if (g->kh == 1 && g->kw == 1 && g->S == 1 && g->U == 1 && g->Ci <= 8 && a.Kp == 8) {
    const int groups = cdiv(g->Co, 8);
    const int gpb = groups < 64 ? groups : 64, ppb = 256 / gpb;
    const int spans = cdiv(a.M, THIN_PIX * ppb);
    const int iters = spans >= 8192 ? 8 : (spans >= 2048 ? 4 : 1);
    dim3 grid(cdiv(spans, iters), cdiv(groups, gpb));
    thin_conv_kernel<<<grid, 256, 0, st>>>(a, groups, gpb, ppb, iters);
    CG_CHECK_LAUNCH("cg_gconv(thin)");
    return CG_OK;
  }
  if (g->kh == 1 && g->kw == 1 && g->S == 1 && g->U == 1 && g->Hin == 1 && g->Win == 1 &&
      a.M <= 512 && !vec) {
    dim3 grid(cdiv(a.M, 64), cdiv(g->Co, 8));
    small_linear_kernel<<<grid, 256, 0, st>>>(a);
    CG_CHECK_LAUNCH("cg_gconv(small linear)");
    return CG_OK;
  }
  if (g->Co > 64)
    launch_gconv<128, 128, 2, 2>(a, vec, st);
  else if (g->Co > 32)
    launch_gconv<128, 64, 2, 2>(a, vec, st);
  else
    launch_gconv<128, 32, 4, 1>(a, vec, st);
  CG_CHECK_LAUNCH("cg_gconv");
  return CG_OK;
}

extern "C" int cg_gconv_ld_supported(const cgConvGeom* g, int in_ld, int out_ld) {
  if (!g || check_geom(g, "cg_gconv_ld_supported")) return 0;
  return cg_fast_conv_ld_supported(g, in_ld, out_ld) ? 1 : 0;
}

extern "C" int cg_gconv_ld(const cgConvGeom* g, const void* in, int in_ld, const void* bt, void* out,
                           int out_ld, int out_is_f32, const float* bias, int relu_cols,
                           cgStream stream) {
  int rc = check_geom(g, "cg_gconv_ld");
  if (rc) return rc;
  if (!in || !bt || !out) CG_FAIL(CG_ERR_BAD_ARG, "cg_gconv_ld: null tensor");
  if (!cg_fast_conv_ld_supported(g, in_ld, out_ld))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_gconv_ld: no kernel with pixel pitches for this geometry "
                            "(cg_gconv_ld_supported)");
  if (((uintptr_t)in & 15) || ((uintptr_t)out & 15))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_gconv_ld: slices must start on 16-byte boundaries");
  if (relu_cols < 0 || relu_cols > g->Co || (relu_cols & 7))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_gconv_ld: relu_cols must be a multiple of 8 in [0, Co]");
  hipStream_t st = (hipStream_t)stream;
  cg_fast_conv_launch_ld_cols(g, in, in_ld, bt, out, out_ld, out_is_f32, bias, relu_cols, st);
  CG_CHECK_LAUNCH("cg_gconv_ld");
  return CG_OK;
}

extern "C" int cg_gconv_fused_phases(const cgConvGeom* g) {
  if (!g || check_geom(g, "cg_gconv_fused_phases") || !cg_hconv_geom_ok(g)) return 0;
  return cg_hconv_stats_phases(g);
}

extern "C" int cg_gconv_fused_rows(const cgConvGeom* g) {
  if (!g || check_geom(g, "cg_gconv_fused_rows")) return 0;
  // the fused kernel is the halo-staged one; small grids are covered too (min work-groups = 1 here:
  // the alternative costs two more full passes over the activation)
  if (!cg_hconv_geom_ok(g)) return 0;
  return cg_hconv_stats_rows(g);
}

extern "C" int cg_gconv_fused_prologue_supported(const cgConvGeom* g) {
  if (!g || check_geom(g, "cg_gconv_fused_prologue_supported")) return 0;
  return (cg_hconv_narrow(g) ? cg_hconv_narrow_ok(g) : cg_hconv_geom_ok(g)) ? 1 : 0;
}

extern "C" int cg_gconv_pool_supported(const cgConvGeom* g) {
  if (!g || check_geom(g, "cg_gconv_pool_supported")) return 0;
  if ((g->Ho & 1) || (g->Wo & 1) || g->U != 1) return 0;
  if (g->Ci == 3)
    return cg_wstem_conv_supported(g, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr) &&
                   (g->Co == 64 || g->Co == 128) &&
                   cg_wstem_wgrad_supported(g, nullptr, nullptr, 0.f, nullptr)
               ? 1 : 0;
  return cg_hconv_geom_ok(g) && cg_hwgrad_workspace_bytes(g) > 0 ? 1 : 0;
}

extern "C" int cg_gconv_fused(const cgConvGeom* g, const void* in, const void* bt, void* out,
                              int out_is_f32, const float* bias, const void* gate_in,
                              float slope_in, const void* gate_out, float slope_out,
                              const void* residual, const cgConvFusion* fu, cgStream stream) {
  int rc = check_geom(g, "cg_gconv_fused");
  if (rc) return rc;
  if (!in || !bt || !out || !fu) CG_FAIL(CG_ERR_BAD_ARG, "cg_gconv_fused: null argument");
  if (gate_in && !(gate_in == in && slope_in == 0.f))
    CG_FAIL(CG_ERR_UNSUPPORTED, "cg_gconv_fused: the input gate must be ReLU of the input itself");
  if ((fu->bn_mean == nullptr) != (fu->bn_var == nullptr))
    CG_FAIL(CG_ERR_BAD_ARG, "cg_gconv_fused: bn_mean and bn_var go together");
  if ((fu->pool_out || fu->in_up) && (g->U != 1 || (g->Ho & 1) || (g->Wo & 1)))
    CG_FAIL(CG_ERR_UNSUPPORTED, "cg_gconv_fused: pooling needs an even unit-stride geometry");
  if (fu->pool_out && gate_out)
    CG_FAIL(CG_ERR_UNSUPPORTED, "cg_gconv_fused: no output gate on a pooled output");
  hipStream_t fst = (hipStream_t)stream;
  if (g->Ci == 3) {
    // RGB-input convolution: only the pooled epilogue exists
    if (!fu->pool_out || fu->bn_mean || fu->stats_out || fu->in_up || residual ||
        !cg_gconv_pool_supported(g))
      CG_FAIL(CG_ERR_UNSUPPORTED, "cg_gconv_fused: RGB-input form only covers pool_out");
    cg_wstem_conv_launch_pool(g, in, bt, out, out_is_f32, bias, gate_in, nullptr, 0.f, 1, fst);
    CG_CHECK_LAUNCH("cg_gconv_fused(wstem)");
    return CG_OK;
  }
  if (cg_hconv_narrow(g)) {
    if (!cg_hconv_narrow_ok(g)) CG_FAIL(CG_ERR_UNSUPPORTED, "cg_gconv_fused: geometry not covered");
    if (fu->stats_out || fu->pool_out || fu->in_up || residual || (gate_out && gate_out != out))
      CG_FAIL(CG_ERR_UNSUPPORTED, "cg_gconv_fused: fewer than 8 output channels: batch-norm prologue only");
  } else if (!cg_hconv_geom_ok(g)) {
    CG_FAIL(CG_ERR_UNSUPPORTED, "cg_gconv_fused: geometry not covered");
  }
  cg_hconv_launch_fused(g, in, bt, out, out_is_f32, bias, gate_in, gate_out, slope_out, residual,
                        fu, fst);
  CG_CHECK_LAUNCH("cg_gconv_fused");
  return CG_OK;
}

extern "C" int cg_gwgrad_pooled(const cgConvGeom* g, const void* in, const void* gate_in,
                                float slope_in, const void* dy_pooled, float* dw, int accumulate,
                                float* dbias, void* ws, size_t ws_bytes, cgStream stream) {
  int rc = check_geom(g, "cg_gwgrad_pooled");
  if (rc) return rc;
  if (!in || !dy_pooled || !dw) CG_FAIL(CG_ERR_BAD_ARG, "cg_gwgrad_pooled: null tensor");
  if (!cg_gconv_pool_supported(g))
    CG_FAIL(CG_ERR_UNSUPPORTED, "cg_gwgrad_pooled: geometry not covered");
  if (gate_in && !(gate_in == in && slope_in == 0.f))
    CG_FAIL(CG_ERR_UNSUPPORTED, "cg_gwgrad_pooled: the input gate must be ReLU of the input itself");
  if (!ws || ws_bytes < cg_gwgrad_workspace_bytes(g))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_gwgrad_pooled: workspace too small");
  hipStream_t fst = (hipStream_t)stream;
  if (g->Ci == 3) {
    CgProfScope prof(CG_PROF_STEM_WGRAD, g, fst);
    int splits = 0;
    cg_wstem_wgrad_launch_pooled(g, in, gate_in, dy_pooled, 1, dbias != nullptr, ws, &splits, fst);
    cg_stem_partial_reduce(g, ws, splits, dw, dbias, accumulate, fst);
  } else {
    cg_hwgrad_launch_pooled(g, in, gate_in, dy_pooled, 1, dw, accumulate, dbias, ws, fst);
  }
  CG_CHECK_LAUNCH("cg_gwgrad_pooled");
  return CG_OK;
}

static size_t gwgrad_slow_workspace_bytes(const cgConvGeom* g);

static void wgrad_plan(const cgConvGeom* g, int* tk, int* tn, int* splits, int* rows_per_split) {
  const int K = g->kh * g->kw * g->Ci;
  const int M = g->N * g->Ho * g->Wo;
  *tk = K > 64 ? 128 : 64;
  *tn = g->Co > 64 ? 128 : 64;
  const int tiles = cdiv(K, *tk) * cdiv(g->Co, *tn);
  int s = cdiv(1024, tiles);
  const int max_by_rows = M / 128 > 0 ? M / 128 : 1;  // at least 4 slices of 32 rows per split
  if (s > max_by_rows) s = max_by_rows;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  int rps = cdiv(M, s);
  rps = (rps + 31) / 32 * 32;
  s = cdiv(M, rps);
  *splits = s;
  *rows_per_split = rps;
}

extern "C" size_t cg_gwgrad_workspace_bytes(const cgConvGeom* g) {
  if (!g) return 0;
  // the fast path may or may not be taken (it depends on the gates): size for the larger need
  size_t fast = 0;
  if (cg_fast_wgrad_supported(g, nullptr, nullptr, 0.f, nullptr))
    fast = cg_fast_wgrad_workspace_bytes(g);
  if (cg_stem_wgrad_supported(g, nullptr, nullptr, 0.f, nullptr))
    fast = cg_stem_wgrad_workspace_bytes(g);
  if (cg_narrow_wgrad_supported(g, nullptr, nullptr)) {
    fast = cg_narrow_wgrad_workspace_bytes(g);
    const size_t cs = cg_colsum_workspace_bytes((int64_t)g->N * g->Ho * g->Wo, g->Co);
    if (cs > fast) fast = cs;
  }
  const size_t halo = cg_hwgrad_workspace_bytes(g);   // 0 when the geometry is not covered
  if (halo > fast) fast = halo;
  const size_t slow = gwgrad_slow_workspace_bytes(g);
  return fast > slow ? fast : slow;
}

static size_t gwgrad_slow_workspace_bytes(const cgConvGeom* g) {
  int tk, tn, splits, rps;
  wgrad_plan(g, &tk, &tn, &splits, &rps);
  if (splits == 1) return 256;
  const size_t K = (size_t)g->kh * g->kw * g->Ci;
  return align_up((size_t)splits * (K * g->Co + g->Co) * sizeof(float), 256);
}

extern "C" int cg_gwgrad(const cgConvGeom* g, const void* in, const void* gate_in, float slope_in,
                         const void* dy, const void* gate_dy, float slope_dy, float* dw,
                         int accumulate, float* dbias, void* ws, size_t ws_bytes,
                         cgStream stream) {
  int rc = check_geom(g, "cg_gwgrad");
  if (rc) return rc;
  if (!in || !dy || !dw) CG_FAIL(CG_ERR_BAD_ARG, "cg_gwgrad: null tensor");
  if (!ws || ws_bytes < cg_gwgrad_workspace_bytes(g))
    CG_FAIL(CG_ERR_WORKSPACE, "cg_gwgrad: workspace too small (%zu < %zu)", ws_bytes,
            cg_gwgrad_workspace_bytes(g));
  if (cg_stem_wgrad_supported(g, in, gate_in, slope_in, gate_dy)) {
    hipStream_t fst = (hipStream_t)stream;
    cg_stem_wgrad_launch(g, in, gate_in, dy, dw, accumulate, dbias, ws, fst);
    CG_CHECK_LAUNCH("cg_gwgrad(stem)");
    return CG_OK;
  }
  if (cg_narrow_wgrad_supported(g, gate_in, gate_dy)) {
    hipStream_t fst = (hipStream_t)stream;
    ReduceDeferSuspend keep(dbias != nullptr);   // cg_colsum below reuses the workspace
    cg_narrow_wgrad_launch(g, in, dy, dw, accumulate, ws, fst);
    CG_CHECK_LAUNCH("cg_gwgrad(narrow)");
    int rc2 = CG_OK;
    if (dbias) {
      if (accumulate) CG_FAIL(CG_ERR_UNSUPPORTED, "cg_gwgrad: accumulate with dbias on a narrow conv");
      rc2 = cg_colsum(dy, (int64_t)g->N * g->Ho * g->Wo, g->Co, dbias, ws, ws_bytes, stream);
    }
    return rc2;
  }
  if (cg_swgrad_supported(g, in, gate_in, slope_in, gate_dy, false)) {
    hipStream_t fst = (hipStream_t)stream;
    cg_swgrad_launch(g, in, gate_in, dy, dw, accumulate, dbias, fst);
    CG_CHECK_LAUNCH("cg_gwgrad(small)");
    return CG_OK;
  }
  if (cg_hwgrad_supported(g, in, gate_in, slope_in, gate_dy)) {
    hipStream_t fst = (hipStream_t)stream;
    cg_hwgrad_launch(g, in, gate_in, dy, dw, accumulate, dbias, ws, fst);
    CG_CHECK_LAUNCH("cg_gwgrad(halo)");
    return CG_OK;
  }
  if (cg_fast_wgrad_supported(g, in, gate_in, slope_in, gate_dy)) {
    hipStream_t fst = (hipStream_t)stream;
    cg_fast_wgrad_launch(g, in, gate_in, dy, dw, accumulate, dbias, ws, fst);
    CG_CHECK_LAUNCH("cg_gwgrad(fast)");
    return CG_OK;
  }
  int tk, tn, splits, rps;
  wgrad_plan(g, &tk, &tn, &splits, &rps);
  const int K = g->kh * g->kw * g->Ci;
  GWgradArgs a;
  a.in = (const bf16_t*)in; a.gate_in = (const bf16_t*)gate_in;
  a.dy = (const bf16_t*)dy; a.gate_dy = (const bf16_t*)gate_dy;
  a.N = g->N; a.Hin = g->Hin; a.Win = g->Win; a.Ci = g->Ci;
  a.Ho = g->Ho; a.Wo = g->Wo; a.Co = g->Co; a.kh = g->kh; a.kw = g->kw;
  a.S = g->S; a.ulog = ilog2_exact(g->U); a.pt = g->pt; a.pl = g->pl;
  a.M = g->N * g->Ho * g->Wo; a.K = K;
  a.HinU = g->Hin * g->U; a.WinU = g->Win * g->U;
  a.rows_per_split = rps;
  a.accumulate = accumulate;
  a.slope_in = slope_in; a.slope_dy = slope_dy;
  a.dWo = make_fastdiv(g->Wo); a.dHo = make_fastdiv(g->Ho);
  a.dCi = make_fastdiv(g->Ci); a.dKw = make_fastdiv(g->kw);
  float* wsf = (float*)ws;
  if (splits == 1) {
    a.out = dw;
    a.bias_out = dbias;
  } else {
    a.out = wsf;
    a.bias_out = dbias ? wsf + (size_t)splits * K * g->Co : nullptr;
  }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(cdiv(K, tk), cdiv(g->Co, tn), splits);
  CgProfScope prof(CG_PROF_GWGRAD_GENERIC, g, st);
  const bool vx = (g->Ci % 8) == 0, vy = (g->Co % 8) == 0;
#define CG_WG(TK_, TN_)                                                        \
  do {                                                                         \
    if (vx && vy) gwgrad_kernel<TK_, TN_, true, true><<<grid, 256, 0, st>>>(a);        \
    else if (vx) gwgrad_kernel<TK_, TN_, true, false><<<grid, 256, 0, st>>>(a);        \
    else if (vy) gwgrad_kernel<TK_, TN_, false, true><<<grid, 256, 0, st>>>(a);        \
    else gwgrad_kernel<TK_, TN_, false, false><<<grid, 256, 0, st>>>(a);               \
  } while (0)
  if (tk == 128 && tn == 128) CG_WG(128, 128);
  else if (tk == 128 && tn == 64) CG_WG(128, 64);
  else if (tk == 64 && tn == 128) CG_WG(64, 128);
  else CG_WG(64, 64);
#undef CG_WG
  CG_CHECK_LAUNCH("cg_gwgrad");
  if (splits > 1) {
    const int64_t n = (int64_t)K * g->Co;
    split_reduce_kernel<<<cdiv(n, 256), 256, 0, st>>>(wsf, splits, n, dw, accumulate);
    CG_CHECK_LAUNCH("cg_gwgrad(reduce)");
    if (dbias) {
      split_reduce_kernel<<<cdiv(g->Co, 256), 256, 0, st>>>(wsf + (size_t)splits * K * g->Co,
                                                             splits, g->Co, dbias, accumulate);
      CG_CHECK_LAUNCH("cg_gwgrad(reduce bias)");
    }
  }
  return CG_OK;
}

// Several weight gradients in one call (the deferred weight gradients of a backward pass,
// compare_gan_amd/hip/functional.py): the small-map ones -- which cannot fill the chip one at a
// time -- share launches of up to CG_SWGRAD_MAX_JOBS layers; the others run one by one.
extern "C" int cg_gwgrad_deferred(const cgConvGeom* g, const void* in, const void* gate_in,
                                  float slope_in, const void* dy, const void* gate_dy,
                                  float slope_dy, float* dw, int accumulate, float* dbias, void* ws,
                                  size_t ws_bytes, cgStream stream, cgDeferCtx* defer) {
  ReduceDeferScope scope(defer);   // the launchers below record into the caller's context
  return cg_gwgrad(g, in, gate_in, slope_in, dy, gate_dy, slope_dy, dw, accumulate, dbias, ws,
                   ws_bytes, stream);
}
extern "C" int cg_gwgrad_pooled_deferred(const cgConvGeom* g, const void* in, const void* gate_in,
                                         float slope_in, const void* dy_pooled, float* dw,
                                         int accumulate, float* dbias, void* ws, size_t ws_bytes,
                                         cgStream stream, cgDeferCtx* defer) {
  ReduceDeferScope scope(defer);
  return cg_gwgrad_pooled(g, in, gate_in, slope_in, dy_pooled, dw, accumulate, dbias, ws, ws_bytes,
                          stream);
}

extern "C" int cg_gwgrad_groupable(const cgConvGeom* g) {
  if (!g || check_geom(g, "cg_gwgrad_groupable")) return 0;
  return cg_swgrad_supported(g, nullptr, nullptr, 0.f, nullptr, true) ? 1 : 0;
}

extern "C" int cg_gwgrad_multi(const cgWgradItem* items_host, int n, void* ws, size_t ws_bytes,
                               cgStream stream) {
  if (n < 0 || (n > 0 && !items_host)) CG_FAIL(CG_ERR_BAD_ARG, "cg_gwgrad_multi: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const cgConvGeom* geoms[CG_SWGRAD_MAX_JOBS];
  const void* ins[CG_SWGRAD_MAX_JOBS];
  const void* dys[CG_SWGRAD_MAX_JOBS];
  float* dws[CG_SWGRAD_MAX_JOBS];
  float* dbs[CG_SWGRAD_MAX_JOBS];
  int relus[CG_SWGRAD_MAX_JOBS], accs[CG_SWGRAD_MAX_JOBS];
  int m = 0;
  auto flush = [&]() -> int {
    if (m == 0) return CG_OK;
    double fl = 0, by = 0;
    if (cg_prof_enabled())
      for (int i = 0; i < m; ++i) {
        double f, b;
        cg_conv_algorithmic_cost(geoms[i], &f, &b);
        fl += f;
        by += b;
      }
    if (cg_prof_enabled()) cg_prof_begin(CG_PROF_SWGRAD, fl, by, st);
    cg_swgrad_launch_multi(geoms, ins, relus, dys, dws, accs, dbs, m, st);
    cg_prof_end(CG_PROF_SWGRAD, st);
    m = 0;
    CG_CHECK_LAUNCH("cg_gwgrad_multi(small)");
    return CG_OK;
  };
  for (int i = 0; i < n; ++i) {
    const cgWgradItem& it = items_host[i];
    int rc = check_geom(&it.geom, "cg_gwgrad_multi");
    if (rc) return rc;
    if (!it.in || !it.dy || !it.dw) CG_FAIL(CG_ERR_BAD_ARG, "cg_gwgrad_multi: null tensor");
    if (cg_swgrad_supported(&it.geom, it.in, it.gate_in, it.slope_in, nullptr, true)) {
      geoms[m] = &it.geom; ins[m] = it.in; dys[m] = it.dy; dws[m] = it.dw; dbs[m] = it.dbias;
      relus[m] = it.gate_in != nullptr; accs[m] = it.accumulate;
      if (++m == CG_SWGRAD_MAX_JOBS) {
        rc = flush();
        if (rc) return rc;
      }
    } else {
      ReduceDeferSuspend keep(true);   // the one-by-one items share `ws`: reduce before the next one
      rc = cg_gwgrad(&it.geom, it.in, it.gate_in, it.slope_in, it.dy, nullptr, 0.f, it.dw,
                     it.accumulate, it.dbias, ws, ws_bytes, stream);
      if (rc) return rc;
    }
  }
  return flush();
}
