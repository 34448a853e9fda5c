// Fast paths of the gather convolution for gfx950 (the shapes that carry the FLOPs of the hot path:
// channel counts that are multiples of 64).  Contract and reference call sites: include/cgamd.h
// (cg_gconv / cg_gwgrad); this file only adds faster kernels behind the same entry points.
//
// Design (MI355X_MICROARCH.md / cdna_hip_programming.md sections 2, 3, 5):
//  * im2col-free implicit GEMM.  One K-slice = 64 consecutive channels of ONE filter tap, i.e. a
//    128-byte contiguous run per output pixel, so a tile row is one cache line and the tap decode is
//    wave-uniform (scalar registers); per-lane work in the K loop is a bounds test and an add.
//  * global -> LDS with the direct-to-LDS DMA (global_load_lds_dwordx4): no staging VGPRs, no
//    ds_write pass.  The LDS image is lane-linear, so the bank-conflict swizzle is applied on the
//    per-lane SOURCE address (16-byte chunk c of row r is stored at slot c ^ ((r >> 1) & 7)) and on
//    the fragment reads; with 128-byte rows this makes every ds_read_b128 lane group hit 16 distinct
//    16-byte slots.  Out-of-image taps / out-of-range rows read a 16-byte zero page instead.
//  * zero-inserted inputs (resnet_ops.unpool + conv, conv2d_transpose, the data gradient of a
//    stride-2 conv) are decomposed into U*U output phases, each a dense convolution over the taps
//    that can touch a non-zero sample: no MAC is spent on structural zeros (4x fewer for U = 2).
//  * MFMA operands are swapped (D[co][pixel] = W^T X^T) so that every lane owns 4 consecutive output
//    channels of one pixel: the epilogue (bias, ReLU'-gate, residual) runs on 8-byte vectors and
//    stores 8 (bf16) / 16 (fp32) bytes per lane.
//  * blockIdx -> tile mapping is XCD-aware: each of the 8 XCDs (private L2) gets a contiguous range
//    of tiles, neighbouring M tiles share their halo rows and the weight panel in that L2.
//  * weight gradient: both operands are K(=pixel)-major in memory but the MFMA wants K contiguous
//    per lane; the 4x4 hardware transpose read ds_read_b64_tr_b16 delivers exactly that from the
//    row-major LDS image, so the DMA staging is shared with the forward kernel.
#include "cg_conv_fast.h"

#include <mutex>
#include <new>
#include <vector>

#include <stdlib.h>

#define CG_FAST_BIAS_SPLITS 512

namespace {

__device__ __attribute__((aligned(16))) uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ __forceinline__ void glds16(const void* gptr, bf16_t* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)gptr, (lds_void_t*)lds_wave_base, 16, 0, 0);
}

// ASMDMA: the LDS-DMA goes through inline asm (cg_common.h: cg_glds16_asm) so that hipcc does not put
// `s_waitcnt vmcnt(0)` in front of the transposed LDS reads that follow it in program order -- the
// weight-gradient kernels below stage slice i + 1 before they multiply slice i
template <bool ASMDMA>
__device__ __forceinline__ void glds16x(const void* gptr, bf16_t* lds_wave_base) {
  if constexpr (ASMDMA) cg_glds16_asm(gptr, cg_lds_addr(lds_wave_base));
  else glds16(gptr, lds_wave_base);
}

__device__ __forceinline__ bf16x8_t relu_bf16x8(bf16x8_t v) {
  // bf16 sign bit == int16 sign bit: max_i16(x, 0) is relu(x) (and maps -0.0 to +0.0)
  s16x8_t s = __builtin_bit_cast(s16x8_t, v);
  const s16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
  s = __builtin_elementwise_max(s, z);
  return __builtin_bit_cast(bf16x8_t, s);
}

// bijective XCD remap (block b runs on XCD b % 8): XCD x owns a contiguous range of work ids
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// -------------------------------------------------------------------------------------------
// forward / data-gradient / transposed convolution
// -------------------------------------------------------------------------------------------
struct FastConvArgs {
  const bf16_t* in;
  const bf16_t* bt;
  void* out;
  const float* bias;
  const bf16_t* gate_out;
  const bf16_t* residual;
  int N, Hin, Win, Ci, Ho, Wo, Co, kh, kw, S, U, pt, pl;
  int Kp;
  int in_ld, out_ld;  // elements between consecutive pixels of the input / output (Ci / Co when dense;
                      // wider when the tensor is a channel slice of a concatenation, cg_gconv_ld)
  int Hp, Wp, Mp;  // per-phase output grid (Ho/U, Wo/U) and N*Hp*Wp
  int cblocks;     // Ci / 64
  int mtiles, ntiles;
  int gm;          // M tiles per tile group (1: all N tiles of an M tile are consecutive work ids)
  FastDiv dGrp;    // gm * ntiles
  int relu_in, out_f32;
  int self_gate;  // gate_out == out: the output activation is applied to the value itself
  int gate_cols;  // ... to output channels [0, gate_cols) only (a multiple of 8; Co: all of them) -- the
                  // coalesced epilogue (Co % 8 == 0); cg_gconv_ld's relu_cols
  float slope_out;
  FastDiv dWp, dHp, dNt;
#ifdef CG_CONV_TIMING
  unsigned long long* tdbg;   // s_memtime stamps of workgroup 0 (scripts/conv_timing.py)
#endif
};
#ifdef CG_CONV_TIMING
#define TSTAMP()                                                                      \
  do {                                                                                \
    if (a.tdbg && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && tsi < 250)       \
      a.tdbg[wave * 256 + tsi] = __builtin_amdgcn_s_memtime();                        \
    ++tsi;                                                                            \
  } while (0)
#else
#define TSTAMP() do {} while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NS = LDS ring depth: NS - 1 K-slices are in flight while one is being consumed.  NS = 2 keeps two
// workgroups per CU (the second hides the load latency); NS = 4 is for grids of about one workgroup
// per CU, where only a deeper ring can hide it.
template <int BM, int BN, bool RELU, int NS>
__global__ __launch_bounds__(256, NS == 1 ? 3 : 1) void fast_conv_kernel(FastConvArgs a) {
  constexpr int WN = BN >= 128 ? 2 : 1;   // waves along the channel dimension
  constexpr int WM = 4 / WN;
  constexpr int AJ = BM / 32;             // A staging instructions per wave (8 rows of 128 B each)
  constexpr int BJ = BN / 32;             // B staging instructions per wave
  constexpr int TM = BM / WM / 32;        // 32-pixel MFMA tiles per wave
  constexpr int TN = BN / WN / 32;        // 32-channel MFMA tiles per wave
  constexpr int A_ELEMS = BM * 64, B_ELEMS = BN * 64;
  // NS = 1: one buffer, two barriers per K-slice, three workgroups per CU overlap each other;
  // the staged epilogue needs (BM / WM) x (BN + 4) floats
  constexpr int EPI_ELEMS = (BM / WM) * (BN + 4) * 2;
  constexpr int RING_ELEMS = NS * (A_ELEMS + B_ELEMS);
  __shared__ __attribute__((aligned(1024)))
  bf16_t smem[RING_ELEMS > EPI_ELEMS || NS > 1 ? RING_ELEMS : EPI_ELEMS];
  constexpr int LOADS = AJ + BJ;   // LDS-DMA instructions per thread per stage
  constexpr int D = NS - 1;        // prefetch distance

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef CG_CONV_TIMING
  int tsi = 0;
#endif
  TSTAMP();   // 0: kernel entry
  const int wm = wave / WN, wn = wave % WN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  // work id -> (M tile, N tile).  gm == 1: the N tiles of an M tile are consecutive (they share the
  // activation rows; the weight panels of all N tiles have to fit the XCD's L2 beside them).  Weight
  // images far larger than an L2 (BigGAN's 3x3x1536x1536: 42 MB) were streamed once per ~6 M tiles that
  // happened to run together (counter traffic 10x algorithmic, profiles/r06_pmc_traffic.json); with
  // gm > 1 a group of gm M tiles walks the N tiles together, M fastest: the workgroups in flight on an XCD
  // hold gm activation tiles and only ~64 / gm weight panels.
  int mt, nt;
  if (a.gm <= 1) {
    mt = (int)fdiv((uint32_t)wg, a.dNt);
    nt = wg - mt * a.ntiles;
  } else {
    const int grp = (int)fdiv((uint32_t)wg, a.dGrp);
    const int r = wg - grp * a.gm * a.ntiles;
    const int mb = grp * a.gm;
    const int gcur = min(a.gm, a.mtiles - mb);   // the last group may be short
    nt = r / gcur;
    mt = mb + (r - nt * gcur);
  }
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- phase geometry (wave-uniform) ----
  const int phase = blockIdx.y;
  int r0 = 0, s0 = 0, nr = a.kh, ns = a.kw, bh = -a.pt, bw = -a.pl, ph = 0, pw = 0;
  if (a.U == 2) {
    ph = phase >> 1;
    pw = phase & 1;
    r0 = (a.pt + ph) & 1;
    s0 = (a.pl + pw) & 1;
    nr = (a.kh - r0 + 1) >> 1;
    ns = (a.kw - s0 + 1) >> 1;
    bh = (ph - a.pt + r0) >> 1;  // exact: the numerator is even
    bw = (pw - a.pl + s0) >> 1;
  }
  const int nk = nr * ns * a.cblocks;

  // ---- per-thread staging descriptors ----
  // instruction group g = wave*AJ + j stages rows g*8 .. g*8+7; lane -> row g*8 + (lane >> 3),
  // LDS slot (lane & 7) which must hold source chunk (lane & 7) ^ ((row >> 1) & 7)
  int a_off[AJ], a_ih[AJ], a_iw[AJ], a_c8[AJ];
  bool a_ok[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int g = wave * AJ + j;
    const int row = g * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    a_c8[j] = c * 8;
    const int m = m0 + row;
    a_ok[j] = m < a.Mp;
    const uint32_t mm = a_ok[j] ? (uint32_t)m : 0u;
    const uint32_t t1 = fdiv(mm, a.dWp);
    const int owp = (int)(mm - t1 * a.Wp);
    const uint32_t n = fdiv(t1, a.dHp);
    const int ohp = (int)(t1 - n * a.Hp);
    a_ih[j] = (a.U == 2) ? ohp + bh : ohp * a.S + bh;
    a_iw[j] = (a.U == 2) ? owp + bw : owp * a.S + bw;
    a_off[j] = (((int)n * a.Hin + a_ih[j]) * a.Win + a_iw[j]) * a.in_ld + c * 8;
  }
  int b_off[BJ], b_c8[BJ];
  bool b_ok[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int g = wave * BJ + j;
    const int row = g * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    b_c8[j] = c * 8;
    b_ok[j] = (n0 + row) < a.Co;
    b_off[j] = (n0 + row) * a.Kp + c * 8;
  }

  auto Abuf = [&](int buf) { return smem + buf * (A_ELEMS + B_ELEMS); };
  auto Bbuf = [&](int buf) { return smem + buf * (A_ELEMS + B_ELEMS) + A_ELEMS; };
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero16);

  // tap iteration state of the NEXT slice to stage (all scalar)
  int st_ri = 0, st_si = 0, st_cb = 0;
  auto stage = [&](int buf) {
    const int tapoff = (st_ri * a.Win + st_si) * a.in_ld + st_cb * 64;
    const int koff = ((r0 + a.U * st_ri) * a.kw + (s0 + a.U * st_si)) * a.Ci + st_cb * 64;
    bf16_t* Ab = Abuf(buf) + (wave * AJ) * 512;
    bf16_t* Bb = Bbuf(buf) + (wave * BJ) * 512;
    const int crem = a.Ci - st_cb * 64;   // channels left in this block (< 64 only for the last
                                          // block of a channel count that is not a multiple of 64)
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const bool ok = a_ok[j] && a_c8[j] < crem &&
                      (unsigned)(a_ih[j] + st_ri) < (unsigned)a.Hin &&
                      (unsigned)(a_iw[j] + st_si) < (unsigned)a.Win;
      const bf16_t* p = ok ? a.in + (int64_t)(a_off[j] + tapoff) : zero;
      glds16(p, Ab + j * 512);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const bf16_t* p = (b_ok[j] && b_c8[j] < crem) ? a.bt + (int64_t)(b_off[j] + koff) : zero;
      glds16(p, Bb + j * 512);
    }
    if (++st_cb == a.cblocks) {
      st_cb = 0;
      if (++st_si == ns) {
        st_si = 0;
        ++st_ri;
      }
    }
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  // fragment addressing: row = base + (lane & 31), chunk = kk*2 + (lane >> 5), swizzled
  const int frow = lane & 31;
  const int swz = (frow >> 1) & 7;
  int koffs[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) koffs[kk] = (((kk * 2 + (lane >> 5)) ^ swz) << 3);
  const int arow0 = (wm * (BM / WM) + frow) * 64;
  const int brow0 = (wn * (BN / WN) + frow) * 64;

  // two of the four 16-channel MFMA steps of a K-slice.  A channel count that is an odd multiple of
  // 32 (Inception: 32 / 96 / 160 / 288; BigGAN: 96) leaves the upper half of its LAST 64-channel
  // block empty: the slice still stages 128-byte rows (zero page), its upper two steps are skipped
  // (wave-uniform: the consumer's channel-block counter c_cb is scalar)
  auto mma2 = [&](const bf16_t* Ab, const bf16_t* Bb, int k0) {
#pragma unroll
    for (int kk = k0; kk < k0 + 2; ++kk) {
      bf16x8_t af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        af[i] = *reinterpret_cast<const bf16x8_t*>(Ab + arow0 + i * 32 * 64 + koffs[kk]);
        if (RELU) af[i] = relu_bf16x8(af[i]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8_t*>(Bb + brow0 + j * 32 * 64 + koffs[kk]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  };
  const bool ragged_k = (a.Ci & 63) != 0;
  int c_cb = 0;   // channel block of the slice being consumed
  auto half_slice = [&]() { return ragged_k && c_cb == a.cblocks - 1; };
  auto next_slice = [&]() { if (++c_cb == a.cblocks) c_cb = 0; };

  int staged = 0;
  TSTAMP();   // 1: descriptors done
  for (; staged < D && staged < nk; ++staged) stage(staged % NS);
  TSTAMP();   // 2: prologue stages issued

  if constexpr (NS == 1) {
    for (int it = 0; it < nk; ++it) {
      stage(0);
      wait_vmcnt<0>();
      __syncthreads();
      const bf16_t* Ab = Abuf(0);
      const bf16_t* Bb = Bbuf(0);
      mma2(Ab, Bb, 0);
      if (!half_slice()) mma2(Ab, Bb, 2);
      next_slice();
      __syncthreads();
    }
  } else
  for (int it = 0; it < nk; ++it) {
    // wait for slice `it` (the oldest in flight), then barrier: every wave's share of it has landed
    // and every wave has finished reading the buffer that is restaged below (slice it - 1's)
    const int ahead = staged - it - 1;
    if (D >= 3 && ahead >= 2) wait_vmcnt<2 * LOADS>();
    else if (D >= 2 && ahead == 1) wait_vmcnt<LOADS>();
    else wait_vmcnt<0>();
    TSTAMP();   // 3 + 4 it: slice landed (this wave's share)
    asm volatile("s_barrier" ::: "memory");
    TSTAMP();   // 4 + 4 it: barrier passed
    if (staged < nk) {
      stage(staged % NS);
      ++staged;
    }
    TSTAMP();   // 5 + 4 it: next slice issued
    const int buf = it % NS;
    const bf16_t* Ab = Abuf(buf);
    const bf16_t* Bb = Bbuf(buf);
    mma2(Ab, Bb, 0);
    if (!half_slice()) mma2(Ab, Bb, 2);
    next_slice();
    TSTAMP();   // 6 + 4 it: MFMA work of the slice issued
  }
  TSTAMP();     // 3 + 4 nk: loop done

  // ---- epilogue, coalesced form (Co % 8 == 0): the accumulators (+ bias, self-activation) go
  // through LDS in WM passes of BM/WM rows, then every thread finishes 8 consecutive channels of
  // one pixel: gate / residual are read and the result is written as 16-byte (bf16) or 2 x 16-byte
  // (fp32) pieces, whole rows contiguous across the lanes.  (The direct form below issues 8-byte
  // stores 256 B apart; measured, the output write was ~20 us of a 67 us launch.)
  if ((a.Co & 7) == 0) {
    constexpr int RP = BM / WM;          // rows per pass
    constexpr int LDC = BN + 4;          // floats per LDS row (+16 B: conflict-free b128 writes)
    constexpr int C8 = BN / 8;           // 8-channel items per row
    float* Cs = reinterpret_cast<float*>(smem);
    for (int h = 0; h < WM; ++h) {
      TSTAMP();   // epilogue pass h: entry
      __syncthreads();
      TSTAMP();   // barrier 1
      if (wm == h) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int col = wn * (BN / WN) + j * 32 + q * 8 + 4 * (lane >> 5);
              float4 v = make_float4(acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1],
                                     acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]);
              if (a.bias && n0 + col < a.Co) {
                const float4 b4 = *reinterpret_cast<const float4*>(a.bias + n0 + col);
                v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
              }
              if (a.self_gate && n0 + col < a.gate_cols) {
                if (!(v.x > 0.f)) v.x *= a.slope_out;
                if (!(v.y > 0.f)) v.y *= a.slope_out;
                if (!(v.z > 0.f)) v.z *= a.slope_out;
                if (!(v.w > 0.f)) v.w *= a.slope_out;
              }
              *reinterpret_cast<float4*>(Cs + (i * 32 + frow) * LDC + col) = v;
            }
      }
      TSTAMP();   // accumulators in LDS
      __syncthreads();
      TSTAMP();   // barrier 2
      for (int t = tid; t < RP * C8; t += 256) {
        const int row = t / C8, c8 = t - row * C8;
        const int m = m0 + h * RP + row;
        const int co = n0 + c8 * 8;
        if (m >= a.Mp || co >= a.Co) continue;
        const uint32_t t1 = fdiv((uint32_t)m, a.dWp);
        const int owp = m - (int)t1 * a.Wp;
        const uint32_t n = fdiv(t1, a.dHp);
        const int ohp = (int)t1 - (int)n * a.Hp;
        const int64_t o =
            ((int64_t)((int)n * a.Ho + ohp * a.U + ph) * a.Wo + (owp * a.U + pw)) * a.out_ld + co;
        const float4 lo = *reinterpret_cast<const float4*>(Cs + row * LDC + c8 * 8);
        const float4 hi = *reinterpret_cast<const float4*>(Cs + row * LDC + c8 * 8 + 4);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (a.gate_out) {
          union { uint4 q; bf16_t h8[8]; } g;
          g.q = *reinterpret_cast<const uint4*>(a.gate_out + o);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (!(bf2f(g.h8[e]) > 0.f)) v[e] *= a.slope_out;
        }
        if (a.residual) {
          union { uint4 q; bf16_t h8[8]; } r;
          r.q = *reinterpret_cast<const uint4*>(a.residual + o);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bf2f(r.h8[e]);
        }
        if (a.out_f32) {
          float* op = reinterpret_cast<float*>(a.out) + o;
          *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          union { uint4 q; bf16_t h8[8]; } w;
#pragma unroll
          for (int e = 0; e < 8; ++e) w.h8[e] = f2bf(v[e]);
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.out) + o) = w.q;
        }
      }
    }
    TSTAMP();   // last: epilogue stores issued
    return;
  }

  // ---- epilogue: lane owns pixel (lane & 31) of each M sub-tile and, per 8-channel group q, the
  // 4 consecutive channels 8*q + 4*(lane >> 5) + {0..3}
  const bool vec4 = (a.Co & 3) == 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * (BM / WM) + i * 32 + frow;
    if (m >= a.Mp) continue;
    const uint32_t t1 = fdiv((uint32_t)m, a.dWp);
    const int owp = m - (int)t1 * a.Wp;
    const uint32_t n = fdiv(t1, a.dHp);
    const int ohp = (int)t1 - (int)n * a.Hp;
    const int64_t opix =
        ((int64_t)((int)n * a.Ho + ohp * a.U + ph) * a.Wo + (owp * a.U + pw)) * a.out_ld;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = n0 + wn * (BN / WN) + j * 32 + q * 8 + 4 * (lane >> 5);
        if (co >= a.Co) continue;
        float v[4] = {acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2],
                      acc[i][j][q * 4 + 3]};
        const int64_t o = opix + co;
        if (vec4) {
          if (a.bias) {
            const float4 b4 = *reinterpret_cast<const float4*>(a.bias + co);
            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
          }
          if (a.self_gate) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (!(v[e] > 0.f)) v[e] *= a.slope_out;
          } else if (a.gate_out) {
            const uint2 g2 = *reinterpret_cast<const uint2*>(a.gate_out + o);
            if (!(bf2f((bf16_t)(g2.x & 0xffff)) > 0.f)) v[0] *= a.slope_out;
            if (!(bf2f((bf16_t)(g2.x >> 16)) > 0.f)) v[1] *= a.slope_out;
            if (!(bf2f((bf16_t)(g2.y & 0xffff)) > 0.f)) v[2] *= a.slope_out;
            if (!(bf2f((bf16_t)(g2.y >> 16)) > 0.f)) v[3] *= a.slope_out;
          }
          if (a.residual) {
            const uint2 r2 = *reinterpret_cast<const uint2*>(a.residual + o);
            v[0] += bf2f((bf16_t)(r2.x & 0xffff));
            v[1] += bf2f((bf16_t)(r2.x >> 16));
            v[2] += bf2f((bf16_t)(r2.y & 0xffff));
            v[3] += bf2f((bf16_t)(r2.y >> 16));
          }
          if (a.out_f32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + o) =
                make_float4(v[0], v[1], v[2], v[3]);
          } else {
            uint2 w2;
            w2.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
            w2.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.out) + o) = w2;
          }
        } else {
          // channel counts that are not a multiple of 4 (RGB outputs): element-wise tail
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (co + e >= a.Co) continue;
            float val = v[e] + (a.bias ? a.bias[co + e] : 0.f);
            if (a.self_gate) {
              if (!(val > 0.f)) val *= a.slope_out;
            } else if (a.gate_out && !(bf2f(a.gate_out[o + e]) > 0.f)) {
              val *= a.slope_out;
            }
            if (a.residual) val += bf2f(a.residual[o + e]);
            if (a.out_f32)
              reinterpret_cast<float*>(a.out)[o + e] = val;
            else
              reinterpret_cast<bf16_t*>(a.out)[o + e] = f2bf(val);
          }
        }
      }
    }
  }
}

// Intra-workgroup split-K form for grids of at most about one workgroup per CU: 8 waves = two groups
// of 4, each the 4-wave kernel above on every other K-slice with its own LDS ring; the groups hide
// each other's LDS-DMA issue time and load latency (what a co-resident workgroup does on large
// grids) and halve the serial slice count.  Group 1 hands its accumulators to group 0 through LDS,
// group 0 runs the epilogue.
template <int BM, int BN, bool RELU>
__global__ __launch_bounds__(512) void fast_conv_sk_kernel(FastConvArgs a) {
  constexpr int NS = (BM * BN >= 128 * 128) ? 2 : 3;
  constexpr int WN = BN >= 128 ? 2 : 1;   // waves along the channel dimension
  constexpr int WM = 4 / WN;
  constexpr int AJ = BM / 32;             // A staging instructions per wave (8 rows of 128 B each)
  constexpr int BJ = BN / 32;             // B staging instructions per wave
  constexpr int TM = BM / WM / 32;        // 32-pixel MFMA tiles per wave
  constexpr int TN = BN / WN / 32;        // 32-channel MFMA tiles per wave
  constexpr int A_ELEMS = BM * 64, B_ELEMS = BN * 64;
  constexpr int RING_ELEMS = NS * (A_ELEMS + B_ELEMS);
  static_assert(RING_ELEMS * 2 >= TM * TN * 16 * 256 * 4, "group hand-over does not fit a ring");
  static_assert(RING_ELEMS * 2 >= (BM / WM) * (BN + 4) * 4, "staged epilogue does not fit a ring");
  __shared__ __attribute__((aligned(1024))) bf16_t smem_all[2 * RING_ELEMS];
  constexpr int LOADS = AJ + BJ;   // LDS-DMA instructions per thread per stage
  constexpr int D = NS - 1;        // prefetch distance

  // kg: K group of this wave (wave-uniform by construction; readfirstlane makes that provable)
  const int kg = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  bf16_t* smem = smem_all + kg * RING_ELEMS;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = (int)fdiv((uint32_t)wg, a.dNt);
  const int nt = wg - mt * a.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- phase geometry (wave-uniform) ----
  const int phase = blockIdx.y;
  int r0 = 0, s0 = 0, nr = a.kh, ns = a.kw, bh = -a.pt, bw = -a.pl, ph = 0, pw = 0;
  if (a.U == 2) {
    ph = phase >> 1;
    pw = phase & 1;
    r0 = (a.pt + ph) & 1;
    s0 = (a.pl + pw) & 1;
    nr = (a.kh - r0 + 1) >> 1;
    ns = (a.kw - s0 + 1) >> 1;
    bh = (ph - a.pt + r0) >> 1;  // exact: the numerator is even
    bw = (pw - a.pl + s0) >> 1;
  }
  const int nk_all = nr * ns * a.cblocks;
  const int nk = (nk_all - kg + 1) >> 1;     // this group's slices: kg, kg + 2, ...
  const int nk_max = (nk_all + 1) >> 1;      // loop trips (group 0's count): barriers are block-wide

  // ---- per-thread staging descriptors ----
  // instruction group g = wave*AJ + j stages rows g*8 .. g*8+7; lane -> row g*8 + (lane >> 3),
  // LDS slot (lane & 7) which must hold source chunk (lane & 7) ^ ((row >> 1) & 7)
  int a_off[AJ], a_ih[AJ], a_iw[AJ], a_c8[AJ];
  bool a_ok[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int g = wave * AJ + j;
    const int row = g * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    a_c8[j] = c * 8;
    const int m = m0 + row;
    a_ok[j] = m < a.Mp;
    const uint32_t mm = a_ok[j] ? (uint32_t)m : 0u;
    const uint32_t t1 = fdiv(mm, a.dWp);
    const int owp = (int)(mm - t1 * a.Wp);
    const uint32_t n = fdiv(t1, a.dHp);
    const int ohp = (int)(t1 - n * a.Hp);
    a_ih[j] = (a.U == 2) ? ohp + bh : ohp * a.S + bh;
    a_iw[j] = (a.U == 2) ? owp + bw : owp * a.S + bw;
    a_off[j] = (((int)n * a.Hin + a_ih[j]) * a.Win + a_iw[j]) * a.in_ld + c * 8;
  }
  int b_off[BJ], b_c8[BJ];
  bool b_ok[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int g = wave * BJ + j;
    const int row = g * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    b_c8[j] = c * 8;
    b_ok[j] = (n0 + row) < a.Co;
    b_off[j] = (n0 + row) * a.Kp + c * 8;
  }

  auto Abuf = [&](int buf) { return smem + buf * (A_ELEMS + B_ELEMS); };
  auto Bbuf = [&](int buf) { return smem + buf * (A_ELEMS + B_ELEMS) + A_ELEMS; };
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero16);

  // tap iteration state of the NEXT slice to stage (all scalar)
  int st_ri = 0, st_si = 0, st_cb = 0;
  auto advance = [&]() {
    if (++st_cb == a.cblocks) {
      st_cb = 0;
      if (++st_si == ns) {
        st_si = 0;
        ++st_ri;
      }
    }
  };
  auto stage = [&](int buf) {
    const int tapoff = (st_ri * a.Win + st_si) * a.in_ld + st_cb * 64;
    const int koff = ((r0 + a.U * st_ri) * a.kw + (s0 + a.U * st_si)) * a.Ci + st_cb * 64;
    bf16_t* Ab = Abuf(buf) + (wave * AJ) * 512;
    bf16_t* Bb = Bbuf(buf) + (wave * BJ) * 512;
    const int crem = a.Ci - st_cb * 64;   // channels left in this block (< 64 only for the last
                                          // block of a channel count that is not a multiple of 64)
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const bool ok = a_ok[j] && a_c8[j] < crem &&
                      (unsigned)(a_ih[j] + st_ri) < (unsigned)a.Hin &&
                      (unsigned)(a_iw[j] + st_si) < (unsigned)a.Win;
      const bf16_t* p = ok ? a.in + (int64_t)(a_off[j] + tapoff) : zero;
      glds16(p, Ab + j * 512);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const bf16_t* p = (b_ok[j] && b_c8[j] < crem) ? a.bt + (int64_t)(b_off[j] + koff) : zero;
      glds16(p, Bb + j * 512);
    }
    advance();
    advance();
  };
  if (kg == 1) advance();

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  // fragment addressing: row = base + (lane & 31), chunk = kk*2 + (lane >> 5), swizzled
  const int frow = lane & 31;
  const int swz = (frow >> 1) & 7;
  int koffs[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) koffs[kk] = (((kk * 2 + (lane >> 5)) ^ swz) << 3);
  const int arow0 = (wm * (BM / WM) + frow) * 64;
  const int brow0 = (wn * (BN / WN) + frow) * 64;

  // (see fast_conv_kernel: the empty upper half of a ragged last channel block is skipped)
  auto mma2 = [&](const bf16_t* Ab, const bf16_t* Bb, int k0) {
#pragma unroll
    for (int kk = k0; kk < k0 + 2; ++kk) {
      bf16x8_t af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        af[i] = *reinterpret_cast<const bf16x8_t*>(Ab + arow0 + i * 32 * 64 + koffs[kk]);
        if (RELU) af[i] = relu_bf16x8(af[i]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8_t*>(Bb + brow0 + j * 32 * 64 + koffs[kk]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  };
  const bool ragged_k = (a.Ci & 63) != 0;
  int c_cb = kg;   // channel block of this group's slice being consumed (slices kg, kg + 2, ...)
  while (c_cb >= a.cblocks) c_cb -= a.cblocks;

  int staged = 0;
  for (; staged < D && staged < nk; ++staged) stage(staged % NS);

  for (int it = 0; it < nk_max; ++it) {
    // wait for this group's slice `it` (its oldest in flight), then the block-wide barrier: every
    // wave's share of it has landed and every wave has finished reading the buffer restaged below
    const int ahead = staged - it - 1;
    if (D >= 3 && ahead >= 2) wait_vmcnt<2 * LOADS>();
    else if (D >= 2 && ahead == 1) wait_vmcnt<LOADS>();
    else wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");
    if (staged < nk) {
      stage(staged % NS);
      ++staged;
    }
    if (it >= nk) continue;   // group 1 has one slice less when the slice count is odd
    const int buf = it % NS;
    const bf16_t* Ab = Abuf(buf);
    const bf16_t* Bb = Bbuf(buf);
    mma2(Ab, Bb, 0);
    if (!(ragged_k && c_cb == a.cblocks - 1)) mma2(Ab, Bb, 2);
    c_cb += 2;                                  // this group's next slice is two further on
    while (c_cb >= a.cblocks) c_cb -= a.cblocks;
  }

  // ---- group 1 -> group 0: accumulators through group 1's ring, one float per thread per register
  {
    __syncthreads();   // every ring read is done
    float* red = reinterpret_cast<float*>(smem_all + RING_ELEMS);
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int v = 0; v < 16; ++v) red[((i * TN + j) * 16 + v) * 256 + tid] = acc[i][j][v];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[i][j][v] += red[((i * TN + j) * 16 + v) * 256 + tid];
    }
  }

  // ---- epilogue, coalesced form (Co % 8 == 0): the accumulators (+ bias, self-activation) go
  // through LDS in WM passes of BM/WM rows, then every thread finishes 8 consecutive channels of
  // one pixel: gate / residual are read and the result is written as 16-byte (bf16) or 2 x 16-byte
  // (fp32) pieces, whole rows contiguous across the lanes.  (The direct form below issues 8-byte
  // stores 256 B apart; measured, the output write was ~20 us of a 67 us launch.)
  if ((a.Co & 7) == 0) {
    constexpr int RP = BM / WM;          // rows per pass
    constexpr int LDC = BN + 4;          // floats per LDS row (+16 B: conflict-free b128 writes)
    constexpr int C8 = BN / 8;           // 8-channel items per row
    float* Cs = reinterpret_cast<float*>(smem);
    for (int h = 0; h < WM; ++h) {
      __syncthreads();
      if (kg == 0 && wm == h) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int col = wn * (BN / WN) + j * 32 + q * 8 + 4 * (lane >> 5);
              float4 v = make_float4(acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1],
                                     acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]);
              if (a.bias && n0 + col < a.Co) {
                const float4 b4 = *reinterpret_cast<const float4*>(a.bias + n0 + col);
                v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
              }
              if (a.self_gate && n0 + col < a.gate_cols) {
                if (!(v.x > 0.f)) v.x *= a.slope_out;
                if (!(v.y > 0.f)) v.y *= a.slope_out;
                if (!(v.z > 0.f)) v.z *= a.slope_out;
                if (!(v.w > 0.f)) v.w *= a.slope_out;
              }
              *reinterpret_cast<float4*>(Cs + (i * 32 + frow) * LDC + col) = v;
            }
      }
      __syncthreads();
      if (kg == 0)
      for (int t = tid; t < RP * C8; t += 256) {
        const int row = t / C8, c8 = t - row * C8;
        const int m = m0 + h * RP + row;
        const int co = n0 + c8 * 8;
        if (m >= a.Mp || co >= a.Co) continue;
        const uint32_t t1 = fdiv((uint32_t)m, a.dWp);
        const int owp = m - (int)t1 * a.Wp;
        const uint32_t n = fdiv(t1, a.dHp);
        const int ohp = (int)t1 - (int)n * a.Hp;
        const int64_t o =
            ((int64_t)((int)n * a.Ho + ohp * a.U + ph) * a.Wo + (owp * a.U + pw)) * a.out_ld + co;
        const float4 lo = *reinterpret_cast<const float4*>(Cs + row * LDC + c8 * 8);
        const float4 hi = *reinterpret_cast<const float4*>(Cs + row * LDC + c8 * 8 + 4);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (a.gate_out) {
          union { uint4 q; bf16_t h8[8]; } g;
          g.q = *reinterpret_cast<const uint4*>(a.gate_out + o);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (!(bf2f(g.h8[e]) > 0.f)) v[e] *= a.slope_out;
        }
        if (a.residual) {
          union { uint4 q; bf16_t h8[8]; } r;
          r.q = *reinterpret_cast<const uint4*>(a.residual + o);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bf2f(r.h8[e]);
        }
        if (a.out_f32) {
          float* op = reinterpret_cast<float*>(a.out) + o;
          *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          union { uint4 q; bf16_t h8[8]; } w;
#pragma unroll
          for (int e = 0; e < 8; ++e) w.h8[e] = f2bf(v[e]);
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.out) + o) = w.q;
        }
      }
    }
    return;
  }

  if (kg != 0) return;   // (no barriers below)
  // ---- epilogue: lane owns pixel (lane & 31) of each M sub-tile and, per 8-channel group q, the
  // 4 consecutive channels 8*q + 4*(lane >> 5) + {0..3}
  const bool vec4 = (a.Co & 3) == 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * (BM / WM) + i * 32 + frow;
    if (m >= a.Mp) continue;
    const uint32_t t1 = fdiv((uint32_t)m, a.dWp);
    const int owp = m - (int)t1 * a.Wp;
    const uint32_t n = fdiv(t1, a.dHp);
    const int ohp = (int)t1 - (int)n * a.Hp;
    const int64_t opix =
        ((int64_t)((int)n * a.Ho + ohp * a.U + ph) * a.Wo + (owp * a.U + pw)) * a.out_ld;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = n0 + wn * (BN / WN) + j * 32 + q * 8 + 4 * (lane >> 5);
        if (co >= a.Co) continue;
        float v[4] = {acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2],
                      acc[i][j][q * 4 + 3]};
        const int64_t o = opix + co;
        if (vec4) {
          if (a.bias) {
            const float4 b4 = *reinterpret_cast<const float4*>(a.bias + co);
            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
          }
          if (a.self_gate) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (!(v[e] > 0.f)) v[e] *= a.slope_out;
          } else if (a.gate_out) {
            const uint2 g2 = *reinterpret_cast<const uint2*>(a.gate_out + o);
            if (!(bf2f((bf16_t)(g2.x & 0xffff)) > 0.f)) v[0] *= a.slope_out;
            if (!(bf2f((bf16_t)(g2.x >> 16)) > 0.f)) v[1] *= a.slope_out;
            if (!(bf2f((bf16_t)(g2.y & 0xffff)) > 0.f)) v[2] *= a.slope_out;
            if (!(bf2f((bf16_t)(g2.y >> 16)) > 0.f)) v[3] *= a.slope_out;
          }
          if (a.residual) {
            const uint2 r2 = *reinterpret_cast<const uint2*>(a.residual + o);
            v[0] += bf2f((bf16_t)(r2.x & 0xffff));
            v[1] += bf2f((bf16_t)(r2.x >> 16));
            v[2] += bf2f((bf16_t)(r2.y & 0xffff));
            v[3] += bf2f((bf16_t)(r2.y >> 16));
          }
          if (a.out_f32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + o) =
                make_float4(v[0], v[1], v[2], v[3]);
          } else {
            uint2 w2;
            w2.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
            w2.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.out) + o) = w2;
          }
        } else {
          // channel counts that are not a multiple of 4 (RGB outputs): element-wise tail
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (co + e >= a.Co) continue;
            float val = v[e] + (a.bias ? a.bias[co + e] : 0.f);
            if (a.self_gate) {
              if (!(val > 0.f)) val *= a.slope_out;
            } else if (a.gate_out && !(bf2f(a.gate_out[o + e]) > 0.f)) {
              val *= a.slope_out;
            }
            if (a.residual) val += bf2f(a.residual[o + e]);
            if (a.out_f32)
              reinterpret_cast<float*>(a.out)[o + e] = val;
            else
              reinterpret_cast<bf16_t*>(a.out)[o + e] = f2bf(val);
          }
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------
// "halo" convolution: stride-1 filters with several taps (3x3, and the 2x2 / 3x3 phase filters of
// zero-inserted inputs).  The plain implicit GEMM above re-stages every input pixel once per tap; its
// speed is set by the global -> LDS fill rate (measured: ~8 TB/s chip-wide at 64 FLOP per staged
// byte).  Here a workgroup stages the input window of its output tile ONCE per 64-channel block --
// TH x TW output pixels need (TH + nr - 1) x (TW + ns - 1) input pixels -- and every tap reads its
// shifted view of that LDS image (fragment address = pixel row + tap shift), so only the weights
// are staged per tap: ~1.8x fewer staged bytes per FLOP for a 128-pixel tile.
// -------------------------------------------------------------------------------------------
struct HaloArgs {
  const bf16_t* in;
  const bf16_t* bt;
  void* out;
  const float* bias;
  const bf16_t* gate_out;
  const bf16_t* residual;
  int N, Hin, Win, Ci, Ho, Wo, Co, kh, kw, U, pt, pl;
  int Kp, cblocks;
  int Hp, Wp;                 // per-phase output grid
  int tw_log, th_log;         // tile = 2^th_log rows x 2^tw_log cols of NI images, 128 pixels
  int NI;
  int tiles_x, tiles_y, img_groups, ntiles;
  int halo_groups;            // LDS-DMA instruction groups (8 rows) of the largest phase's halo
  int relu_in, out_f32, self_gate;
  float slope_out;
  FastDiv dNt, dTx, dTy;
};

constexpr int HALO_SLOTS = 9;   // halo staging instructions per thread (covers 288 rows)

template <bool RELU>
__global__ __launch_bounds__(256) void halo_conv_kernel(HaloArgs a) {
  constexpr int BN = 128, B_ELEMS = BN * 64;
  extern __shared__ __attribute__((aligned(1024))) bf16_t hsm[];
  const int halo_elems = a.halo_groups * 512;          // per halo buffer
  bf16_t* Hbuf0 = hsm;
  bf16_t* Bbuf0 = hsm + 2 * halo_elems;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int st = (int)fdiv((uint32_t)wg, a.dNt);       // spatial tile
  const int nt = wg - st * a.ntiles;
  const int n0 = nt * BN;
  const int t1 = (int)fdiv((uint32_t)st, a.dTx);
  const int tx = st - t1 * a.tiles_x;
  const int ig = (int)fdiv((uint32_t)t1, a.dTy);
  const int ty = t1 - ig * a.tiles_y;
  const int TW = 1 << a.tw_log, TH = 1 << a.th_log;

  // ---- phase geometry (as fast_conv_kernel) ----
  const int phase = blockIdx.y;
  int r0 = 0, s0 = 0, nr = a.kh, ns = a.kw, bh = -a.pt, bw = -a.pl, ph = 0, pw = 0;
  if (a.U == 2) {
    ph = phase >> 1;
    pw = phase & 1;
    r0 = (a.pt + ph) & 1;
    s0 = (a.pl + pw) & 1;
    nr = (a.kh - r0 + 1) >> 1;
    ns = (a.kw - s0 + 1) >> 1;
    bh = (ph - a.pt + r0) >> 1;
    bw = (pw - a.pl + s0) >> 1;
  }
  const int ntaps = nr * ns;
  const int nk = ntaps * a.cblocks;
  const int HH = TH + nr - 1, HW = TW + ns - 1;
  const int hrows = a.NI * HH * HW;                    // halo pixels of this phase
  const int groups = (hrows + 7) >> 3;                 // <= a.halo_groups
  const int gpt = (groups + 3) >> 2;                   // staging slots per thread (<= HALO_SLOTS)
  const int slice = ntaps > 0 ? (gpt + ntaps - 1) / ntaps : 0;   // slots issued per tap iteration

  // ---- halo staging descriptors: slot j of this thread = group j*4 + wave, row g*8 + (lane>>3)
  int h_off[HALO_SLOTS];
  bool h_ok[HALO_SLOTS];
  const int hhw = HH * HW;
#pragma unroll
  for (int j = 0; j < HALO_SLOTS; ++j) {
    const int g = j * 4 + wave;
    const int row = g * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    const int il = row / hhw;
    const int rem = row - il * hhw;
    const int hy = rem / HW, hx = rem - hy * HW;
    const int n = ig * a.NI + il;
    const int ih = ty * TH + bh + hy, iw = tx * TW + bw + hx;
    h_ok[j] = (j < gpt) && row < hrows && n < a.N && (unsigned)ih < (unsigned)a.Hin &&
              (unsigned)iw < (unsigned)a.Win;
    h_off[j] = ((n * a.Hin + ih) * a.Win + iw) * a.Ci + c * 8;
  }
  int b_off[4];
  bool b_ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = wave * 4 + j;
    const int row = g * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    b_ok[j] = (n0 + row) < a.Co;
    b_off[j] = (n0 + row) * a.Kp + c * 8;
  }
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero16);

  auto stage_halo = [&](int hb, int cb, int j0, int j1) {
    bf16_t* Hb = Hbuf0 + hb * halo_elems;
#pragma unroll
    for (int j = 0; j < HALO_SLOTS; ++j) {
      if (j >= j0 && j < j1) {
        const bf16_t* p = h_ok[j] ? a.in + (int64_t)(h_off[j] + cb * 64) : zero;
        glds16(p, Hb + (j * 4 + wave) * 512);
      }
    }
  };
  auto stage_b = [&](int bb, int cb, int tap) {
    const int ri = tap / ns, si = tap - ri * ns;
    const int koff = ((r0 + a.U * ri) * a.kw + (s0 + a.U * si)) * a.Ci + cb * 64;
    bf16_t* Bb = Bbuf0 + bb * B_ELEMS + (wave * 4) * 512;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16_t* p = b_ok[j] ? a.bt + (int64_t)(b_off[j] + koff) : zero;
      glds16(p, Bb + j * 512);
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  // fragment rows: pixel p of the tile -> halo row of its tap-(0,0) input pixel
  const int frow = lane & 31;
  int hr0[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = wm * 64 + i * 32 + frow;
    const int il = p >> (a.tw_log + a.th_log);
    const int y = (p >> a.tw_log) & (TH - 1), x = p & (TW - 1);
    hr0[i] = il * hhw + y * HW + x;
  }
  const int half = lane >> 5;
  const int bswz = (frow >> 1) & 7;
  const int brow0 = (wn * 64 + frow) * 64;

  if (nk > 0) {
    stage_halo(0, 0, 0, gpt);
    stage_b(0, 0, 0);
  }
  int cb = 0, tap = 0;
  for (int it = 0; it < nk; ++it) {
    wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");
    // next iteration's weights, and this tap's share of the next channel block's halo
    int ncb = cb, ntap = tap + 1;
    if (ntap == ntaps) {
      ntap = 0;
      ncb = cb + 1;
    }
    if (it + 1 < nk) stage_b((it + 1) & 1, ncb, ntap);
    if (cb + 1 < a.cblocks) stage_halo((cb + 1) & 1, cb + 1, tap * slice, min(gpt, (tap + 1) * slice));
    const bf16_t* Hb = Hbuf0 + (cb & 1) * halo_elems;
    const bf16_t* Bb = Bbuf0 + (it & 1) * B_ELEMS;
    const int ri = tap / ns, si = tap - ri * ns;
    const int tshift = ri * HW + si;
    int arow[2], aswz[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int hr = hr0[i] + tshift;
      arow[i] = hr * 64;
      aswz[i] = (hr >> 1) & 7;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int ch = kk * 2 + half;
      bf16x8_t af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const bf16x8_t*>(Hb + arow[i] + ((ch ^ aswz[i]) << 3));
        if (RELU) af[i] = relu_bf16x8(af[i]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8_t*>(Bb + brow0 + j * 32 * 64 + ((ch ^ bswz) << 3));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    cb = ncb;
    tap = ntap;
  }

  // ---- epilogue ----
  const bool vec4 = (a.Co & 3) == 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = wm * 64 + i * 32 + frow;
    const int il = p >> (a.tw_log + a.th_log);
    const int y = (p >> a.tw_log) & (TH - 1), x = p & (TW - 1);
    const int n = ig * a.NI + il;
    const int ohp = ty * TH + y, owp = tx * TW + x;
    if (n >= a.N || ohp >= a.Hp || owp >= a.Wp) continue;
    const int64_t opix = ((int64_t)(n * a.Ho + ohp * a.U + ph) * a.Wo + (owp * a.U + pw)) * a.Co;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = n0 + wn * 64 + j * 32 + q * 8 + 4 * (lane >> 5);
        if (co >= a.Co) continue;
        float v[4] = {acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2],
                      acc[i][j][q * 4 + 3]};
        const int64_t o = opix + co;
        if (vec4) {
          if (a.bias) {
            const float4 b4 = *reinterpret_cast<const float4*>(a.bias + co);
            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
          }
          if (a.self_gate) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (!(v[e] > 0.f)) v[e] *= a.slope_out;
          } else if (a.gate_out) {
            const uint2 g2 = *reinterpret_cast<const uint2*>(a.gate_out + o);
            if (!(bf2f((bf16_t)(g2.x & 0xffff)) > 0.f)) v[0] *= a.slope_out;
            if (!(bf2f((bf16_t)(g2.x >> 16)) > 0.f)) v[1] *= a.slope_out;
            if (!(bf2f((bf16_t)(g2.y & 0xffff)) > 0.f)) v[2] *= a.slope_out;
            if (!(bf2f((bf16_t)(g2.y >> 16)) > 0.f)) v[3] *= a.slope_out;
          }
          if (a.residual) {
            const uint2 r2 = *reinterpret_cast<const uint2*>(a.residual + o);
            v[0] += bf2f((bf16_t)(r2.x & 0xffff));
            v[1] += bf2f((bf16_t)(r2.x >> 16));
            v[2] += bf2f((bf16_t)(r2.y & 0xffff));
            v[3] += bf2f((bf16_t)(r2.y >> 16));
          }
          if (a.out_f32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + o) =
                make_float4(v[0], v[1], v[2], v[3]);
          } else {
            uint2 w2;
            w2.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
            w2.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.out) + o) = w2;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (co + e >= a.Co) continue;
            float val = v[e] + (a.bias ? a.bias[co + e] : 0.f);
            if (a.self_gate) {
              if (!(val > 0.f)) val *= a.slope_out;
            } else if (a.gate_out && !(bf2f(a.gate_out[o + e]) > 0.f)) {
              val *= a.slope_out;
            }
            if (a.residual) val += bf2f(a.residual[o + e]);
            if (a.out_f32)
              reinterpret_cast<float*>(a.out)[o + e] = val;
            else
              reinterpret_cast<bf16_t*>(a.out)[o + e] = f2bf(val);
          }
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------
// weight gradient
// -------------------------------------------------------------------------------------------
struct FastWgradArgs {
  const bf16_t* in;
  const bf16_t* dy;
  float* out;       // dw (splits == 1) or partials [splits][K*Co]
  float* bias_out;  // NULL, dbias (splits == 1) or partials [splits][Co]
  int N, Hin, Win, Ci, Ho, Wo, Co, kh, kw, S, U, pt, pl;
  int Hp, Wp, Mp;   // per-phase output grid
  int K;            // kh*kw*Ci
  int cblocks;      // Ci / 128
  int ktiles, ntiles;
  int rows_per_split;  // multiple of 64
  int relu_in, accumulate;
  FastDiv dWp, dHp, dCb, dNt;
};

// tile: TKC (k: one tap x TKC channels, TKC = 128 or 64) x 128 (co), reduction over pixels in
// slices of 64 rows.  LDS rows are TKC*2 bytes (x) / 256 B (dy); chunk c (16 B) of row r is stored at
// slot c ^ ((r & 3) << 1), which spreads the 4 rows of every transpose-read block over 4 distinct
// 32-byte bank windows.
template <int TKC, bool RELU, bool ASMDMA>
__global__ __launch_bounds__(256) void fast_wgrad_kernel(FastWgradArgs a) {
  constexpr int MR = 64;
  constexpr int X_ELEMS = MR * TKC, Y_ELEMS = MR * 128;
  constexpr int RX = 512 / TKC;        // x rows per 1 KiB staging instruction (4 or 8)
  constexpr int XJ = MR / RX / 4;      // x staging instructions per wave (4 or 2)
  constexpr int LPR = 64 / RX;         // lanes (16-byte chunks) per x row (16 or 8)
  constexpr int FK = TKC / 64;         // 32-wide k tiles per wave (2 or 1)
  __shared__ __attribute__((aligned(1024))) bf16_t smem[2 * (X_ELEMS + Y_ELEMS)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave >> 1, wn = wave & 1;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int kt = (int)fdiv((uint32_t)wg, a.dNt);
  const int nt = wg - kt * a.ntiles;
  const int c0 = nt * 128;
  // k tile -> (tap, channel block)
  const int tap = (int)fdiv((uint32_t)kt, a.dCb);
  const int cb = kt - tap * a.cblocks;
  const int r = tap / a.kw, s = tap - r * a.kw;
  int ph = 0, pw = 0, bh, bw;
  if (a.U == 2) {
    ph = (a.pt + r) & 1;
    pw = (a.pl + s) & 1;
    bh = (ph - a.pt + r) >> 1;
    bw = (pw - a.pl + s) >> 1;
  } else {
    bh = r - a.pt;
    bw = s - a.pl;
  }
  const int mbeg = blockIdx.y * a.rows_per_split;
  const int mend = min(a.Mp, mbeg + a.rows_per_split);
  const int nit = (mend - mbeg + MR - 1) / MR;

  // staging.  dy: instruction group g = wave*4 + j covers rows g*4 .. g*4+3 (256 B each), lane ->
  // row g*4 + (lane >> 4), slot lane & 15 <- source chunk (lane & 15) ^ ((row & 3) << 1).
  // x: rows of TKC*2 bytes, RX rows per instruction, same swizzle rule.
  const int yrow = lane >> 4;
  const int ychunk = (lane & 15) ^ (yrow << 1);
  const bool y_ok = (c0 + ychunk * 8) < a.Co;  // Co % 8 == 0
  const int xrow = lane / LPR;
  const int xchunk = (lane % LPR) ^ ((xrow & 3) << 1);
  auto Xbuf = [&](int buf) { return smem + buf * (X_ELEMS + Y_ELEMS); };
  auto Ybuf = [&](int buf) { return smem + buf * (X_ELEMS + Y_ELEMS) + X_ELEMS; };
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero16);

  auto stage = [&](int buf, int it) {
    const int mb = mbeg + it * MR;
    bf16_t* Xb = Xbuf(buf) + (wave * XJ) * 512;
    bf16_t* Yb = Ybuf(buf) + (wave * 4) * 512;
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int m = mb + (wave * XJ + j) * RX + xrow;
      const bool mok = m < mend;
      const uint32_t mm = mok ? (uint32_t)m : 0u;
      const uint32_t t1 = fdiv(mm, a.dWp);
      const int owp = (int)(mm - t1 * a.Wp);
      const uint32_t n = fdiv(t1, a.dHp);
      const int ohp = (int)(t1 - n * a.Hp);
      const int ih = (a.U == 2) ? ohp + bh : ohp * a.S + bh;
      const int iw = (a.U == 2) ? owp + bw : owp * a.S + bw;
      const bool xok = mok && (cb * TKC + xchunk * 8) < a.Ci && (unsigned)ih < (unsigned)a.Hin &&
                       (unsigned)iw < (unsigned)a.Win;
      const int64_t xoff =
          ((int64_t)((int)n * a.Hin + ih) * a.Win + iw) * a.Ci + cb * TKC + xchunk * 8;
      glds16x<ASMDMA>(xok ? a.in + xoff : zero, Xb + j * 512);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = mb + (wave * 4 + j) * 4 + yrow;
      const bool mok = m < mend;
      const uint32_t mm = mok ? (uint32_t)m : 0u;
      const uint32_t t1 = fdiv(mm, a.dWp);
      const int owp = (int)(mm - t1 * a.Wp);
      const uint32_t n = fdiv(t1, a.dHp);
      const int ohp = (int)(t1 - n * a.Hp);
      const int64_t yoff =
          ((int64_t)((int)n * a.Ho + ohp * a.U + ph) * a.Wo + (owp * a.U + pw)) * a.Co + c0 +
          ychunk * 8;
      glds16x<ASMDMA>((mok && y_ok) ? a.dy + yoff : zero, Yb + j * 512);
    }
  };

  f32x16_t acc[FK][2];
#pragma unroll
  for (int i = 0; i < FK; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
  float bias_acc = 0.f;
  const bool do_bias = (a.bias_out != nullptr) && (kt == 0);

  // transpose-read addressing.  MFMA operand element (col = lane & 31, k = (lane >> 5)*8 + e):
  // a 16-lane group reads a [4 rows][16 cols] block, lane supplying row (l16 >> 2), 4 consecutive
  // columns (l16 & 3)*4 and receiving column l16 of the 4 rows.
  const int l16 = lane & 15;
  const int trow = l16 >> 2;                                 // row within the 4-row block
  const int tcol = ((lane >> 4) & 1) * 16 + (l16 & 3) * 4;   // element column within a 32-col tile
  const int mrow0 = (lane >> 5) * 8 + trow;                  // + mm*16 (+4 for the second read)
  auto tr_addr = [&](const bf16_t* base, int ld, int row, int col) {
    // row-major [64][ld] image with the 16-byte chunk swizzle
    const int chunk = (col >> 3) ^ ((row & 3) << 1);
    return base + row * ld + chunk * 8 + (col & 7);
  };

  if (nit > 0) {
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  for (int it = 0; it < nit; ++it) {
    const int buf = it & 1;
    if (it + 1 < nit) stage(buf ^ 1, it + 1);
    const bf16_t* Xb = Xbuf(buf);
    const bf16_t* Yb = Ybuf(buf);
#pragma unroll
    for (int mm = 0; mm < MR / 16; ++mm) {
      bf16x8_t xf[FK], yf[2];
      const int row = mm * 16 + mrow0;
#pragma unroll
      for (int i = 0; i < FK; ++i) {
        const int col = wk * (TKC / 2) + i * 32 + tcol;
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)tr_addr(Xb, TKC, row, col));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)tr_addr(Xb, TKC, row + 4, col));
        s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        xf[i] = __builtin_bit_cast(bf16x8_t, v);
        if (RELU) xf[i] = relu_bf16x8(xf[i]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = wn * 64 + j * 32 + tcol;
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)tr_addr(Yb, 128, row, col));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)tr_addr(Yb, 128, row + 4, col));
        s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        yf[j] = __builtin_bit_cast(bf16x8_t, v);
      }
#pragma unroll
      for (int i = 0; i < FK; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[i], yf[j], acc[i][j], 0, 0, 0);
    }
    if (do_bias && tid < 128) {
      // column sums of the dy tile (bias gradient); slot of (row, chunk c) is c ^ ((row & 3) << 1)
#pragma unroll 8
      for (int rr = 0; rr < MR; ++rr) {
        const int chunk = (tid >> 3) ^ ((rr & 3) << 1);
        bias_acc += bf2f(Yb[rr * 128 + chunk * 8 + (tid & 7)]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  const bool direct = (gridDim.y == 1);
  float* outp = a.out + (direct ? 0 : (int64_t)blockIdx.y * a.K * a.Co);
  const int kbase = tap * a.Ci + cb * TKC;
#pragma unroll
  for (int i = 0; i < FK; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int co = c0 + wn * 64 + j * 32 + (lane & 31);
      if (co >= a.Co) continue;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int kl = wk * (TKC / 2) + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
        if (cb * TKC + kl >= a.Ci) continue;   // padding rows of a partial channel block
        const int k = kbase + kl;
        const int64_t o = (int64_t)k * a.Co + co;
        if (direct && a.accumulate)
          outp[o] += acc[i][j][v];
        else
          outp[o] = acc[i][j][v];
      }
    }
  }
  if (do_bias && tid < 128 && c0 + tid < a.Co) {
    float* bp = a.bias_out + (direct ? 0 : (int64_t)blockIdx.y * a.Co);
    if (direct && a.accumulate)
      bp[c0 + tid] += bias_acc;
    else
      bp[c0 + tid] = bias_acc;
  }
}

// -------------------------------------------------------------------------------------------
// "stem" convolutions: image-like inputs (Ci <= 4, K = kh*kw*Ci <= 128).  Almost no FLOPs, HBM-bound
// on the activation they write (forward) / read (weight gradient).  The im2col tile is built in LDS
// by a scalar gather of the tiny, cache-resident input; the contraction still runs on MFMA.
// -------------------------------------------------------------------------------------------
struct StemArgs {
  const bf16_t* in;
  const bf16_t* bt;   // forward: [Co][Kp] bf16
  const bf16_t* dy;   // weight gradient: [M][Co] bf16
  void* out;          // forward: [M][Co] bf16/f32;  weight gradient: partials [splits][K*Co + Co]
  const float* bias;
  int N, Hin, Win, Ci, Ho, Wo, Co, kh, kw, S, pt, pl;
  int M, K, Kp, KS;   // KS = K rounded up to 32
  int relu_in, out_f32, want_bias;
  int self_gate;     // forward: y = lrelu_slope(conv + bias)
  float slope_out;
  int rows_per_split;
  int adjoint_out;   // weight gradient computed on the adjoint geometry: store as [kh,kw,Co,Ci]
                     // with flipped taps, i.e. directly in the forward conv's HWIO layout
  FastDiv dWo, dHo, dCi, dKw;
};

// per-k decode table (built once per block): r | s << 8 | valid << 16, element offset of the tap
struct StemTap {
  int rs;
  int off;
};
__device__ __forceinline__ void stem_build_taps(const StemArgs& a, StemTap* taps, int KS) {
  for (int k = threadIdx.x; k < KS; k += blockDim.x) {
    StemTap t;
    t.rs = 0;
    t.off = 0;
    if (k < a.K) {
      const uint32_t tap = fdiv((uint32_t)k, a.dCi);
      const int ci = k - (int)tap * a.Ci;
      const int r = (int)fdiv(tap, a.dKw);
      const int s = (int)tap - r * a.kw;
      t.rs = r | (s << 8) | (1 << 16);
      t.off = (r * a.Win + s) * a.Ci + ci;
    }
    taps[k] = t;
  }
}
// gathers element k of the im2col row whose tap (0,0) pixel is (ihb, iwb) at element offset base;
// zero outside the image / beyond K
__device__ __forceinline__ bf16_t stem_gather(const StemArgs& a, const StemTap* taps,
                                              int64_t base, int ihb, int iwb, int k, bool row_ok) {
  const StemTap t = taps[k];
  const int ih = ihb + (t.rs & 0xff), iw = iwb + ((t.rs >> 8) & 0xff);
  if (!row_ok || !(t.rs >> 16) || (unsigned)ih >= (unsigned)a.Hin ||
      (unsigned)iw >= (unsigned)a.Win)
    return 0;
  bf16_t v = a.in[base + t.off];
  if (a.relu_in && (v & 0x8000)) v = 0;
  return v;
}

// forward: block = 128 pixels x the 128 output channels from 128 * blockIdx.y (one such tile for the
// stems proper; the data gradient of a generator's RGB convolution, read as a 3 -> 256 channel
// convolution, takes two -- it ran the generic gather kernel at 16 TFLOP/s, 52 us per cifar step).
// LDS: A [128][KS], W [128][KS].
template <int KS>
__global__ __launch_bounds__(256) void stem_fwd_kernel(StemArgs a) {
  constexpr int LDA = KS + 8;  // +16 B row padding: conflict-free ds_read_b128 fragments
  __shared__ __attribute__((aligned(16))) bf16_t As[128 * LDA];
  __shared__ __attribute__((aligned(16))) bf16_t Ws[128 * LDA];
  __shared__ StemTap taps[KS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * 128;
  const int cb = blockIdx.y * 128;   // first output channel of this block
  stem_build_taps(a, taps, KS);
  __syncthreads();
  // weights -> LDS (zero rows / columns beyond Co / K)
  for (int i = tid; i < 128 * (KS / 8); i += 256) {
    const int row = i / (KS / 8), ch = i % (KS / 8);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (cb + row < a.Co && ch * 8 < a.Kp)
      v = *reinterpret_cast<const uint4*>(a.bt + (int64_t)(cb + row) * a.Kp + ch * 8);
    *reinterpret_cast<uint4*>(Ws + row * LDA + ch * 8) = v;
  }
  // im2col tile: thread -> pixel (tid & 127), k half (tid >> 7)
  {
    const int p = tid & 127, half = tid >> 7;
    const int m = m0 + p;
    const bool ok = m < a.M;
    const uint32_t mm = ok ? (uint32_t)m : 0u;
    const uint32_t t1 = fdiv(mm, a.dWo);
    const int ow = (int)(mm - t1 * a.Wo);
    const uint32_t n = fdiv(t1, a.dHo);
    const int oh = (int)(t1 - n * a.Ho);
    const int ihb = oh * a.S - a.pt, iwb = ow * a.S - a.pl;
    const int64_t base = ((int64_t)((int)n * a.Hin + ihb) * a.Win + iwb) * a.Ci;
#pragma unroll 8
    for (int kk = 0; kk < KS / 2; kk += 2) {
      const int k = half * (KS / 2) + kk;
      const uint32_t lo = stem_gather(a, taps, base, ihb, iwb, k, ok);
      const uint32_t hi = stem_gather(a, taps, base, ihb, iwb, k + 1, ok);
      *reinterpret_cast<uint32_t*>(As + p * LDA + k) = lo | (hi << 16);
    }
  }
  __syncthreads();
  const int frow = lane & 31;
  const int m = m0 + wave * 32 + frow;
  const int64_t opix = (int64_t)m * a.Co;
  const int nct = (min(a.Co - cb, 128) + 31) / 32;
  for (int ct = 0; ct < nct; ++ct) {
    f32x16_t acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KS / 16; ++kk) {
      const int ko = kk * 16 + (lane >> 5) * 8;
      const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(As + (wave * 32 + frow) * LDA + ko);
      const bf16x8_t bf = *reinterpret_cast<const bf16x8_t*>(Ws + (ct * 32 + frow) * LDA + ko);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, af, acc, 0, 0, 0);
    }
    if (m >= a.M) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = cb + ct * 32 + q * 8 + 4 * (lane >> 5);
      if (co >= a.Co) continue;
      if ((a.Co & 3) == 0) {
        float v[4] = {acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]};
        if (a.bias) {
          const float4 b4 = *reinterpret_cast<const float4*>(a.bias + co);
          v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        }
        if (a.self_gate) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (!(v[e] > 0.f)) v[e] *= a.slope_out;
        }
        if (a.out_f32) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + opix + co) =
              make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2 w2;
          w2.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
          w2.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.out) + opix + co) = w2;
        }
        continue;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (co + e >= a.Co) continue;
        float val = acc[q * 4 + e] + (a.bias ? a.bias[co + e] : 0.f);
        if (a.self_gate && !(val > 0.f)) val *= a.slope_out;
        if (a.out_f32)
          reinterpret_cast<float*>(a.out)[opix + co + e] = val;
        else
          reinterpret_cast<bf16_t*>(a.out)[opix + co + e] = f2bf(val);
      }
    }
  }
}

// weight gradient: grid (co tiles of 128, splits).  Per 64-pixel slice: im2col tile [64][KS] by
// gather, dy tile [64][128] by LDS-DMA (same image and swizzle as fast_wgrad_kernel), operands by
// transpose reads.  Waves split the 128 channels; every wave owns all KS/32 k tiles.
template <int KS>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(StemArgs a) {
  constexpr int MR = 64;
  constexpr int KT = KS / 32;
  __shared__ __attribute__((aligned(1024))) bf16_t Ys[2 * MR * 128];
  __shared__ __attribute__((aligned(16))) bf16_t Xs[2 * MR * KS];
  __shared__ StemTap taps[KS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * 128;
  stem_build_taps(a, taps, KS);
  __syncthreads();
  const int mbeg = blockIdx.y * a.rows_per_split;
  const int mend = min(a.M, mbeg + a.rows_per_split);
  const int nit = (mend - mbeg + MR - 1) / MR;
  const int yrow = lane >> 4;
  const int ychunk = (lane & 15) ^ (yrow << 1);
  const bool y_ok = (c0 + ychunk * 8) < a.Co;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero16);

  auto stage = [&](int buf, int it) {
    const int mb = mbeg + it * MR;
    bf16_t* Yb = Ys + buf * MR * 128 + (wave * 4) * 512;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = mb + (wave * 4 + j) * 4 + yrow;
      const bool mok = m < mend;
      const int64_t yoff = (int64_t)(mok ? m : 0) * a.Co + c0 + ychunk * 8;
      glds16((mok && y_ok) ? a.dy + yoff : zero, Yb + j * 512);
    }
    // im2col gather: thread -> pixel (tid & 63), k quarter (tid >> 6)
    const int p = tid & 63, quarter = tid >> 6;
    const int m = mb + p;
    const bool ok = m < mend;
    const uint32_t mm = ok ? (uint32_t)m : 0u;
    const uint32_t t1 = fdiv(mm, a.dWo);
    const int ow = (int)(mm - t1 * a.Wo);
    const uint32_t n = fdiv(t1, a.dHo);
    const int oh = (int)(t1 - n * a.Ho);
    const int ihb = oh * a.S - a.pt, iwb = ow * a.S - a.pl;
    const int64_t base = ((int64_t)((int)n * a.Hin + ihb) * a.Win + iwb) * a.Ci;
    bf16_t* Xb = Xs + buf * MR * KS;
#pragma unroll
    for (int kk = 0; kk < KS / 4; kk += 2) {
      const int k = quarter * (KS / 4) + kk;
      const uint32_t lo = stem_gather(a, taps, base, ihb, iwb, k, ok);
      const uint32_t hi = stem_gather(a, taps, base, ihb, iwb, k + 1, ok);
      *reinterpret_cast<uint32_t*>(Xb + p * KS + k) = lo | (hi << 16);
    }
  };

  f32x16_t acc[KT];
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
  float bias_acc = 0.f;

  const int l16 = lane & 15;
  const int trow = l16 >> 2;
  const int tcol = ((lane >> 4) & 1) * 16 + (l16 & 3) * 4;
  const int mrow0 = (lane >> 5) * 8 + trow;

  if (nit > 0) {
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  for (int it = 0; it < nit; ++it) {
    const int buf = it & 1;
    if (it + 1 < nit) stage(buf ^ 1, it + 1);
    const bf16_t* Xb = Xs + buf * MR * KS;
    const bf16_t* Yb = Ys + buf * MR * 128;
#pragma unroll
    for (int mm = 0; mm < MR / 16; ++mm) {
      const int row = mm * 16 + mrow0;
      bf16x8_t yf;
      {
        const int col = wave * 32 + tcol;
        const int ch0 = (col >> 3) ^ ((row & 3) << 1), ch1 = (col >> 3) ^ (((row + 4) & 3) << 1);
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(Yb + row * 128 + ch0 * 8 + (col & 7)));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(Yb + (row + 4) * 128 + ch1 * 8 + (col & 7)));
        s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        yf = __builtin_bit_cast(bf16x8_t, v);
      }
#pragma unroll
      for (int i = 0; i < KT; ++i) {
        const int col = i * 32 + tcol;
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(Xb + row * KS + col));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(Xb + (row + 4) * KS + col));
        s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, v), yf,
                                                         acc[i], 0, 0, 0);
      }
    }
    if (a.want_bias && tid < 128) {
#pragma unroll 8
      for (int rr = 0; rr < MR; ++rr) {
        const int chunk = (tid >> 3) ^ ((rr & 3) << 1);
        bias_acc += bf2f(Yb[rr * 128 + chunk * 8 + (tid & 7)]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  float* outp = reinterpret_cast<float*>(a.out) + (int64_t)blockIdx.y * ((int64_t)a.K * a.Co + a.Co);
  const int co = c0 + wave * 32 + (lane & 31);
  if (co < a.Co) {
#pragma unroll
    for (int i = 0; i < KT; ++i)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int k = i * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
        if (k >= a.K) continue;
        if (!a.adjoint_out) {
          outp[(int64_t)k * a.Co + co] = acc[i][v];
        } else {
          // k = (tap', c) of the adjoint conv (its Ci = the forward conv's Co); forward tap index
          // = T-1-tap' (both axes flipped); forward layout [tap][Ci_f = a.Co][Co_f = a.Ci]
          const uint32_t tapa = fdiv((uint32_t)k, a.dCi);
          const int c = k - (int)tapa * a.Ci;
          const int tapf = a.kh * a.kw - 1 - (int)tapa;
          outp[((int64_t)tapf * a.Co + co) * a.Ci + c] = acc[i][v];
        }
      }
  }
  if (a.want_bias && tid < 128 && c0 + tid < a.Co) outp[(int64_t)a.K * a.Co + c0 + tid] = bias_acc;
}

// -------------------------------------------------------------------------------------------
// weight gradient, "halo" form for unit-stride multi-tap filters (3x3): fast_wgrad_kernel handles one
// tap per workgroup, so the activations and the output gradients are re-read once per tap (9x).
// Here a workgroup owns a 64-channel x 64-out-channel block of ALL taps: per 64-pixel slice it
// stages the input window (with its halo) and the dy slice once, and every tap's operand is a
// shifted transpose-read of that window.  9 accumulator tiles per wave (144 registers).
// -------------------------------------------------------------------------------------------
struct HaloWgradArgs {
  const bf16_t* in;
  const bf16_t* dy;
  float* out;   // dw (splits == 1) or partials [splits][K*Co]
  float* bias;  // nullptr, dbias (splits == 1) or partials [splits][Co]
  int N, Hin, Win, Ci, Ho, Wo, Co, kh, kw, pt, pl;
  int K, cblocks, ntiles;
  int tw_log, th_log, NI, tiles_x, tiles_y, nslices, slices_per_split;
  int relu_in, accumulate;
  FastDiv dNt, dTx, dTy;
};
constexpr int HW_GROUPS = 20;   // halo staging groups (8 rows each) per buffer: up to 160 pixels
constexpr int HW_SLOTS = HW_GROUPS / 4;

template <bool RELU, bool W4, bool ASMDMA>
__global__ __launch_bounds__(256, 2) void halo_wgrad_kernel(HaloWgradArgs a) {
  __shared__ __attribute__((aligned(1024))) bf16_t smem[2 * (HW_GROUPS * 512 + 64 * 64)];
  auto Xh = [&](int buf) { return smem + buf * (HW_GROUPS * 512 + 64 * 64); };
  auto Ys = [&](int buf) { return smem + buf * (HW_GROUPS * 512 + 64 * 64) + HW_GROUPS * 512; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave >> 1, wn = wave & 1;
  const int cb = (int)fdiv((uint32_t)blockIdx.x, a.dNt);
  const int nt = blockIdx.x - cb * a.ntiles;
  const int c0 = nt * 64;
  const int TW = 1 << a.tw_log, TH = 1 << a.th_log;
  constexpr int ntaps = 9;   // 3x3 only: the tap loops must unroll
  const int HH = TH + 2, HW = TW + 2, hhw = HH * HW;
  const int hrows = a.NI * hhw;
  const int gpt = ((hrows + 7) / 8 + 3) / 4;   // <= HW_SLOTS
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero16);

  // LDS rows are 128 B (64 channels); 16-B chunk c of row r sits at chunk c ^ (((r >> 1) & 1) << 2),
  // i.e. byte ^ ((r & 2) << 5): any 4 consecutive rows of a transpose read then hit distinct banks,
  // whatever the parity of the first row (odd tap shifts), and rows r, r + 4 differ by 512 B exactly.
  // halo staging slots: row (j*4 + wave)*8 + (lane >> 3); descriptor packed
  // il | hy << 4 | hx << 10 | chunk << 16 | valid << 20
  int hdesc[HW_SLOTS];
#pragma unroll
  for (int j = 0; j < HW_SLOTS; ++j) {
    const int row = (j * 4 + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (((row >> 1) & 1) << 2);
    const int il = row / hhw;
    const int rem = row - il * hhw;
    const int hy = rem / HW, hx = rem - hy * HW;
    const int ok = j < gpt && row < hrows && (cb * 64 + c * 8) < a.Ci;
    hdesc[j] = il | (hy << 4) | (hx << 10) | (c << 16) | (ok << 20);
  }
  // dy staging: row (wave*2 + j)*8 + (lane >> 3) = pixel of the slice; il | y << 4 | x << 10 | chunk << 16
  int ydesc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = (wave * 2 + j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (((p >> 1) & 1) << 2);
    ydesc[j] = (p >> (a.tw_log + a.th_log)) | (((p >> a.tw_log) & (TH - 1)) << 4) |
               ((p & (TW - 1)) << 10) | (c << 16);
  }
  auto stage = [&](int buf, int sl) {
    // slice -> (image group, tile row, tile col): wave-uniform
    const int t1 = (int)fdiv((uint32_t)sl, a.dTx);
    const int tx = sl - t1 * a.tiles_x;
    const int ig = (int)fdiv((uint32_t)t1, a.dTy);
    const int ty = t1 - ig * a.tiles_y;
    bf16_t* Xb = Xh(buf);
    bf16_t* Yb = Ys(buf) + (wave * 2) * 512;
    const int n0 = ig * a.NI, ih0 = ty * TH - a.pt, iw0 = tx * TW - a.pl;
#pragma unroll
    for (int j = 0; j < HW_SLOTS; ++j) {
      if (j < gpt) {
        const int d = hdesc[j];
        const int n = n0 + (d & 15);
        const int ih = ih0 + ((d >> 4) & 63), iw = iw0 + ((d >> 10) & 63);
        const bool ok = (d >> 20) && n < a.N && (unsigned)ih < (unsigned)a.Hin &&
                        (unsigned)iw < (unsigned)a.Win;
        const int64_t off =
            ((int64_t)(n * a.Hin + ih) * a.Win + iw) * a.Ci + cb * 64 + ((d >> 16) & 7) * 8;
        glds16x<ASMDMA>(ok ? a.in + off : zero, Xb + (j * 4 + wave) * 512);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int d = ydesc[j];
      const int n = n0 + (d & 15);
      const int oh = ty * TH + ((d >> 4) & 63), ow = tx * TW + ((d >> 10) & 63);
      const int c8 = ((d >> 16) & 7) * 8;
      const bool ok = n < a.N && (c0 + c8) < a.Co;
      const int64_t off = ((int64_t)(n * a.Ho + oh) * a.Wo + ow) * a.Co + c0 + c8;
      glds16x<ASMDMA>(ok ? a.dy + off : zero, Yb + j * 512);
    }
  };

  f32x16_t acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
  // bias gradient = column sums of dy: one more MFMA per k-step with an all-ones A operand, in the
  // workgroups of channel block 0 only (wave-uniform)
  const bool want_bias = a.bias != nullptr && cb == 0;
  f32x16_t accb;
#pragma unroll
  for (int v = 0; v < 16; ++v) accb[v] = 0.f;
  const s16x8_t ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);

  // transpose-read addressing (see fast_wgrad_kernel): for k-step mm this lane supplies pixel
  // p = mm*16 + (lane >> 5)*8 + (l16 >> 2) (lo read) and p + 4 (hi read: the same tile row, +512 B,
  // when tiles are >= 8 pixels wide), 4 columns at tcol
  const int l16 = lane & 15;
  const int tcol = ((lane >> 4) & 1) * 16 + (l16 & 3) * 4;
  const int xcolb = (wk * 32 + tcol) * 2, ycolb = (wn * 32 + tcol) * 2;   // byte column
  int hr00[4];   // halo row of the tap-(0,0) input pixel of p
  int yoff[4];   // byte offset of the dy read
#pragma unroll
  for (int mm = 0; mm < 4; ++mm) {
    const int p = mm * 16 + (lane >> 5) * 8 + (l16 >> 2);
    const int il = p >> (a.tw_log + a.th_log);
    const int y = (p >> a.tw_log) & (TH - 1), x = p & (TW - 1);
    hr00[mm] = il * hhw + y * HW + x;
    yoff[mm] = ((p << 7) | ycolb) ^ ((p & 2) << 5);
  }
  typedef __attribute__((address_space(3))) s16x4_t* lds_tr_ptr;
  typedef __attribute__((address_space(3))) char* lds_char_ptr;

  const int sbeg = blockIdx.y * a.slices_per_split;
  const int send = min(a.nslices, sbeg + a.slices_per_split);
  if (sbeg < send) {
    stage(0, sbeg);
    wait_vmcnt<0>();
  }
  __syncthreads();
  for (int sl = sbeg; sl < send; ++sl) {
    const int buf = (sl - sbeg) & 1;
    if (sl + 1 < send) stage(buf ^ 1, sl + 1);
    lds_char_ptr Xb = (lds_char_ptr)Xh(buf);
    lds_char_ptr Yb = (lds_char_ptr)Ys(buf);
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) {
      bf16x8_t yf;
      {
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(Yb + yoff[mm]));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(Yb + yoff[mm] + 512));
        s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        yf = __builtin_bit_cast(bf16x8_t, v);
      }
      if (want_bias) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, yf, accb, 0, 0, 0);
      // keep the 9 x 4 tap addresses out of registers: recompute them from hr00 in every slice
      int hq = hr00[mm];
      asm volatile("" : "+v"(hq));
      bf16x8_t xf[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int h = hq + (t / 3) * HW + (t % 3);
        const int off = ((h << 7) | xcolb) ^ ((h & 2) << 5);
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(Xb + off));
        s16x4_t hi;
        if (W4) {
          // 4-pixel-wide tiles: pixel p + 4 is the next tile row, HW (not 4) window rows further
          const int h2 = h + HW;
          const int off2 = ((h2 << 7) | xcolb) ^ ((h2 & 2) << 5);
          hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(Xb + off2));
        } else {
          hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(Xb + off + 512));
        }
        s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        xf[t] = __builtin_bit_cast(bf16x8_t, v);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (RELU) xf[t] = relu_bf16x8(xf[t]);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[t], yf, acc[t], 0, 0, 0);
      }
    }
    wait_vmcnt<0>();
    __syncthreads();
  }

  const bool direct = (gridDim.y == 1);
  float* outp = a.out + (direct ? 0 : (int64_t)blockIdx.y * a.K * a.Co);
  const int co = c0 + wn * 32 + (lane & 31);
  if (co < a.Co) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int ch = cb * 64 + wk * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
        if (ch >= a.Ci) continue;
        const int64_t o = ((int64_t)t * a.Ci + ch) * a.Co + co;
        if (direct && a.accumulate)
          outp[o] += acc[t][v];
        else
          outp[o] = acc[t][v];
      }
    }
    // every row of accb holds the column sums; row 0 lives in accb[0] of lanes 0..31
    if (want_bias && wk == 0 && lane < 32) {
      float* bp = a.bias + (direct ? 0 : (int64_t)blockIdx.y * a.Co) + co;
      *bp = (direct && a.accumulate) ? *bp + accb[0] : accb[0];
    }
  }
}

// out[i] = (accumulate ? out[i] : 0) + sum_z part[z * stride + i]; block = 32 outputs x 32 split
// lanes, 4 loads in flight per lane (the walk over the partials is pure load latency)
constexpr int SR_ZL = 32;
__global__ __launch_bounds__(32 * SR_ZL) void split_reduce_strided_kernel(
    const float* __restrict__ part, int splits, int64_t stride, int64_t n, float* __restrict__ out,
    int accumulate) {
  __shared__ float sm[SR_ZL][33];
  const int il = threadIdx.x & 31, zl = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + il;
  float s = 0.f;
  if (i < n) {
#pragma unroll 4
    for (int z = zl; z < splits; z += SR_ZL) s += part[(int64_t)z * stride + i];
  }
  sm[zl][il] = s;
  __syncthreads();
  if (zl == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < SR_ZL; ++r) t += sm[r][il];
    out[i] = accumulate ? out[i] + t : t;
  }
}

// out[i] = (accumulate ? out[i] : 0) + sum_z part[z][i], 4 floats per thread
__global__ __launch_bounds__(256) void split_reduce4_kernel(const float* __restrict__ part,
                                                            int splits, int64_t n4,
                                                            float* __restrict__ out,
                                                            int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z = 0; z < splits; ++z) {
    const float4 p = reinterpret_cast<const float4*>(part)[(int64_t)z * n4 + i];
    s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
  }
  float4* o = reinterpret_cast<float4*>(out) + i;
  if (accumulate) {
    const float4 q = *o;
    s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
  }
  *o = s;
}

// same sum, 8 split lanes per output float4: the serial loop over the partials is 8x shorter and 4
// loads are in flight per thread (a single thread walking 128 partials is pure HBM latency)
__global__ __launch_bounds__(256) void split_reduce4x8_kernel(
    const float* __restrict__ part_a, int64_t n4_a, float* __restrict__ out_a, int blocks_a,
    const float* __restrict__ part_b, int64_t n4_b, float* __restrict__ out_b, int splits,
    int accumulate) {
  // two reductions in one launch (weight gradient + bias gradient): blocks >= blocks_a do the second
  __shared__ float4 sm[8][32];
  const bool second = (int)blockIdx.x >= blocks_a;
  const float* part = second ? part_b : part_a;
  const int64_t n4 = second ? n4_b : n4_a;
  float* out = second ? out_b : out_a;
  const int il = threadIdx.x & 31, zl = threadIdx.x >> 5;
  const int64_t i = (int64_t)(second ? blockIdx.x - blocks_a : blockIdx.x) * 32 + il;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float4* p4 = reinterpret_cast<const float4*>(part) + i;
#pragma unroll 4
    for (int z = zl; z < splits; z += 8) {
      const float4 p = p4[(int64_t)z * n4];
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
  }
  sm[zl][il] = s;
  __syncthreads();
  if (zl == 0 && i < n4) {
    float4 t = sm[0][il];
#pragma unroll
    for (int r = 1; r < 8; ++r) {
      const float4 q = sm[r][il];
      t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
    }
    float4* o = reinterpret_cast<float4*>(out) + i;
    if (accumulate) {
      const float4 q = *o;
      t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
    }
    *o = t;
  }
}
// ---- deferred reductions (cgDeferCtx, include/cgamd.h) -------------------------------------------
// A backward pass launches one small fixed-order reduction behind every weight-gradient kernel that
// splits its pixels (60 launches of 4-6 us per ResNet-CIFAR step).  Through cg_gwgrad_deferred /
// cg_gwgrad_pooled_deferred those reductions are only RECORDED in the caller's context -- same
// partial layout, same summation order per output -- and cg_defer_flush() runs all of them in one
// launch per kernel form.  The context belongs to the caller (one per stream / replica); the only
// library-side state is the thread-local pointer a ReduceDeferScope holds during one call.
}  // namespace
struct cgReduceJob {
  const float* part;
  float* out;
  int64_t n;        // outputs: float4 units (kinds 0, 1) or floats (kind 2)
  int64_t stride;   // kind 2: floats between consecutive partials
  int splits, accumulate;
  int kind;         // 0: split_reduce4x8 (8 split lanes), 1: split_reduce4 (serial), 2: strided
};
struct cgDeferCtx {
  std::mutex mu;    // (a context is meant for one thread at a time; the lock makes misuse benign)
  std::vector<cgReduceJob> jobs;
};
namespace {
typedef cgReduceJob ReduceJob;
thread_local cgDeferCtx* t_defer = nullptr;
thread_local int t_defer_suspend = 0;

static bool defer_reduce(const ReduceJob& j) {
  cgDeferCtx* c = t_defer;
  if (!c || t_defer_suspend > 0) return false;
  std::lock_guard<std::mutex> lk(c->mu);
  c->jobs.push_back(j);
  return true;
}

constexpr int RJ_MAX = 48;   // jobs per launch (their descriptors travel by value in the kernel arguments)
struct ReduceChunk {
  const float* part[RJ_MAX];
  float* out[RJ_MAX];
  int64_t n[RJ_MAX], stride[RJ_MAX];
  int splits[RJ_MAX], flags[RJ_MAX];   // flags: bit 0 accumulate, bit 1 serial form (kind 1)
  int blk[RJ_MAX + 1];
  int cnt;
};
__device__ __forceinline__ int rj_find(const int* blk, int n, int b) {
  int i = 0;
  while (i + 1 < n && blk[i + 1] <= b) ++i;
  return i;
}
// kinds 0 / 1 of many jobs in one launch; per output the arithmetic of split_reduce4x8_kernel /
// split_reduce4_kernel (bit-identical to the separate launches)
__global__ __launch_bounds__(256) void split_reduce4_multi_kernel(ReduceChunk c) {
  __shared__ float4 sm[8][32];
  const int it = rj_find(c.blk, c.cnt, blockIdx.x);
  const int b = blockIdx.x - c.blk[it];
  const float* __restrict__ part = c.part[it];
  float* __restrict__ out = c.out[it];
  const int64_t n4 = c.n[it];
  const int splits = c.splits[it], accumulate = c.flags[it] & 1;
  if (c.flags[it] & 2) {   // (block-uniform)
    const int64_t i = (int64_t)b * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < splits; ++z) {
      const float4 p = reinterpret_cast<const float4*>(part)[(int64_t)z * n4 + i];
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    float4* o = reinterpret_cast<float4*>(out) + i;
    if (accumulate) {
      const float4 q = *o;
      s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
    }
    *o = s;
    return;
  }
  const int il = threadIdx.x & 31, zl = threadIdx.x >> 5;
  const int64_t i = (int64_t)b * 32 + il;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float4* p4 = reinterpret_cast<const float4*>(part) + i;
#pragma unroll 4
    for (int z = zl; z < splits; z += 8) {
      const float4 p = p4[(int64_t)z * n4];
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
  }
  sm[zl][il] = s;
  __syncthreads();
  if (zl == 0 && i < n4) {
    float4 t = sm[0][il];
#pragma unroll
    for (int r = 1; r < 8; ++r) {
      const float4 q = sm[r][il];
      t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
    }
    float4* o = reinterpret_cast<float4*>(out) + i;
    if (accumulate) {
      const float4 q = *o;
      t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
    }
    *o = t;
  }
}
// kind 2 (split_reduce_strided_kernel's arithmetic)
__global__ __launch_bounds__(32 * SR_ZL) void split_reduce_strided_multi_kernel(ReduceChunk c) {
  __shared__ float sm[SR_ZL][33];
  const int it = rj_find(c.blk, c.cnt, blockIdx.x);
  const float* __restrict__ part = c.part[it];
  float* __restrict__ out = c.out[it];
  const int64_t n = c.n[it], stride = c.stride[it];
  const int splits = c.splits[it], accumulate = c.flags[it] & 1;
  const int il = threadIdx.x & 31, zl = threadIdx.x >> 5;
  const int64_t i = (int64_t)(blockIdx.x - c.blk[it]) * 32 + il;
  float s = 0.f;
  if (i < n) {
#pragma unroll 4
    for (int z = zl; z < splits; z += SR_ZL) s += part[(int64_t)z * stride + i];
  }
  sm[zl][il] = s;
  __syncthreads();
  if (zl == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < SR_ZL; ++r) t += sm[r][il];
    out[i] = accumulate ? out[i] + t : t;
  }
}

// weight-gradient partials (+ optionally the bias-gradient partials of the same splits) in one launch
static void launch_split_reduce4_pair(const float* part_a, int64_t n4_a, float* out_a,
                                      const float* part_b, int64_t n4_b, float* out_b, int splits,
                                      int accumulate, hipStream_t st) {
  if (defer_reduce({part_a, out_a, n4_a, 0, splits, accumulate, 0})) {
    if (out_b) defer_reduce({part_b, out_b, n4_b, 0, splits, accumulate, 0});
    return;
  }
  const int ba = cdiv(n4_a, 32), bb = out_b ? cdiv(n4_b, 32) : 0;
  split_reduce4x8_kernel<<<ba + bb, 256, 0, st>>>(part_a, n4_a, out_a, ba, part_b, n4_b, out_b,
                                                  splits, accumulate);
}
static void launch_split_reduce4(const float* part, int splits, int64_t n4, float* out,
                                 int accumulate, hipStream_t st) {
  if (splits >= 8) {
    launch_split_reduce4_pair(part, n4, out, nullptr, 0, nullptr, splits, accumulate, st);
  } else {
    if (defer_reduce({part, out, n4, 0, splits, accumulate, 1})) return;
    split_reduce4_kernel<<<cdiv(n4, 256), 256, 0, st>>>(part, splits, n4, out, accumulate);
  }
}
static void launch_split_reduce_strided(const float* part, int splits, int64_t stride, int64_t n,
                                        float* out, int accumulate, hipStream_t st) {
  if (defer_reduce({part, out, n, stride, splits, accumulate, 2})) return;
  split_reduce_strided_kernel<<<cdiv(n, 32), 32 * SR_ZL, 0, st>>>(part, splits, stride, n, out,
                                                                  accumulate);
}

// part[z][c] = sum over rows [z*rps, (z+1)*rps) of y[row][c]   (bias gradient when the weight
// gradient itself only visits one output phase per tap); 8 channels per thread, 8 rows in flight
__global__ __launch_bounds__(256) void colsum_part8_kernel(const bf16_t* __restrict__ y,
                                                           int64_t rows, int C, int64_t rps,
                                                           float* __restrict__ part) {
  __shared__ float sm[8][32][9];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int cv = blockIdx.x * 32 + cg;
  const int64_t r0 = (int64_t)blockIdx.y * rps, r1 = min(rows, r0 + rps);
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (cv * 8 < C) {
#pragma unroll 4
    for (int64_t r = r0 + rl; r < r1; r += 8) {
      union { uint4 q; bf16_t h[8]; } v;
      v.q = *reinterpret_cast<const uint4*>(y + r * C + (int64_t)cv * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += bf2f(v.h[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) sm[rl][cg][e] = s[e];
  __syncthreads();
  const int cg2 = threadIdx.x >> 3, e2 = threadIdx.x & 7;
  const int cv2 = blockIdx.x * 32 + cg2;
  if (cv2 * 8 < C) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += sm[r][cg2][e2];
    part[(int64_t)blockIdx.y * C + cv2 * 8 + e2] = t;
  }
}


bool phase_ok(const cgConvGeom* g) {
  if (g->U == 1) return true;
  return g->U == 2 && g->S == 1 && (g->Ho % 2) == 0 && (g->Wo % 2) == 0;
}

}  // namespace

ReduceDeferScope::ReduceDeferScope(cgDeferCtx* ctx) : old_(t_defer) { t_defer = ctx; }
ReduceDeferScope::~ReduceDeferScope() { t_defer = old_; }
ReduceDeferSuspend::ReduceDeferSuspend(bool on) : on_(on) {
  if (on_) ++t_defer_suspend;
}
ReduceDeferSuspend::~ReduceDeferSuspend() {
  if (on_) --t_defer_suspend;
}

extern "C" cgDeferCtx* cg_defer_create(void) { return new (std::nothrow) cgDeferCtx(); }
extern "C" void cg_defer_destroy(cgDeferCtx* ctx) { delete ctx; }
extern "C" int cg_defer_pending(cgDeferCtx* ctx) {
  if (!ctx) return 0;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return (int)ctx->jobs.size();
}
extern "C" int cg_defer_abort(cgDeferCtx* ctx) {
  if (!ctx) CG_FAIL(CG_ERR_BAD_ARG, "cg_defer_abort: null context");
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->jobs.clear();
  return CG_OK;
}
extern "C" int cg_defer_flush(cgDeferCtx* ctx, cgStream stream) {
  if (!ctx) CG_FAIL(CG_ERR_BAD_ARG, "cg_defer_flush: null context");
  std::vector<ReduceJob> jobs;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    jobs.swap(ctx->jobs);
  }
  if (jobs.empty()) return CG_OK;   // (no launch, no device needed)
  hipStream_t st = (hipStream_t)stream;
  for (int form = 0; form < 2; ++form) {   // 0: float4 kinds (0, 1), 1: strided
    ReduceChunk c;
    c.cnt = 0;
    int blk = 0;
    auto launch = [&]() {
      if (c.cnt == 0) return;
      c.blk[c.cnt] = blk;
      if (form == 0) split_reduce4_multi_kernel<<<blk, 256, 0, st>>>(c);
      else split_reduce_strided_multi_kernel<<<blk, 32 * SR_ZL, 0, st>>>(c);
      c.cnt = 0;
      blk = 0;
    };
    for (const ReduceJob& j : jobs) {
      if ((j.kind == 2) != (form == 1)) continue;
      const int i = c.cnt++;
      c.part[i] = j.part; c.out[i] = j.out; c.n[i] = j.n; c.stride[i] = j.stride;
      c.splits[i] = j.splits;
      c.flags[i] = (j.accumulate ? 1 : 0) | (j.kind == 1 ? 2 : 0);
      c.blk[i] = blk;
      blk += (int)(j.kind == 1 ? cdiv(j.n, 256) : cdiv(j.n, 32));
      if (c.cnt == RJ_MAX) launch();
    }
    launch();
  }
  CG_CHECK_LAUNCH("cg_defer_flush");
  return CG_OK;
}

void cg_split_reduce4_pair(const float* part_a, int64_t n4_a, float* out_a, const float* part_b,
                           int64_t n4_b, float* out_b, int splits, int accumulate,
                           hipStream_t st) {
  launch_split_reduce4_pair(part_a, n4_a, out_a, part_b, n4_b, out_b, splits, accumulate, st);
}

static int ilog2x(int x);

#ifdef CG_CONV_TIMING
static unsigned long long* g_conv_tdbg = nullptr;
extern "C" void cg_debug_set_conv_timing_buffer(void* p) { g_conv_tdbg = (unsigned long long*)p; }
#endif

bool cg_fast_conv_supported(const cgConvGeom* g, const void* in, const void* gate_in,
                            float slope_in) {
  if (g->Ci % 32 != 0) return false;   // 64-channel K slices; the last one may be half empty
  if (!phase_ok(g)) return false;
  if (gate_in && !(gate_in == in && slope_in == 0.f)) return false;
  if ((int64_t)g->N * g->Hin * g->Win * g->Ci >= (1ll << 31)) return false;
  if ((int64_t)g->Co * (((int64_t)g->kh * g->kw * g->Ci + 7) & ~7ll) >= (1ll << 31)) return false;
  return true;
}

void cg_fast_conv_launch(const cgConvGeom* g, const void* in, const void* bt, void* out,
                         int out_is_f32, const float* bias, const void* gate_in,
                         const void* gate_out, float slope_out, const void* residual,
                         hipStream_t st) {
  cg_fast_conv_launch_ld(g, in, g->Ci, bt, out, g->Co, out_is_f32, bias, gate_in, gate_out, slope_out,
                         residual, st);
}

static thread_local int g_gate_cols = -1;   // cg_fast_conv_launch_ld_cols -> cg_fast_conv_launch_ld

void cg_fast_conv_launch_ld_cols(const cgConvGeom* g, const void* in, int in_ld, const void* bt,
                                 void* out, int out_ld, int out_is_f32, const float* bias,
                                 int relu_cols, hipStream_t st) {
  g_gate_cols = relu_cols;
  cg_fast_conv_launch_ld(g, in, in_ld, bt, out, out_ld, out_is_f32, bias, nullptr,
                         relu_cols > 0 ? out : nullptr, 0.f, nullptr, st);
  g_gate_cols = -1;
}

bool cg_fast_conv_ld_supported(const cgConvGeom* g, int in_ld, int out_ld) {
  if (in_ld < g->Ci || out_ld < g->Co || (in_ld % 8) != 0 || (out_ld % 8) != 0 || (g->Co % 8) != 0)
    return false;
  if ((int64_t)g->N * g->Hin * g->Win * in_ld >= (1ll << 31)) return false;
  return cg_fast_conv_supported(g, nullptr, nullptr, 0.f);
}

void cg_fast_conv_launch_ld(const cgConvGeom* g, const void* in, int in_ld, const void* bt, void* out,
                            int out_ld, int out_is_f32, const float* bias, const void* gate_in,
                            const void* gate_out, float slope_out, const void* residual,
                            hipStream_t st) {
  FastConvArgs a;
  a.in_ld = in_ld;
  a.out_ld = out_ld;
  a.gm = 1;
  a.dGrp = make_fastdiv(1);
  a.in = (const bf16_t*)in;
  a.bt = (const bf16_t*)bt;
  a.out = out;
  a.bias = bias;
  a.self_gate = (gate_out != nullptr && gate_out == out);
  a.gate_cols = g_gate_cols >= 0 ? g_gate_cols : g->Co;
  a.gate_out = a.self_gate ? nullptr : (const bf16_t*)gate_out;
  a.residual = (const bf16_t*)residual;
  a.N = g->N; a.Hin = g->Hin; a.Win = g->Win; a.Ci = g->Ci;
  a.Ho = g->Ho; a.Wo = g->Wo; a.Co = g->Co; a.kh = g->kh; a.kw = g->kw;
  a.S = g->S; a.U = g->U; a.pt = g->pt; a.pl = g->pl;
  a.Kp = (g->kh * g->kw * g->Ci + 7) & ~7;
  a.Hp = g->Ho / g->U; a.Wp = g->Wo / g->U;
  a.Mp = g->N * a.Hp * a.Wp;
  a.cblocks = (g->Ci + 63) / 64;
#ifdef CG_CONV_TIMING
  a.tdbg = g_conv_tdbg;
#endif
  a.relu_in = gate_in != nullptr;
  a.out_f32 = out_is_f32;
  a.slope_out = slope_out;
  a.dWp = make_fastdiv(a.Wp); a.dHp = make_fastdiv(a.Hp);
  const int phases = g->U * g->U;
  // several taps, unit stride, wide output: stage the input window once per channel block
  // (experimental, CGAMD_HALO=1: measured on MI355X it only ties the plain kernel -- the staged bytes
  // were not the limiter, the epilogue and the per-iteration issue overheads were)
  static const int use_halo = [] {
    const char* e = getenv("CGAMD_HALO");
    return e ? atoi(e) : 0;
  }();
  if (use_halo && (g->Ci % 64) == 0 && g->S == 1 && g->Co > 64 && g->kh * g->kw >= 4 && (a.Hp & (a.Hp - 1)) == 0 &&
      (a.Wp & (a.Wp - 1)) == 0 && a.Hp >= 4 && a.Wp >= 4 && g->kh <= 5 && g->kw <= 5) {
    HaloArgs h;
    h.in = a.in; h.bt = a.bt; h.out = a.out; h.bias = a.bias;
    h.gate_out = a.gate_out; h.residual = a.residual;
    h.N = g->N; h.Hin = g->Hin; h.Win = g->Win; h.Ci = g->Ci; h.Ho = g->Ho; h.Wo = g->Wo;
    h.Co = g->Co; h.kh = g->kh; h.kw = g->kw; h.U = g->U; h.pt = g->pt; h.pl = g->pl;
    h.Kp = a.Kp; h.cblocks = a.cblocks; h.Hp = a.Hp; h.Wp = a.Wp;
    const int TW = a.Wp < 16 ? a.Wp : 16;
    int TH = 128 / TW;
    if (TH > a.Hp) TH = a.Hp;
    h.NI = 128 / (TW * TH);
    h.tw_log = ilog2x(TW); h.th_log = ilog2x(TH);
    h.tiles_x = a.Wp / TW; h.tiles_y = a.Hp / TH;
    h.img_groups = cdiv(g->N, h.NI);
    h.ntiles = cdiv(g->Co, 128);
    // largest halo over the phases: nr <= ceil(kh / U) rows more than the tile
    const int nr_max = (g->kh + g->U - 1) / g->U, ns_max = (g->kw + g->U - 1) / g->U;
    const int hrows = h.NI * (TH + nr_max - 1) * (TW + ns_max - 1);
    h.halo_groups = ((hrows + 7) / 8 + 3) / 4 * 4;
    if (h.halo_groups <= 4 * HALO_SLOTS) {
      h.relu_in = a.relu_in; h.out_f32 = a.out_f32; h.self_gate = a.self_gate;
      h.slope_out = a.slope_out;
      h.dNt = make_fastdiv(h.ntiles); h.dTx = make_fastdiv(h.tiles_x);
      h.dTy = make_fastdiv(h.tiles_y);
      const size_t lds = (size_t)(2 * h.halo_groups * 512 + 2 * 128 * 64) * sizeof(bf16_t);
      static const bool attr_set = [] {
        // more than 64 KiB of dynamic LDS needs the opt-in attribute
        (void)hipFuncSetAttribute((const void*)halo_conv_kernel<true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)halo_conv_kernel<false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
      }();
      (void)attr_set;
      dim3 grid(h.img_groups * h.tiles_y * h.tiles_x * h.ntiles, phases);
      CgProfScope prof(CG_PROF_HALO_CONV, g, st);
      if (h.relu_in)
        halo_conv_kernel<true><<<grid, 256, lds, st>>>(h);
      else
        halo_conv_kernel<false><<<grid, 256, lds, st>>>(h);
      return;
    }
  }
  // ring depth: a grid of <= ~1.5 workgroups per CU cannot rely on a co-resident workgroup to hide
  // the load latency -> deep ring (NS = 4); grids of >= 3 workgroups per CU run single-buffered
  // (NS = 1: 33 KiB of LDS, 3-4 co-resident workgroups overlap each other's loads, MFMA work and
  // epilogues: +15-25 % on the 1024-workgroup shapes); in between, NS = 2 with two per CU
  static const int ns_env = [] {
    const char* e = getenv("CGAMD_CONV_NS");
    return e ? atoi(e) : 0;
  }();
  // grids of at most one workgroup per CU: 8-wave intra-workgroup split-K form (CGAMD_CONV_SK =
  // largest such grid, 0 disables).  Measured -15 % on the 8x8 layers, -2.3 % on the CIFAR step.
  static const int sk_env = [] {
    const char* e = getenv("CGAMD_CONV_SK");
    return e ? atoi(e) : 256;
  }();
#define CG_LAUNCH_CONV(BM_, BN_, GRID_)                                                  \
  do {                                                                                   \
    const int blocks_ = (GRID_).x * (GRID_).y;                                           \
    const int ns_ = ns_env ? ns_env : (blocks_ <= 384 ? 4 : (blocks_ >= 768 ? 1 : 2));   \
    if (sk_env && blocks_ <= sk_env) {                                                   \
      if (a.relu_in) fast_conv_sk_kernel<BM_, BN_, true><<<GRID_, 512, 0, st>>>(a);      \
      else fast_conv_sk_kernel<BM_, BN_, false><<<GRID_, 512, 0, st>>>(a);               \
    } else if (ns_ >= 4) {                                                                      \
      if (a.relu_in) fast_conv_kernel<BM_, BN_, true, 4><<<GRID_, 256, 0, st>>>(a);      \
      else fast_conv_kernel<BM_, BN_, false, 4><<<GRID_, 256, 0, st>>>(a);               \
    } else if (ns_ == 1) {                                                               \
      if (a.relu_in) fast_conv_kernel<BM_, BN_, true, 1><<<GRID_, 256, 0, st>>>(a);      \
      else fast_conv_kernel<BM_, BN_, false, 1><<<GRID_, 256, 0, st>>>(a);               \
    } else {                                                                             \
      if (a.relu_in) fast_conv_kernel<BM_, BN_, true, 2><<<GRID_, 256, 0, st>>>(a);      \
      else fast_conv_kernel<BM_, BN_, false, 2><<<GRID_, 256, 0, st>>>(a);               \
    }                                                                                    \
  } while (0)
  if (g->Co <= 32) {
    // narrow outputs (RGB images, logits): 32-channel tile, the pixel dimension carries the grid
    a.ntiles = 1;
    a.dNt = make_fastdiv(1);
    a.mtiles = cdiv(a.Mp, 128);
    dim3 grid(a.mtiles, phases);
    CgProfScope prof(CG_PROF_FAST_CONV_128x32, g, st);
    CG_LAUNCH_CONV(128, 32, grid);
    return;
  }
  if (g->Co <= 64) {
    // 64-channel tile: no MFMA work is spent on padding channels
    a.ntiles = 1;
    a.dNt = make_fastdiv(1);
    a.mtiles = cdiv(a.Mp, 128);
    dim3 grid(a.mtiles, phases);
    CgProfScope prof(CG_PROF_FAST_CONV_128x64, g, st);
    CG_LAUNCH_CONV(128, 64, grid);
    return;
  }
  // 192-channel tiles for the channel counts 128-wide tiles pad by a quarter or more (Inception: 160 /
  // 192 outputs; BigGAN: 192) on grids large enough for the shallow rings: -13 ... -18 % per launch on
  // the 17 x 17 layers of an Inception batch (profiles/r06_inception_tiles_ab.txt).  A 96-wide tile
  // (one 32 x 96 MFMA tile per wave) measured equal on 3x3 and slower on 1x1 layers with 96 outputs and
  // was dropped.  CGAMD_CONV_BN_WIDE=0: the 128-wide tiles everywhere
  static const int wide_env = [] {
    const char* e = getenv("CGAMD_CONV_BN_WIDE");
    return e ? atoi(e) : 1;
  }();
  if (wide_env) {
    const int pad128 = cdiv(g->Co, 128) * 128;
    const int bn = cdiv(g->Co, 192) * 192 < pad128 ? 192 : 128;
    const int blocks = cdiv(a.Mp, 128) * cdiv(g->Co, bn) * phases;
    if (bn != 128 && blocks >= 512) {
      a.ntiles = cdiv(g->Co, bn);
      a.dNt = make_fastdiv(a.ntiles);
      a.mtiles = cdiv(a.Mp, 128);
      dim3 grid(a.mtiles * a.ntiles, phases);
      const bool one = ns_env ? ns_env == 1 : blocks >= 768;
#define CG_LAUNCH_WIDE(BN_)                                                              \
  do {                                                                                   \
    if (one) {                                                                           \
      if (a.relu_in) fast_conv_kernel<128, BN_, true, 1><<<grid, 256, 0, st>>>(a);       \
      else fast_conv_kernel<128, BN_, false, 1><<<grid, 256, 0, st>>>(a);                \
    } else {                                                                             \
      if (a.relu_in) fast_conv_kernel<128, BN_, true, 2><<<grid, 256, 0, st>>>(a);       \
      else fast_conv_kernel<128, BN_, false, 2><<<grid, 256, 0, st>>>(a);                \
    }                                                                                    \
  } while (0)
      CgProfScope prof(CG_PROF_FAST_CONV_128x192, g, st);
      CG_LAUNCH_WIDE(192);
#undef CG_LAUNCH_WIDE
      return;
    }
  }
  a.ntiles = cdiv(g->Co, 128);
  a.dNt = make_fastdiv(a.ntiles);
  const int tiles128 = cdiv(a.Mp, 128) * a.ntiles * phases;
  // 128x128 tiles only once they give > 2 workgroups per CU; below that 64x128 tiles double the
  // grid, which buys more overlap than the larger tile saves in operand traffic (measured: -2 % step)
  static const int t128_min = [] {
    const char* e = getenv("CGAMD_CONV_T128_MIN");
    return e ? atoi(e) : 513;
  }();
  // ... except when the 128-row tiles fit ONE round of single-resident workgroups (<= 256, deep
  // ring) and the 64-row tiles would need two: same MFMA time per CU, half the weight traffic
  // (BigGAN's 4x4x1536 layers, K = 13824: 295 -> 172 us, profiles/r03_biggan_launches.txt)
  const int tiles64 = cdiv(a.Mp, 64) * a.ntiles * phases;
  if (tiles128 >= t128_min || (tiles128 <= 256 && tiles64 > 256 && tiles64 <= 384)) {
    a.mtiles = cdiv(a.Mp, 128);
    // tile groups (see the kernel) where the weight image dwarfs an XCD's 4 MB L2 and there are enough M
    // tiles to form groups; CGAMD_CONV_GM = group height (1 = off, 0 = this policy)
    static const int gm_env = [] {
      const char* e = getenv("CGAMD_CONV_GM");
      return e ? atoi(e) : 0;
    }();
    const int64_t wbytes = (int64_t)g->Co * a.Kp * 2;
    int gm = gm_env ? gm_env : ((wbytes >= (8ll << 20) && a.mtiles >= 32 && a.ntiles >= 2) ? 16 : 1);
    if (gm > a.mtiles) gm = a.mtiles;
    if (gm > 1) {
      a.gm = gm;
      a.dGrp = make_fastdiv(gm * a.ntiles);
    }
    dim3 grid(a.mtiles * a.ntiles, phases);
    CgProfScope prof(CG_PROF_FAST_CONV_128x128, g, st);
    CG_LAUNCH_CONV(128, 128, grid);
  } else {
    a.mtiles = cdiv(a.Mp, 64);
    dim3 grid(a.mtiles * a.ntiles, phases);
    CgProfScope prof(CG_PROF_FAST_CONV_64x128, g, st);
    CG_LAUNCH_CONV(64, 128, grid);
  }
#undef CG_LAUNCH_CONV
}

static int ilog2x(int x) {
  int l = 0;
  while ((1 << l) < x) ++l;
  return ((1 << l) == x) ? l : -1;
}

static bool stem_geom_ok(const cgConvGeom* g) {
  return g->U == 1 && g->Ci <= 4 && g->kh * g->kw * g->Ci <= 128 &&
         (int64_t)g->N * g->Ho * g->Wo < (1ll << 31);
}

static void stem_fill(const cgConvGeom* g, StemArgs* a) {
  a->N = g->N; a->Hin = g->Hin; a->Win = g->Win; a->Ci = g->Ci;
  a->Ho = g->Ho; a->Wo = g->Wo; a->Co = g->Co; a->kh = g->kh; a->kw = g->kw;
  a->S = g->S; a->pt = g->pt; a->pl = g->pl;
  a->M = g->N * g->Ho * g->Wo;
  a->K = g->kh * g->kw * g->Ci;
  a->Kp = (a->K + 7) & ~7;
  a->KS = (a->K + 31) & ~31;
  a->dWo = make_fastdiv(g->Wo); a->dHo = make_fastdiv(g->Ho);
  a->dCi = make_fastdiv(g->Ci); a->dKw = make_fastdiv(g->kw);
}

bool cg_stem_conv_supported(const cgConvGeom* g, const void* in, const void* out,
                            const void* gate_in, float slope_in, const void* gate_out,
                            const void* residual) {
  if (!stem_geom_ok(g) || g->Co > 1024 || (gate_out && gate_out != out) || residual) return false;
  if (gate_in && !(gate_in == in && slope_in == 0.f)) return false;
  return true;
}

void cg_stem_conv_launch(const cgConvGeom* g, const void* in, const void* bt, void* out,
                         int out_is_f32, const float* bias, const void* gate_in,
                         const void* gate_out, float slope_out, hipStream_t st) {
  StemArgs a;
  stem_fill(g, &a);
  a.self_gate = gate_out != nullptr;
  a.slope_out = slope_out;
  a.in = (const bf16_t*)in; a.bt = (const bf16_t*)bt; a.dy = nullptr; a.out = out; a.bias = bias;
  a.relu_in = gate_in != nullptr; a.out_f32 = out_is_f32; a.want_bias = 0; a.rows_per_split = 0;
  a.adjoint_out = 0;
  const dim3 grid(cdiv(a.M, 128), cdiv(g->Co, 128));
  CgProfScope prof(CG_PROF_STEM_FWD, g, st);
  switch (a.KS) {
    case 32: stem_fwd_kernel<32><<<grid, 256, 0, st>>>(a); break;
    case 64: stem_fwd_kernel<64><<<grid, 256, 0, st>>>(a); break;
    case 96: stem_fwd_kernel<96><<<grid, 256, 0, st>>>(a); break;
    default: stem_fwd_kernel<128><<<grid, 256, 0, st>>>(a); break;
  }
}

bool cg_stem_wgrad_supported(const cgConvGeom* g, const void* in, const void* gate_in,
                             float slope_in, const void* gate_dy) {
  if (!stem_geom_ok(g) || gate_dy || g->Co % 8 != 0) return false;
  if (gate_in && !(gate_in == in && slope_in == 0.f)) return false;
  return true;
}

static void stem_wgrad_plan(const cgConvGeom* g, int* splits, int* rps) {
  const int M = g->N * g->Ho * g->Wo;
  int s = M / 256 > 0 ? M / 256 : 1;   // the partial image is tiny (K <= 128 rows): split finely
  if (s > 1024) s = 1024;
  int r = cdiv(M, s);
  r = (r + 63) / 64 * 64;
  *splits = cdiv(M, r);
  *rps = r;
}

size_t cg_stem_wgrad_workspace_bytes(const cgConvGeom* g) {
  int splits, rps;
  stem_wgrad_plan(g, &splits, &rps);
  const size_t K = (size_t)g->kh * g->kw * g->Ci;
  const size_t a = align_up((size_t)splits * (K * g->Co + g->Co) * sizeof(float), 256);
  const size_t b = cg_wstem_wgrad_workspace_bytes(g);   // window-staged form (0 if not covered)
  return a > b ? a : b;
}

// geometry of the adjoint (data-gradient) convolution: input = g's output space
static cgConvGeom adjoint_geom(const cgConvGeom* g) {
  cgConvGeom t;
  t.N = g->N; t.Hin = g->Ho; t.Win = g->Wo; t.Ci = g->Co;
  t.Ho = g->Hin; t.Wo = g->Win; t.Co = g->Ci;
  t.kh = g->kh; t.kw = g->kw; t.S = g->U; t.U = g->S;
  t.pt = g->kh - 1 - g->pt; t.pl = g->kw - 1 - g->pl;
  return t;
}

// narrow outputs (Co <= 4, e.g. the generator's RGB convolution): dw[k][co] = sum x[k] dy[co] is the
// stem weight gradient of the ADJOINT convolution with the operands swapped
bool cg_narrow_wgrad_supported(const cgConvGeom* g, const void* gate_in, const void* gate_dy) {
  if (gate_in || gate_dy) return false;
  if (g->S != 1 || g->U != 1 || g->Co > 4 || g->Ci % 8 != 0) return false;
  const cgConvGeom t = adjoint_geom(g);
  return stem_geom_ok(&t);
}

size_t cg_narrow_wgrad_workspace_bytes(const cgConvGeom* g) {
  const cgConvGeom t = adjoint_geom(g);
  return cg_stem_wgrad_workspace_bytes(&t);
}

static void stem_wgrad_run(const cgConvGeom* g, const void* in, int relu_in, const void* dy,
                           float* dw, int accumulate, float* dbias, void* ws, int adjoint_out,
                           hipStream_t st);

void cg_narrow_wgrad_launch(const cgConvGeom* g, const void* in, const void* dy, float* dw,
                            int accumulate, void* ws, hipStream_t st) {
  const cgConvGeom t = adjoint_geom(g);
  // adjoint conv: "input" = dy (Co channels), "output gradient" = x
  stem_wgrad_run(&t, dy, 0, in, dw, accumulate, nullptr, ws, 1, st);
}

void cg_stem_partial_reduce(const cgConvGeom* g, const void* ws, int splits, float* dw,
                            float* dbias, int accumulate, hipStream_t st) {
  const int64_t KC = (int64_t)g->kh * g->kw * g->Ci * g->Co, stride = KC + g->Co;
  launch_split_reduce_strided((const float*)ws, splits, stride, KC, dw, accumulate, st);
  if (dbias)
    launch_split_reduce_strided((const float*)ws + KC, splits, stride, g->Co, dbias, accumulate, st);
}

void cg_stem_wgrad_launch(const cgConvGeom* g, const void* in, const void* gate_in,
                          const void* dy, float* dw, int accumulate, float* dbias, void* ws,
                          hipStream_t st) {
  if (cg_wstem_wgrad_supported(g, in, gate_in, gate_in ? 0.f : 0.f, nullptr)) {
    // window-staged kernel (cg_conv_halo.hip); same partial layout, same strided reduce
    CgProfScope prof(CG_PROF_STEM_WGRAD, g, st);
    int splits = 0;
    cg_wstem_wgrad_launch(g, in, gate_in, dy, dbias != nullptr, ws, &splits, st);
    cg_stem_partial_reduce(g, ws, splits, dw, dbias, accumulate, st);
    return;
  }
  stem_wgrad_run(g, in, gate_in != nullptr, dy, dw, accumulate, dbias, ws, 0, st);
}

static void stem_wgrad_run(const cgConvGeom* g, const void* in, int relu_in, const void* dy,
                           float* dw, int accumulate, float* dbias, void* ws, int adjoint_out,
                           hipStream_t st) {
  StemArgs a;
  stem_fill(g, &a);
  int splits, rps;
  stem_wgrad_plan(g, &splits, &rps);
  a.in = (const bf16_t*)in; a.bt = nullptr; a.dy = (const bf16_t*)dy; a.out = ws; a.bias = nullptr;
  a.relu_in = relu_in; a.out_f32 = 1; a.want_bias = dbias != nullptr;
  a.adjoint_out = adjoint_out;
  a.self_gate = 0; a.slope_out = 0.f;
  a.rows_per_split = rps;
  dim3 grid(cdiv(g->Co, 128), splits);
  CgProfScope prof(CG_PROF_STEM_WGRAD, g, st);
  switch (a.KS) {
    case 32: stem_wgrad_kernel<32><<<grid, 256, 0, st>>>(a); break;
    case 64: stem_wgrad_kernel<64><<<grid, 256, 0, st>>>(a); break;
    case 96: stem_wgrad_kernel<96><<<grid, 256, 0, st>>>(a); break;
    default: stem_wgrad_kernel<128><<<grid, 256, 0, st>>>(a); break;
  }
  const int64_t KC = (int64_t)a.K * g->Co, stride = KC + g->Co;
  launch_split_reduce_strided((const float*)ws, splits, stride, KC, dw, accumulate, st);
  if (dbias)
    launch_split_reduce_strided((const float*)ws + KC, splits, stride, g->Co, dbias, accumulate, st);
}

bool cg_fast_wgrad_supported(const cgConvGeom* g, const void* in, const void* gate_in,
                             float slope_in, const void* gate_dy) {
  if (g->Ci % 32 != 0 || g->Co % 8 != 0) return false;
  if (!phase_ok(g)) return false;
  if (gate_dy) return false;
  if (gate_in && !(gate_in == in && slope_in == 0.f)) return false;
  if ((int64_t)g->kh * g->kw * g->Ci * g->Co >= (1ll << 31)) return false;
  return true;
}

void cg_fast_wgrad_plan(const cgConvGeom* g, int* splits, int* rows_per_split) {
  const int Mp = g->N * (g->Ho / g->U) * (g->Wo / g->U);
  const int tkc = (g->Ci % 128 == 0) ? 128 : 64;
  const int tiles = g->kh * g->kw * cdiv(g->Ci, tkc) * cdiv(g->Co, 128);
  // each split costs one fp32 partial image of the whole weight (written, then re-read by the
  // reduce): only split as far as filling the chip needs, and never below 8 row slices per split
  int s = cdiv(384, tiles);
  const int max_by_rows = Mp / 512 > 0 ? Mp / 512 : 1;
  // ... unless the weight is tiny (1x1 projections of the attention block, 96 x 32: 12 KiB per
  // partial): the launch is a pure reduction over half a million pixels then, and 2 tiles x 64
  // splits left half the chip idle at 0.9 TB/s -- up to 256 splits while the partials stay < 8 MiB
  const int64_t kc = (int64_t)g->kh * g->kw * g->Ci * g->Co;
  const int cap = kc * 4 * 256 <= (8ll << 20) ? 256 : 64;
  if (cap > 64) s = cdiv(512, tiles);
  if (s > max_by_rows) s = max_by_rows;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  int rps = cdiv(Mp, s);
  rps = (rps + 63) / 64 * 64;
  s = cdiv(Mp, rps);
  *splits = s;
  *rows_per_split = rps;
}

// CGAMD_ASM_DMA=0 selects the builtin LDS-DMA forms of the weight-gradient kernels (A/B switch)
static bool asm_dma_enabled() {
  static const int on = [] {
    const char* e = getenv("CGAMD_ASM_DMA");
    return e ? atoi(e) : 1;
  }();
  return on != 0;
}

static bool halo_wgrad_ok(const cgConvGeom* g) {
  static const int off = [] {
    const char* e = getenv("CGAMD_NO_HALO_WGRAD");
    return e ? atoi(e) : 0;
  }();
  return !off && g->S == 1 && g->U == 1 && g->kh == 3 && g->kw == 3 &&
         (g->Ho & (g->Ho - 1)) == 0 && (g->Wo & (g->Wo - 1)) == 0 && g->Ho >= 4 && g->Wo >= 4 &&
         g->Ho == g->Hin && g->Wo == g->Win;
}

struct HaloWgradPlan {
  int TW, TH, NI, tiles_x, tiles_y, img_groups, nslices, splits, sps;
};
static HaloWgradPlan halo_wgrad_plan(const cgConvGeom* g) {
  HaloWgradPlan p;
  p.TW = g->Wo < 16 ? g->Wo : 16;
  p.TH = 64 / p.TW;
  if (p.TH > g->Ho) p.TH = g->Ho;
  p.NI = 64 / (p.TW * p.TH);
  p.tiles_x = g->Wo / p.TW;
  p.tiles_y = g->Ho / p.TH;
  p.img_groups = cdiv(g->N, p.NI);
  p.nslices = p.img_groups * p.tiles_y * p.tiles_x;
  const int tiles = cdiv(g->Ci, 64) * cdiv(g->Co, 64);
  // two resident workgroups per CU hide the staging latency; every split costs a K x Co fp32 partial
  static const int target = [] {
    const char* e = getenv("CGAMD_HALO_BLOCKS");
    return e ? atoi(e) : 512;
  }();
  int s = cdiv(target, tiles);
  const int max_by = p.nslices / 4 > 0 ? p.nslices / 4 : 1;   // >= 4 slices per split
  if (s > max_by) s = max_by;
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  p.sps = cdiv(p.nslices, s);
  p.splits = cdiv(p.nslices, p.sps);
  return p;
}

size_t cg_fast_wgrad_workspace_bytes(const cgConvGeom* g) {
  int splits, rps;
  cg_fast_wgrad_plan(g, &splits, &rps);
  if (halo_wgrad_ok(g)) {
    const HaloWgradPlan p = halo_wgrad_plan(g);
    const size_t K = (size_t)g->kh * g->kw * g->Ci;
    size_t need = p.splits > 1 ? (size_t)p.splits * (K + 1) * g->Co * sizeof(float) : 256;
    const size_t bias_need = (size_t)CG_FAST_BIAS_SPLITS * g->Co * sizeof(float);
    if (need < bias_need) need = bias_need;
    return align_up(need, 256);
  }
  const size_t K = (size_t)g->kh * g->kw * g->Ci;
  size_t need = 256;
  if (splits > 1) need = (size_t)splits * (K * g->Co + g->Co) * sizeof(float);
  const size_t bias_need = (size_t)CG_FAST_BIAS_SPLITS * g->Co * sizeof(float);
  if (g->U != 1 && need < bias_need) need = bias_need;
  return align_up(need, 256);
}

void cg_fast_wgrad_launch(const cgConvGeom* g, const void* in, const void* gate_in,
                          const void* dy, float* dw, int accumulate, float* dbias, void* ws,
                          hipStream_t st) {
  if (halo_wgrad_ok(g)) {
    const HaloWgradPlan p = halo_wgrad_plan(g);
    HaloWgradArgs h;
    h.in = (const bf16_t*)in; h.dy = (const bf16_t*)dy;
    h.N = g->N; h.Hin = g->Hin; h.Win = g->Win; h.Ci = g->Ci; h.Ho = g->Ho; h.Wo = g->Wo;
    h.Co = g->Co; h.kh = g->kh; h.kw = g->kw; h.pt = g->pt; h.pl = g->pl;
    h.K = g->kh * g->kw * g->Ci;
    h.cblocks = cdiv(g->Ci, 64); h.ntiles = cdiv(g->Co, 64);
    h.tw_log = ilog2x(p.TW); h.th_log = ilog2x(p.TH); h.NI = p.NI;
    h.tiles_x = p.tiles_x; h.tiles_y = p.tiles_y; h.nslices = p.nslices;
    h.slices_per_split = p.sps;
    h.relu_in = gate_in != nullptr; h.accumulate = accumulate;
    h.dNt = make_fastdiv(h.ntiles); h.dTx = make_fastdiv(p.tiles_x);
    h.dTy = make_fastdiv(p.tiles_y);
    float* wsf = (float*)ws;
    const size_t KC = (size_t)h.K * g->Co;
    h.out = p.splits == 1 ? dw : wsf;
    h.bias = !dbias ? nullptr : (p.splits == 1 ? dbias : wsf + (size_t)p.splits * KC);
    dim3 grid(h.cblocks * h.ntiles, p.splits);
    CgProfScope prof(CG_PROF_HALO_WGRAD, g, st);
    const bool w4 = p.TW == 4;
#define HALO_WG(ASM_)                                                                  \
  do {                                                                                 \
    if (h.relu_in) {                                                                   \
      if (w4) halo_wgrad_kernel<true, true, ASM_><<<grid, 256, 0, st>>>(h);            \
      else halo_wgrad_kernel<true, false, ASM_><<<grid, 256, 0, st>>>(h);              \
    } else {                                                                           \
      if (w4) halo_wgrad_kernel<false, true, ASM_><<<grid, 256, 0, st>>>(h);           \
      else halo_wgrad_kernel<false, false, ASM_><<<grid, 256, 0, st>>>(h);             \
    }                                                                                  \
  } while (0)
    if (asm_dma_enabled()) HALO_WG(true);
    else HALO_WG(false);
#undef HALO_WG
    if (p.splits > 1) {
      const int64_t n4 = (int64_t)(KC / 4);
      launch_split_reduce4_pair(wsf, n4, dw, wsf + (size_t)p.splits * KC, g->Co / 4, dbias,
                                p.splits, accumulate, st);
    }
    return;
  }
  int splits, rps;
  cg_fast_wgrad_plan(g, &splits, &rps);
  FastWgradArgs a;
  a.in = (const bf16_t*)in;
  a.dy = (const bf16_t*)dy;
  a.N = g->N; a.Hin = g->Hin; a.Win = g->Win; a.Ci = g->Ci;
  a.Ho = g->Ho; a.Wo = g->Wo; a.Co = g->Co; a.kh = g->kh; a.kw = g->kw;
  a.S = g->S; a.U = g->U; a.pt = g->pt; a.pl = g->pl;
  a.Hp = g->Ho / g->U; a.Wp = g->Wo / g->U;
  a.Mp = g->N * a.Hp * a.Wp;
  a.K = g->kh * g->kw * g->Ci;
  const int tkc = (g->Ci % 128 == 0) ? 128 : 64;
  a.cblocks = cdiv(g->Ci, tkc);
  a.ktiles = g->kh * g->kw * a.cblocks;
  a.ntiles = cdiv(g->Co, 128);
  a.rows_per_split = rps;
  a.relu_in = gate_in != nullptr;
  a.accumulate = accumulate;
  a.dWp = make_fastdiv(a.Wp); a.dHp = make_fastdiv(a.Hp);
  a.dCb = make_fastdiv(a.cblocks); a.dNt = make_fastdiv(a.ntiles);
  float* wsf = (float*)ws;
  const size_t KC = (size_t)a.K * g->Co;
  // with zero insertion every tap only visits one output phase, so the bias gradient (a sum over
  // ALL output pixels) is reduced by its own pass below
  const bool bias_in_kernel = dbias && g->U == 1;
  if (splits == 1) {
    a.out = dw;
    a.bias_out = bias_in_kernel ? dbias : nullptr;
  } else {
    a.out = wsf;
    a.bias_out = bias_in_kernel ? wsf + (size_t)splits * KC : nullptr;
  }
  dim3 grid(a.ktiles * a.ntiles, splits);
  CgProfScope prof(tkc == 128 ? CG_PROF_FAST_WGRAD_128 : CG_PROF_FAST_WGRAD_64, g, st);
#define FAST_WG(TKC_, ASM_)                                                            \
  do {                                                                                 \
    if (a.relu_in) fast_wgrad_kernel<TKC_, true, ASM_><<<grid, 256, 0, st>>>(a);       \
    else fast_wgrad_kernel<TKC_, false, ASM_><<<grid, 256, 0, st>>>(a);                \
  } while (0)
  if (tkc == 128) {
    if (asm_dma_enabled()) FAST_WG(128, true);
    else FAST_WG(128, false);
  } else {
    if (asm_dma_enabled()) FAST_WG(64, true);
    else FAST_WG(64, false);
  }
#undef FAST_WG
  const int64_t c4 = g->Co / 4;
  if (splits > 1) {
    const int64_t n4 = (int64_t)(KC / 4);  // Ci % 64 == 0 and Co % 8 == 0 -> divisible
    // (not deferred when the bias partials below reuse the workspace the dw partials sit in)
    ReduceDeferSuspend keep(dbias && !bias_in_kernel);
    launch_split_reduce4_pair(wsf, n4, dw, wsf + (size_t)splits * KC, c4,
                              bias_in_kernel ? dbias : nullptr, splits, accumulate, st);
  }
  if (dbias && !bias_in_kernel) {
    // stream-ordered after the reduce above: the partial area of the workspace is free again
    const int64_t rows = (int64_t)g->N * g->Ho * g->Wo;
    int bs = (int)(rows / 256 > 0 ? rows / 256 : 1);
    if (bs > CG_FAST_BIAS_SPLITS) bs = CG_FAST_BIAS_SPLITS;
    const int64_t rps = (rows + bs - 1) / bs;
    bs = (int)((rows + rps - 1) / rps);
    dim3 bgrid(cdiv(g->Co / 8, 32), bs);
    colsum_part8_kernel<<<bgrid, 256, 0, st>>>((const bf16_t*)dy, rows, g->Co, rps, wsf);
    launch_split_reduce4(wsf, bs, c4, dbias, accumulate, st);
  }
}
