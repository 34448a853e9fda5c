// Halo-staged implicit-GEMM convolution for gfx950: unit-stride filters of up to 3x3 taps (and the
// 1..4-tap phase filters of zero-inserted inputs) on feature maps of at least 16x16.
// Contract and reference call sites: include/cgamd.h (cg_gconv: arch_ops.conv2d, arch_ops.py:559-573;
// resnet_ops.unpool + conv, resnet_ops.py:35-56,112-134; and their data gradients); this file only adds a
// faster kernel behind the same entry point.
//
// Why (profiles/r01_conv_phase_stamps.txt, r01_pmc_traffic.json): the one-tap-per-K-slice kernel
// (cg_conv_fast.hip) re-stages every input pixel once per tap -- 9x for a 3x3 filter -- and its
// speed is set by the L2 -> LDS fill path (64 B/clk/CU) and by the VALU cost of the per-piece address
// arithmetic.  Here a workgroup owns a 256-pixel spatial tile (8x32 or 16x16) x BN output channels:
//  * the input window of the tile WITH its halo ((TH+2) x (TW+2) pixels x 64 channels = 42.5 KiB) is
//    staged ONCE per 64-channel block and every tap reads its shifted view of that LDS image, so only
//    the weights (BN x 64 x 2 B per tap) stream per K-slice: ~4x fewer staged bytes per MFMA;
//  * staging uses buffer_load ... lds (raw buffer, 16 B per lane): the per-lane byte offsets are
//    computed once per workgroup, the tap / channel-block offset travels in the scalar offset, and
//    padding is the hardware bounds check (offset 0x80000000 -> zeros), so issuing a 1-KiB piece is
//    one vector-memory instruction with no address arithmetic in the K loop;
//  * 8 waves (4 along pixels x 2 along channels, 64x64 or 64x32 per wave), at most 128 VGPRs and
//    75 KiB of LDS: two workgroups (16 waves) per CU overlap each other's staging, MFMA work and
//    epilogues;
//  * LDS image: 128-byte rows (one pixel x 64 channels), 16-byte chunk c of a row stored at chunk
//    c ^ ((halo_x >> 1) & 7): the 16 lanes of every ds_read_b128 group hold 16 distinct halo_x values
//    for any tap shift, so fragment reads are conflict-free (halo pitch TW + 2 is even);
//  * epilogue: per-wave staging through LDS (one workgroup barrier), bias / activation / gate /
//    residual on 8 consecutive channels per lane, hardware bf16 conversion, 16-byte stores.
#include "cg_conv_fast.h"

#include <stdlib.h>

#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;

constexpr uint32_t HC_OOB = 0x80000000u;   // voffset of a lane that must read zeros (bounds check)
constexpr int HC_HALO_PIECES = 43;         // 1-KiB pieces (8 halo pixels each): 10x34 = 340 rows
constexpr int HC_HALO_BYTES = HC_HALO_PIECES * 1024;
constexpr int HC_HSLOTS = 6;               // halo pieces per wave (8 waves)

struct HConvArgs {
  const bf16_t* in;
  const bf16_t* bt;
  void* out;
  const float* bias;
  const bf16_t* gate_out;
  const bf16_t* residual;
  uint32_t in_bytes, bt_bytes;
  int N, Hin, Win, Ci, Ho, Wo, Co, kh, kw, U, pt, pl;
  int Kp, cblocks;
  int tiles_x, tiles_y, ntiles;
  int out_f32, self_gate;
  float slope_out;
  // fused batch norm (no-gradient forward passes): input prologue relu(((x - mean) * rstd) * gamma
  // + beta) applied to the staged window in LDS, and per-channel partial sums of the stored output
  // for the NEXT batch norm: stats[row][0:Co] = sum, [Co:2Co] = sum of squares, row = phase *
  // (spatial tiles) + spatial tile
  const float* bn_mean;
  const float* bn_var;
  const float* bn_gamma;
  const float* bn_beta;
  float bn_eps;
  int bn_per_sample;
  int bn_stat_group;   // > 0: bn_mean / bn_var are [N / bn_stat_group][Ci]
  float* stats;
  // 2x2 average pooling fused around the convolution (resnet_ops.py:131-133 and its gradient):
  //  pool   : the output is pooled in the epilogue -> [N, Ho/2, Wo/2, Co] (bias before, residual
  //           after the pooling); no output gates
  //  in_up  : the input tensor is [N, Hin/2, Win/2, Ci] and read as its nearest-neighbour
  //           up-sampling (the gradient of the pooling is that up-sampling times 1/4 = out_scale)
  int pool, in_up;
  float out_scale;
  FastDiv dNt, dTx, dTy;
#ifdef CG_CONV_TIMING
  unsigned long long* tdbg;   // 8 stamps per workgroup (scripts/hconv_timeline.py)
#endif
};
#ifdef CG_CONV_TIMING
#define HC_STAMP(slot)                                                                          \
  do {                                                                                          \
    if (a.tdbg && tid == 0)                                                                     \
      a.tdbg[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define HC_STAMP(slot) do {} while (0)
#endif

__device__ __forceinline__ int hc_xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// 4-byte LDS-DMA from a per-lane global address: lane l lands at lds_wave_base + 4 l
__device__ __forceinline__ void hc_glds4(const float* gptr, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (lds_void_t*)lds_wave_base, 4, 0, 0);
}

__device__ __forceinline__ void hc_dma16(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff,
                                         unsigned char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)lds_wave_base, 16, voff, soff, 0, 0);
}

// x of the lane `r` positions to the left within its 16-lane row (DPP row_ror:r, r compile-time)
__device__ __forceinline__ float hc_row_ror(float x, int r) {
  const int xi = __builtin_bit_cast(int, x);
  int y;
  switch (r) {
    case 1: y = __builtin_amdgcn_update_dpp(xi, xi, 0x121, 0xf, 0xf, false); break;
    case 2: y = __builtin_amdgcn_update_dpp(xi, xi, 0x122, 0xf, 0xf, false); break;
    case 4: y = __builtin_amdgcn_update_dpp(xi, xi, 0x124, 0xf, 0xf, false); break;
    default: y = __builtin_amdgcn_update_dpp(xi, xi, 0x128, 0xf, 0xf, false); break;
  }
  return __builtin_bit_cast(float, y);
}

__device__ __forceinline__ bf16x8_t hc_relu(bf16x8_t v) {
  s16x8_t s = __builtin_bit_cast(s16x8_t, v);
  const s16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
  s = __builtin_elementwise_max(s, z);
  return __builtin_bit_cast(bf16x8_t, s);
}

// BN: output channels per workgroup (128 or 64); TWL: log2 of the tile width (5: 8x32, 4: 16x16)
// FUSE: 0 = plain, 1 = batch-norm prologue / statistics epilogue, 2 = pooled epilogue (separate
// instantiations: the fusions must not cost the plain kernel registers), 3 = narrow outputs (Co < 8:
// the RGB convolution that ends every generator, resnet_cifar.py:108-111, resnet5.py:90-93; BN = 64,
// one 32-row MFMA tile whose K range the two channel-half waves split between them)
template <int BN, bool RELU, int TWL, int FUSE>
__global__ __launch_bounds__(512, 4) void hconv_kernel(HConvArgs a) {
  constexpr bool NARROW = FUSE == 3;
  static_assert(!NARROW || BN == 64, "narrow outputs use the 64-channel tile");
  // TWL == 3: FOUR whole 8x8 images per 256-pixel tile (the 8x8 x 512-channel blocks of the ResNet5
  // discriminator, resnet5.py:99-145: a single image is a quarter of a tile); pixel p of the tile
  // is (image p >> 6, row (p >> 3) & 7, column p & 7), every image has its own 10x10 halo window
  constexpr bool MI = TWL == 3;
  static_assert(!MI || (FUSE == 0 && BN == 64), "multi-image tiles: plain epilogue, 64-channel tile");
  constexpr int TW = MI ? 8 : (1 << TWL), TH = MI ? 8 : (256 >> TWL), PITCH = TW + 2;
  constexpr int IMG_ROWS = (TH + 2) * PITCH;                       // halo rows of one image (MI)
  constexpr int HALO_PIECES = MI ? (4 * IMG_ROWS + 7) / 8 : HC_HALO_PIECES;
  constexpr int HALO_BYTES = HALO_PIECES * 1024;
  constexpr int HSLOTS = (HALO_PIECES + 7) / 8;                    // halo pieces per wave
  constexpr int TN = BN / 64;         // 32-channel MFMA tiles per wave (2 waves along channels)
  constexpr int BJ = BN / 64;         // weight staging pieces per wave and K-slice
  constexpr int B_BYTES = BN * 128;   // one K-slice of weights: BN rows x 64 k x 2 B
  constexpr int LDS_BYTES = HALO_BYTES + 2 * B_BYTES;
  constexpr int TAB_OFF = LDS_BYTES;  // [4][64] floats: mean, rstd, gamma, beta of the channel block
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES + 1024];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // waves w and w + 4 share a SIMD (a workgroup's waves go to the SIMDs cyclically): they are the two
  // CHANNEL halves of one pixel quarter, so that a channel half with fewer real output channels
  // (Co = 96, 192: half of every second 64-channel group is empty) halves the matrix work of EVERY
  // SIMD instead of idling two of the four
  const int wm = wave & 3, wn = wave >> 2;
  const int frow = lane & 31, half = lane >> 5;
  HC_STAMP(0);
#ifdef CG_CONV_TIMING
  if (a.tdbg && tid == 0) {
    a.tdbg[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + 5] = __builtin_amdgcn_s_memrealtime();
    a.tdbg[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] =
        __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |                 // HW_ID
        ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);  // XCC_ID
  }
#endif

  // ---- workgroup -> (image, tile row, tile column, channel tile), XCD-contiguous ----
  const int wg = hc_xcd_remap(blockIdx.x, gridDim.x);
  const int st = (int)fdiv((uint32_t)wg, a.dNt);
  const int nt = wg - st * a.ntiles;
  const int t1 = (int)fdiv((uint32_t)st, a.dTx);
  const int tx = st - t1 * a.tiles_x;
  const int n = (int)fdiv((uint32_t)t1, a.dTy);
  const int ty = t1 - n * a.tiles_y;
  const int n0 = nt * BN;

  // ---- phase geometry (wave-uniform; as fast_conv_kernel) ----
  const int phase = blockIdx.y;
  int r0 = 0, s0 = 0, nr = a.kh, ns = a.kw, bh = -a.pt, bw = -a.pl, ph = 0, pw = 0;
  if (a.U == 2) {
    ph = phase >> 1;
    pw = phase & 1;
    r0 = (a.pt + ph) & 1;
    s0 = (a.pl + pw) & 1;
    nr = (a.kh - r0 + 1) >> 1;
    ns = (a.kw - s0 + 1) >> 1;
    bh = (ph - a.pt + r0) >> 1;   // exact: the numerator is even
    bw = (pw - a.pl + s0) >> 1;
  }
  const int ntaps = nr * ns;
  const int nk = ntaps * a.cblocks;
  const int HH = TH + nr - 1, HWID = TW + ns - 1;        // halo rows / columns actually read
  const int npieces = MI ? HALO_PIECES : (HH * PITCH + 7) >> 3;   // <= HALO_PIECES

  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bt =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.bt, 0, a.bt_bytes, 0x00020000);

  // ---- staging descriptors: byte offsets, computed once ----
  // halo piece p = wave + 8 j covers halo rows 8 p .. 8 p + 7; lane -> row 8 p + (lane >> 3), LDS
  // chunk (lane & 7) which must hold source chunk (lane & 7) ^ ((hx >> 1) & 7)
  uint32_t hvoff[HSLOTS];
  int hc8[HSLOTS];   // source channel offset of the slot's chunk (only read when Ci % 64 != 0)
  const int iy0 = ty * TH + bh, ix0 = tx * TW + bw;
  const bool ragged_ci = (a.Ci & 63) != 0;   // last channel block half empty (Ci % 32 == 0)
  {
#pragma unroll
    for (int j = 0; j < HSLOTS; ++j) {
      const int row = (wave + 8 * j) * 8 + (lane >> 3);
      if constexpr (MI) {
        // chunk swizzle ((hx >> 1) & 3) | ((hy & 1) << 2): the 16 lanes of a fragment read cover 8
        // columns of two rows, whose halo rows are 10 apart (the same bank half)
        const int img = row / IMG_ROWS, rr = row - img * IMG_ROWS;
        const int hy = rr / PITCH, hx = rr - hy * PITCH;
        const int c = (lane & 7) ^ (((hx >> 1) & 3) | ((hy & 1) << 2));
        const int iy = iy0 + hy, ix = ix0 + hx, ni = n * 4 + img;
        const bool ok = img < 4 && ni < a.N && (unsigned)iy < (unsigned)a.Hin &&
                        (unsigned)ix < (unsigned)a.Win;
        hvoff[j] = ok ? (uint32_t)((((ni * a.Hin + iy) * a.Win + ix) * a.Ci + c * 8) * 2) : HC_OOB;
        hc8[j] = c * 8;
        continue;
      }
      const int hy = row / PITCH, hx = row - hy * PITCH;
      const int c = (lane & 7) ^ ((hx >> 1) & 7);
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = hy < HH && hx < HWID && (unsigned)iy < (unsigned)a.Hin &&
                      (unsigned)ix < (unsigned)a.Win;
      const int us = a.in_up;   // 0 / 1: logical pixel (iy, ix) lives at (iy >> us, ix >> us)
      hvoff[j] = ok ? (uint32_t)((((n * (a.Hin >> us) + (iy >> us)) * (a.Win >> us) + (ix >> us)) *
                                      a.Ci + c * 8) * 2)
                    : HC_OOB;
      hc8[j] = c * 8;
    }
  }
  uint32_t bvoff[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int row = (wave * BJ + j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    bvoff[j] = (n0 + row) < a.Co ? (uint32_t)(((n0 + row) * a.Kp + c * 8) * 2) : HC_OOB;
  }
  auto issue_halo = [&](int cb) {
    const int crem = a.Ci - cb * 64;   // channels left in this block (32 in a ragged last block)
#pragma unroll
    for (int j = 0; j < HSLOTS; ++j)
      if (wave + 8 * j < npieces) {
        const uint32_t vo = (ragged_ci && hc8[j] >= crem) ? HC_OOB : hvoff[j];
        hc_dma16(rs_in, vo, (uint32_t)(cb * 128), smem + (wave + 8 * j) * 1024);
      }
  };
  auto issue_b = [&](int slot, int koff) {
#pragma unroll
    for (int j = 0; j < BJ; ++j)
      hc_dma16(rs_bt, bvoff[j], (uint32_t)(koff * 2),
               smem + HALO_BYTES + slot * B_BYTES + (wave * BJ + j) * 1024);
  };

  // fused batch-norm prologue: per channel block, the coefficients of its 64 channels go to LDS ...
  const bool bnp = (FUSE == 1 || FUSE == 3) && a.bn_mean != nullptr;   // wave-uniform
  // the block's coefficient rows go to LDS by 4-byte LDS-DMAs of wave 0 (mean | variance | gamma |
  // beta, 64 floats each): with VGPR-destination loads hipcc waits vmcnt(0) where the values are used,
  // i.e. for the whole window DMA issued around them (r06: +3 k cycles per channel block); absent
  // gamma / beta rows are written once as 1 / 0
  auto fetch_bn_table = [&](int cb) {
    if (bnp && wave == 0) {
      float* tab = reinterpret_cast<float*>(smem + TAB_OFF);
      const int ch = min(cb * 64 + lane, a.Ci - 1);   // (a ragged last block only uses its first half)
      const int64_t pidx = a.bn_per_sample ? (int64_t)n * a.Ci + ch : ch;
      const int64_t sidx = a.bn_stat_group > 0 ? (int64_t)(n / a.bn_stat_group) * a.Ci + ch : ch;
      hc_glds4(a.bn_mean + sidx, tab);
      hc_glds4(a.bn_var + sidx, tab + 64);
      if (a.bn_gamma) hc_glds4(a.bn_gamma + pidx, tab + 128);
      if (a.bn_beta) hc_glds4(a.bn_beta + pidx, tab + 192);
    }
  };
  if (bnp && tid < 64) {
    float* tab = reinterpret_cast<float*>(smem + TAB_OFF);
    if (!a.bn_gamma) tab[128 + tid] = 1.f;
    if (!a.bn_beta) tab[192 + tid] = 0.f;
  }
  // ... and the staged window is normalised in place (same operation order as cg_bn_apply,
  // arch_ops.py:306-312); padding pixels stay zero: the padding applies to the BN output
  auto bn_transform = [&](int cb) {
    const float* tab = reinterpret_cast<const float*>(smem + TAB_OFF);
    const int crem = a.Ci - cb * 64;
    if constexpr (BN == 64) {
      // a thread keeps ONE 8-channel group (tid & 7) for all its rows, so the 32 coefficients come
      // out of the LDS table once per block instead of once per chunk (32 of the ~36 LDS operations
      // a chunk cost); the chunk of that group in row r sits at slot group ^ ((hx >> 1) & 7).
      // Only where the registers are free: the 128-channel form (126 VGPRs) would spill.
      const int c8 = (tid & 7) * 8;
      if (c8 >= crem) return;   // zero-filled half of a ragged last channel block
      float cm[8], cr[8], cg[8], cbt[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        cm[e] = tab[c8 + e];
        cr[e] = rsqrtf(tab[64 + c8 + e] + a.bn_eps);
        cg[e] = tab[128 + c8 + e];
        cbt[e] = tab[192 + c8 + e];
      }
      // batches of three rows: all LDS reads of a batch are in flight before the first is used (one
      // row per iteration left a read -> 40 VALU -> write chain of ~400 cycles per row exposed)
      constexpr int NIT = (HALO_PIECES * 8 + 63) / 64, NB = 3;
#pragma unroll
      for (int b0 = 0; b0 < NIT; b0 += NB) {
        uint4* pp[NB];
        uint4 raw[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int row = (tid >> 3) + 64 * (b0 + b);
          const int hy = row / PITCH, hx = row - hy * PITCH;
          const int iy = iy0 + hy, ix = ix0 + hx;
          const bool ok = b0 + b < NIT && row < npieces * 8 && hy < HH && hx < HWID &&
                          (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
          pp[b] = ok ? reinterpret_cast<uint4*>(smem + (row * 8 + ((tid & 7) ^ ((hx >> 1) & 7))) * 16)
                     : nullptr;
          if (pp[b]) raw[b] = *pp[b];
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (!pp[b]) continue;
          float v[8];
          unpack8_bf16(raw[b], v);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t = (v[e] - cm[e]) * cr[e];
            t = t * cg[e] + cbt[e];
            v[e] = fmaxf(t, 0.f);
          }
          *pp[b] = pack8_bf16(v);
        }
      }
      return;
    }
    // 128-channel tile: the same thread -> (8-channel group, rows) map, in two passes of FOUR channels
    // (16 coefficient registers; the accumulators leave no room for 32) with 8-byte LDS accesses; the
    // round-5 form read the 4 x 8 coefficients of every chunk from the LDS table (8 of its 10 LDS
    // operations per chunk)
    const int c8 = (tid & 7) * 8;
    if (c8 >= crem) return;   // zero-filled half of a ragged last channel block
    constexpr int NIT = (HALO_PIECES * 8 + 63) / 64, NB = 2;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      float cm[4], cr[4], cg[4], cbt[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        cm[e] = tab[c8 + 4 * h + e];
        cr[e] = rsqrtf(tab[64 + c8 + 4 * h + e] + a.bn_eps);
        cg[e] = tab[128 + c8 + 4 * h + e];
        cbt[e] = tab[192 + c8 + 4 * h + e];
      }
#pragma unroll
      for (int b0 = 0; b0 < NIT; b0 += NB) {
        int pp[NB];   // LDS byte offset of the row's half chunk (-1: padding / outside)
        uint2 raw[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int row = (tid >> 3) + 64 * (b0 + b);
          const int hy = row / PITCH, hx = row - hy * PITCH;
          const int iy = iy0 + hy, ix = ix0 + hx;
          const bool ok = b0 + b < NIT && row < npieces * 8 && hy < HH && hx < HWID &&
                          (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
          pp[b] = ok ? (row * 8 + ((tid & 7) ^ ((hx >> 1) & 7))) * 16 + 8 * h : -1;
          if (pp[b] >= 0) raw[b] = *reinterpret_cast<const uint2*>(smem + pp[b]);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (pp[b] < 0) continue;
          float v[4] = {__uint_as_float(raw[b].x << 16), __uint_as_float(raw[b].x & 0xffff0000u),
                        __uint_as_float(raw[b].y << 16), __uint_as_float(raw[b].y & 0xffff0000u)};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = (v[e] - cm[e]) * cr[e];
            t = t * cg[e] + cbt[e];
            v[e] = fmaxf(t, 0.f);
          }
          *reinterpret_cast<uint2*>(smem + pp[b]) = pack4_bf16(v);
        }
      }
    }
  };

  // the first loads leave before the rest of the set-up (their latency is the longest pole)
  issue_halo(0);
  issue_b(0, ((r0 * a.kw) + s0) * a.Ci);
  fetch_bn_table(0);

  // ---- fragment addressing ----
  // pixel p = wm*64 + i*32 + frow of the tile -> (y, x); halo row of its tap-(0,0) input pixel
  int hb[2], hx0[2], hy0[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = wm * 64 + i * 32 + frow;
    const int y = MI ? (p >> 3) & 7 : p >> TWL, x = p & (TW - 1);
    hb[i] = ((MI ? (p >> 6) * IMG_ROWS : 0) + y * PITCH + x) * 128;
    hx0[i] = x;
    hy0[i] = y;
  }
  // weights: row = wn*(BN/2) + j*32 + frow, chunk (kk*2 + half) ^ ((frow >> 1) & 7)
  int bko[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
    bko[kk] = ((NARROW ? 0 : wn * (BN / 2)) + frow) * 128 +
              ((((NARROW ? 2 * wn + kk : kk) * 2 + half) ^ ((frow >> 1) & 7)) << 4);

  // 32-channel tiles of this wave with real output channels (wave-uniform; Co % 8 == 0)
  const int cleft = a.Co - n0 - wn * (BN / 2);
  const int jn = NARROW ? 1 : (cleft <= 0 ? 0 : (cleft >= 32 * TN ? TN : (cleft + 31) >> 5));

  f32x16_t acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  // ---- main loop: K-slice = (channel block, tap); the halo is staged once per channel block ----
  HC_STAMP(1);
  int tap = 0, cb = 0, ri = 0, si = 0;
  // one K-slice.  JN: 32-channel MFMA tiles of this wave that hold real output channels; NKQ:
  // 16-channel k-steps of the slice that hold real input channels (2 in the ragged last block of
  // Ci % 64 == 32).  Both are chosen OUTSIDE the loop (run_slices below): a wave-uniform branch
  // around the MFMAs inside it makes hipcc copy the accumulator tuples at the merge points and
  // spill (measured: 10-20x slower).
  auto slice = [&](int it, auto jn_c, auto nkq_c) {
    // this wave's pieces of slice `it` (and of the halo, on a block's first tap) have landed; after
    // the barrier so have everybody's, and every wave is done with the weight slot restaged below
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (RELU && tap == 0) {
      // ReLU of the input (resnet_ops.py:165,175): ONCE per staged window, in LDS, by the wave that
      // staged the piece (its own pieces have landed), instead of on every one of the 9 x 8 fragment
      // reads that consume the window (2 VALU instructions per MFMA in an issue-bound loop)
#pragma unroll
      for (int j = 0; j < HSLOTS; ++j)
        if (wave + 8 * j < npieces) {
          bf16x8_t* p = reinterpret_cast<bf16x8_t*>(smem + (wave + 8 * j) * 1024 + lane * 16);
          *p = hc_relu(*p);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (bnp && tap == 0) {
      __syncthreads();   // window, weights and coefficient table are in LDS
      bn_transform(cb);
      __syncthreads();
    } else {
      asm volatile("s_barrier" ::: "memory");
    }
    if (it == 0) HC_STAMP(2);
    int ntap = tap + 1, ncb = cb, nri = ri, nsi = si + 1;
    if (nsi == ns) {
      nsi = 0;
      ++nri;
    }
    if (ntap == ntaps) {
      ntap = 0;
      nri = 0;
      nsi = 0;
      ncb = cb + 1;
    }
    if (it + 1 < nk)
      issue_b((it + 1) & 1, ((r0 + a.U * nri) * a.kw + (s0 + a.U * nsi)) * a.Ci + ncb * 64);

    const unsigned char* Bs = smem + HALO_BYTES + (it & 1) * B_BYTES;
    const int tshift = (ri * PITCH + si) * 128;
    int abase[2], aswz[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      abase[i] = hb[i] + tshift;
      aswz[i] = MI ? ((((hx0[i] + si) >> 1) & 3) | (((hy0[i] + ri) & 1) << 2))
                   : (((hx0[i] + si) >> 1) & 7);
    }
    {
      constexpr int JN = decltype(jn_c)::value, NKQ = decltype(nkq_c)::value;
#pragma unroll
      for (int kq = 0; kq < NKQ; ++kq) {
        const int kk = kq;                             // index of bko[]
        const int ka = NARROW ? 2 * wn + kq : kq;      // 16-channel step within the slice
        bf16x8_t af[2], bfr[JN > 0 ? JN : 1];
        if constexpr (JN > 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            af[i] = *reinterpret_cast<const bf16x8_t*>(smem + abase[i] +
                                                       (((ka * 2 + half) ^ aswz[i]) << 4));
          }
#pragma unroll
          for (int j = 0; j < JN; ++j)
            bfr[j] = *reinterpret_cast<const bf16x8_t*>(Bs + bko[kk] + j * 32 * 128);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < JN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
      }
    }
    if (ntap == 0 && ncb < a.cblocks) {
      // channel block finished: every wave is done with the halo image before it is overwritten
      asm volatile("s_barrier" ::: "memory");
      issue_halo(ncb);
      fetch_bn_table(ncb);
    }
    tap = ntap;
    cb = ncb;
    ri = nri;
    si = nsi;
  };
  using HI0 = std::integral_constant<int, 0>;
  using HI1 = std::integral_constant<int, 1>;
  using HI2 = std::integral_constant<int, 2>;
  using HI4 = std::integral_constant<int, 4>;
  using HITN = std::integral_constant<int, TN>;
  // the slices of the ragged last channel block (if any) are the last `ntaps` ones
  const int nk_full = (!NARROW && ragged_ci) ? nk - ntaps : nk;
  auto run_slices = [&](auto jn_c) {
    int it = 0;
    if constexpr (NARROW) {
      for (; it < nk; ++it) slice(it, HI1(), HI2());
    } else {
      for (; it < nk_full; ++it) slice(it, jn_c, HI4());
      for (; it < nk; ++it) slice(it, jn_c, HI2());
    }
  };
  if (jn == TN) run_slices(HITN());
  else if (TN == 2 && jn == 1) run_slices(HI1());
  else run_slices(HI0());

  HC_STAMP(3);
  // ---- epilogue: every wave stages its own 64-pixel x WCO-channel accumulator tile through a
  // private LDS region (fp32, two passes of 32 pixels), then each lane finishes 8 consecutive channels
  // of one pixel (bias, activation, gate, residual: one rounding to bf16) and writes 16 (bf16) / 32
  // (fp32) contiguous bytes; 8 (4) lanes cover a pixel's WCO channels.  One workgroup barrier (the
  // halo / weight images are dead), nothing block-wide after it: LDS operations of one wave complete
  // in order.
  constexpr int WCO = BN / 2;            // channels per wave
  constexpr int SP = WCO * 4 + 16;       // staging row pitch in bytes (+16: conflict-free b128 writes)
  constexpr int G8 = WCO / 8;            // 8-channel groups per row
  constexpr int RPI = 64 / G8;           // rows per sweep of the 64 lanes
  static_assert(8 * 32 * SP + 8 * 2 * WCO * 4 <= LDS_BYTES, "per-wave epilogue staging does not fit");
  unsigned char* Sw = smem + wave * (32 * SP);
  const int g8 = lane & (G8 - 1), rl = lane / G8;
  const int co = n0 + wn * WCO + g8 * 8;
  const bool co_ok = co < a.Co;          // Co % 8 == 0
  float bv[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = s1[e] = s2[e] = 0.f;
  if constexpr (NARROW) {
    // ---- narrow epilogue (Co < 8).  The two channel-half waves of a pixel group hold partial sums
    // (over their halves of every 64-channel slice) of the same 32 accumulator rows, of which rows
    // 0 .. Co-1 are real: lanes 0-31 hold rows 0-3 of pixel `frow` in acc[.][0][0..3], lanes 32-63
    // rows 4-7.  The odd wave hands its four values over through LDS, the even wave adds, finishes
    // (bias, leaky self-gate) and stores Co scalars per pixel.
    __syncthreads();   // the halo / weight images are dead
    float4* X = reinterpret_cast<float4*>(smem);   // [4 wm][2 i][64 lanes]
    if (wn == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        X[(wm * 2 + i) * 64 + lane] = make_float4(acc[i][0][0], acc[i][0][1], acc[i][0][2], acc[i][0][3]);
    }
    __syncthreads();
    if (wn == 0) {
      const float osc = a.out_scale;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 o4 = X[(wm * 2 + i) * 64 + lane];
        float v[4] = {acc[i][0][0] + o4.x, acc[i][0][1] + o4.y, acc[i][0][2] + o4.z, acc[i][0][3] + o4.w};
        const int p = wm * 64 + i * 32 + frow;
        const int y = p >> TWL, x = p & (TW - 1);
        const int oy = ty * TH + y, ox = tx * TW + x;   // (U == 1)
        const int64_t o = ((int64_t)(n * a.Ho + oy) * a.Wo + ox) * a.Co;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = half * 4 + r;
          if (c < a.Co) {
            float t = v[r] * osc + (a.bias ? a.bias[c] : 0.f);
            if (a.self_gate && !(t > 0.f)) t *= a.slope_out;
            if (a.out_f32) reinterpret_cast<float*>(a.out)[o + c] = t;
            else reinterpret_cast<bf16_t*>(a.out)[o + c] = f2bf(t);
          }
        }
      }
    }
    HC_STAMP(4);
    return;
  }
  if (a.bias && co_ok) {
    const float4 b0 = *reinterpret_cast<const float4*>(a.bias + co);
    const float4 b1 = *reinterpret_cast<const float4*>(a.bias + co + 4);
    bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
    bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
  }
  __syncthreads();
  const float osc = a.out_scale;
  if constexpr (FUSE == 2) {
    // ---- pooled epilogue.  The wave's 64 pixels are two tile rows (8x32 tiles: rows wm*2 + i, so
    // vertical neighbours are acc[0] / acc[1] of the SAME lane: summed in registers, one staging
    // pass) or four half rows (16x16 tiles: staged rows r and r + 16 of a pass are vertical
    // neighbours).  Every lane then finishes pooled pixels from 2 (4) staged rows: no cross-lane
    // traffic.
    constexpr int NPASS = TWL == 5 ? 1 : 2;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 t = make_float4(acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2],
                                 acc[i][j][q * 4 + 3]);
          if (TWL == 5) {
            t.x += acc[1][j][q * 4 + 0]; t.y += acc[1][j][q * 4 + 1];
            t.z += acc[1][j][q * 4 + 2]; t.w += acc[1][j][q * 4 + 3];
          }
          *reinterpret_cast<float4*>(Sw + frow * SP + (j * 32 + q * 8 + 4 * half) * 4) = t;
        }
      __builtin_amdgcn_wave_barrier();
      // items: (pooled pixel x2, channel group); 8x32: 16 x G8 per pass, 16x16: 8 x G8 per pass
      constexpr int NPX = TWL == 5 ? 16 : 8;
#pragma unroll
      for (int it = lane; it < NPX * G8; it += 64) {
        const int x2 = it / G8, gg = it - x2 * G8;
        const unsigned char* r0 = Sw + (2 * x2) * SP + gg * 32;
        float v[8];
        {
          const float4 a0 = *reinterpret_cast<const float4*>(r0);
          const float4 a1 = *reinterpret_cast<const float4*>(r0 + 16);
          const float4 b0 = *reinterpret_cast<const float4*>(r0 + SP);
          const float4 b1 = *reinterpret_cast<const float4*>(r0 + SP + 16);
          v[0] = a0.x + b0.x; v[1] = a0.y + b0.y; v[2] = a0.z + b0.z; v[3] = a0.w + b0.w;
          v[4] = a1.x + b1.x; v[5] = a1.y + b1.y; v[6] = a1.z + b1.z; v[7] = a1.w + b1.w;
        }
        if (TWL == 4) {
          const float4 a0 = *reinterpret_cast<const float4*>(r0 + 16 * SP);
          const float4 a1 = *reinterpret_cast<const float4*>(r0 + 16 * SP + 16);
          const float4 b0 = *reinterpret_cast<const float4*>(r0 + 17 * SP);
          const float4 b1 = *reinterpret_cast<const float4*>(r0 + 17 * SP + 16);
          v[0] += a0.x + b0.x; v[1] += a0.y + b0.y; v[2] += a0.z + b0.z; v[3] += a0.w + b0.w;
          v[4] += a1.x + b1.x; v[5] += a1.y + b1.y; v[6] += a1.z + b1.z; v[7] += a1.w + b1.w;
        }
        const int cq = n0 + wn * WCO + gg * 8;
        if (cq >= a.Co) continue;
        const int y2 = TWL == 4 ? wm * 2 + i : wm;   // pooled row / column within the tile
        const int oy = ((ty * TH) >> 1) + y2, ox = ((tx * TW) >> 1) + x2;
        const int64_t o = ((int64_t)(n * (a.Ho >> 1) + oy) * (a.Wo >> 1) + ox) * a.Co + cq;
        if (a.bias) {
          const float4 b0 = *reinterpret_cast<const float4*>(a.bias + cq);
          const float4 b1 = *reinterpret_cast<const float4*>(a.bias + cq + 4);
          v[0] = 0.25f * v[0] + b0.x; v[1] = 0.25f * v[1] + b0.y; v[2] = 0.25f * v[2] + b0.z;
          v[3] = 0.25f * v[3] + b0.w; v[4] = 0.25f * v[4] + b1.x; v[5] = 0.25f * v[5] + b1.y;
          v[6] = 0.25f * v[6] + b1.z; v[7] = 0.25f * v[7] + b1.w;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= 0.25f;
        }
        if (a.residual) {
          float rv[8];
          unpack8_bf16(*reinterpret_cast<const uint4*>(a.residual + o), rv);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rv[e];
        }
        if (a.out_f32) {
          float* op = reinterpret_cast<float*>(a.out) + o;
          *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.out) + o) = pack8_bf16(v);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    HC_STAMP(4);
    return;
  }
  // row `rl + RPI k` of pass i: output offset of this lane's 8 channels (-1: nothing to store)
  constexpr int NKR = 32 / RPI;
  auto row_off = [&](int i, int k) -> int64_t {
    const int row = rl + RPI * k;
    const int p = wm * 64 + i * 32 + row;
    const int y = MI ? (p >> 3) & 7 : p >> TWL, x = p & (TW - 1);
    const int oy = (ty * TH + y) * a.U + ph, ox = (tx * TW + x) * a.U + pw;
    const int ni = MI ? n * 4 + (p >> 6) : n;
    if (!co_ok || (MI && ni >= a.N)) return -1;   // (ragged last image group)
    return ((int64_t)(ni * a.Ho + oy) * a.Wo + ox) * a.Co + co;
  };
  // the gate tensor (data gradients: every launch of a backward pass) or the residual of a pass's rows
  // is requested BEFORE the accumulators of the pass go through LDS: a load issued where its value is
  // used costs its whole latency per row, four to eight times per workgroup (r06_hconv_timeline.txt:
  // epilogue 3.0 k -> 8.5 k cycles with a residual).  One buffer: with both tensors present the gate is
  // prefetched and the residual read late.
  const bf16_t* pf = a.gate_out ? a.gate_out : a.residual;   // workgroup-uniform
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    uint4 pre[NKR];
    if (pf) {
#pragma unroll
      for (int k = 0; k < NKR; ++k) {
        const int64_t o = row_off(i, k);
        pre[k] = *reinterpret_cast<const uint4*>(pf + (o < 0 ? 0 : o));
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(Sw + frow * SP + (j * 32 + q * 8 + 4 * half) * 4) =
            make_float4(acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2],
                        acc[i][j][q * 4 + 3]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < NKR; ++k) {
      const int row = rl + RPI * k;
      const float4 lo = *reinterpret_cast<const float4*>(Sw + row * SP + g8 * 32);
      const float4 hi = *reinterpret_cast<const float4*>(Sw + row * SP + g8 * 32 + 16);
      const int64_t o = row_off(i, k);
      if (o < 0) continue;
      float v[8] = {lo.x * osc + bv[0], lo.y * osc + bv[1], lo.z * osc + bv[2], lo.w * osc + bv[3],
                    hi.x * osc + bv[4], hi.y * osc + bv[5], hi.z * osc + bv[6], hi.w * osc + bv[7]};
      if (a.self_gate) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (!(v[e] > 0.f)) v[e] *= a.slope_out;
      }
      if (a.gate_out) {
        float gv[8];
        unpack8_bf16(pre[k], gv);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (!(gv[e] > 0.f)) v[e] *= a.slope_out;
      }
      if (a.residual) {
        float rv[8];
        unpack8_bf16(a.gate_out ? *reinterpret_cast<const uint4*>(a.residual + o) : pre[k], rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      if (a.out_f32) {
        float* op = reinterpret_cast<float*>(a.out) + o;
        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        const uint4 pk = pack8_bf16(v);
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.out) + o) = pk;
        if (FUSE == 1 && a.stats) unpack8_bf16(pk, v);   // statistics of the STORED values
      }
      if (FUSE == 1 && a.stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s1[e] += v[e];
          s2[e] += v[e] * v[e];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (FUSE == 1 && a.stats) {   // wave-uniform
    // lane = rl * G8 + g8: the lanes of one 16-lane DPP row with the same g8 (16 / G8 of them) are
    // summed by row rotations (one VALU instruction each, no LDS round trip: the round-5 butterfly was
    // 64 ds_bpermute per wave); the 4 rows of a wave and the 4 pixel-waves of a channel half are then
    // combined through LDS in a fixed order.  A wave's partials go into its OWN staging rows (its
    // LDS operations complete in order).
#pragma unroll
    for (int r = G8; r < 16; r <<= 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] += hc_row_ror(s1[e], r);
        s2[e] += hc_row_ror(s2[e], r);
      }
    }
    float* sreg = reinterpret_cast<float*>(Sw);   // [4 DPP rows][2][WCO]
    if ((lane & 15) < G8) {
      const int dr = lane >> 4;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sreg[(dr * 2 + 0) * WCO + g8 * 8 + e] = s1[e];
        sreg[(dr * 2 + 1) * WCO + g8 * 8 + e] = s2[e];
      }
    }
    __syncthreads();
    if (tid < BN) {
      const int cw = tid / WCO, cc = tid - cw * WCO;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4) {
        const float* sw = reinterpret_cast<const float*>(smem + (cw * 4 + m4) * (32 * SP));   // wave (wm = m4, wn = cw)
#pragma unroll
        for (int dr = 0; dr < 4; ++dr) {
          t1 += sw[(dr * 2 + 0) * WCO + cc];
          t2 += sw[(dr * 2 + 1) * WCO + cc];
        }
      }
      const int cch = n0 + tid;
      if (cch < a.Co) {
        const int64_t row = (int64_t)blockIdx.y * (gridDim.x / a.ntiles) + st;
        a.stats[row * 2 * a.Co + cch] = t1;
        a.stats[row * 2 * a.Co + a.Co + cch] = t2;
      }
    }
  }
  HC_STAMP(4);
#ifdef CG_CONV_TIMING
  if (a.tdbg && tid == 0)
    a.tdbg[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + 6] = __builtin_amdgcn_s_memrealtime();
#endif
}


// -------------------------------------------------------------------------------------------
// 64 -> 64 channel 3x3 convolutions (ResNet5 D block B0 / G block B5 at 128x128: 15 % of the D-step
// FLOPs).  With one 64-channel block the K loop of hconv_kernel is only 9 slices, and its per-
// workgroup set-up, first-load latency and epilogue cost as much as the loop (workgroup timeline in
// profiles/r02_hconv_timeline.txt: 13 k of 26 k cycles).  Here the whole weight panel of a wave
// (9 taps x 64 k x 32 out-channels = 36 KiB -> 144 VGPRs) stays in registers, so nothing but the
// input window is staged: persistent 4-wave workgroups (2 per CU) walk 128-pixel tiles (4 x 32),
// the next tile's window (26 KiB) is in flight while the current one is multiplied, one barrier per
// tile, and the per-wave epilogue of one workgroup overlaps the MFMA work of the other.
// -------------------------------------------------------------------------------------------
constexpr int RW_PIECES = 26;             // 6 x 34 = 204 halo rows -> 26 pieces of 8 rows
constexpr int RW_HB = RW_PIECES * 1024;   // one window buffer
constexpr int RW_SLOTS = 7;               // pieces per wave (4 waves)

// POOL: 2x2 average pooling in the epilogue (the wave's two tile rows are vertical neighbours:
// acc[0] + acc[1] in registers, horizontal pairs from the staging rows); a.in_up: the input is the
// pooled-resolution tensor read through its nearest-neighbour up-sampling (data gradient of a pooled
// convolution, with a.out_scale = 1/4) -- the two forms of hconv_kernel<64, *, 5, 2>
template <bool RELU, bool POOL>
__global__ __launch_bounds__(256, 2) void hconv_rw_kernel(HConvArgs a) {
  constexpr int TW = 32, TH = 4, PITCH = TW + 2, HROWS = (TH + 2) * PITCH;
  constexpr int SP = 32 * 4 + 16;         // epilogue staging row pitch (32 channels fp32 + pad)
  constexpr int STG = 4 * 32 * SP;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * RW_HB + STG];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31, half = lane >> 5;

  // ---- this wave's weight panel: out-channels wn*32 + frow, all 9 taps x 64 k ----
  bf16x8_t bw[9][4];
  {
    const bf16_t* wp = a.bt + (int64_t)(wn * 32 + frow) * a.Kp + half * 8;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        bw[t][kk] = *reinterpret_cast<const bf16x8_t*>(wp + t * 64 + kk * 16);
  }
  // ---- window staging descriptors, relative to the tile's window origin ----
  uint32_t hrel[RW_SLOTS];
  int hyx[RW_SLOTS];
#pragma unroll
  for (int j = 0; j < RW_SLOTS; ++j) {
    const int row = (wave + 4 * j) * 8 + (lane >> 3);
    const int hy = row / PITCH, hx = row - hy * PITCH;
    const int c = (lane & 7) ^ ((hx >> 1) & 7);
    // in_up: window pixel (iy0 + hy, ix0 + hx) with iy0, ix0 odd lives at ((iy0 + hy) >> 1, ...) of
    // the half-resolution tensor = (hy + 1) >> 1 rows below the row of iy0 (see stage())
    hrel[j] = a.in_up ? (uint32_t)(((((hy + 1) >> 1) * (a.Win >> 1) + ((hx + 1) >> 1)) * 64 + c * 8) * 2)
                      : (uint32_t)(((hy * a.Win + hx) * 64 + c * 8) * 2);
    hyx[j] = row < HROWS ? (hy | (hx << 16)) : 0x7fff7fff;
  }
  auto stage = [&](int buf, int t) {
    const int q = (int)fdiv((uint32_t)t, a.dTx);
    const int tx = t - q * a.tiles_x;
    const int n = (int)fdiv((uint32_t)q, a.dTy);
    const int ty = q - n * a.tiles_y;
    const int iy0 = ty * TH - a.pt, ix0 = tx * TW - a.pl;   // pt = pl = 1: both odd
    const bf16_t* xo =
        a.in_up ? a.in + (((int64_t)n * (a.Hin >> 1) + ((iy0 - 1) >> 1)) * (a.Win >> 1) +
                          ((ix0 - 1) >> 1)) * 64
                : a.in + (((int64_t)n * a.Hin + iy0) * a.Win + ix0) * 64;
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc((void*)xo, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int j = 0; j < RW_SLOTS; ++j) {
      if (wave + 4 * j < RW_PIECES) {
        const int hy = hyx[j] & 0xffff, hx = hyx[j] >> 16;
        const bool ok = (unsigned)(iy0 + hy) < (unsigned)a.Hin &&
                        (unsigned)(ix0 + hx) < (unsigned)a.Win;
        hc_dma16(rx, ok ? hrel[j] : HC_OOB, 0, smem + buf * RW_HB + (wave + 4 * j) * 1024);
      }
    }
  };
  // ---- fragment addressing: pixel (y = wm*2 + i, x = frow); chunk swizzle by halo column ----
  int colx[3][4];
#pragma unroll
  for (int si = 0; si < 3; ++si)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      colx[si][kk] = ((kk * 2 + half) ^ (((frow + si) >> 1) & 7)) << 4;
  const int hb0 = ((wm * 2) * PITCH + frow) * 128;

  // ---- epilogue constants (per-wave staging as in hconv_kernel: 32 channels per wave) ----
  constexpr int G8 = 4, RPI = 16;
  unsigned char* Sw = smem + 2 * RW_HB + wave * (32 * SP);
  const int g8 = lane & (G8 - 1), rl = lane / G8;
  const int co = wn * 32 + g8 * 8;
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = 0.f;
  if (a.bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(a.bias + co);
    const float4 b1 = *reinterpret_cast<const float4*>(a.bias + co + 4);
    bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
    bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
  }

  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  int t = blockIdx.x;
#ifdef CG_CONV_TIMING
  unsigned long long tq_wait = 0, tq_mma = 0, tq_epi = 0, tq_t0 = __builtin_amdgcn_s_memtime(), tq_a, tq_b;
#define RW_T(x) x = __builtin_amdgcn_s_memtime()
#else
#define RW_T(x) do {} while (0)
#endif
  if (t < ntiles) stage(0, t);
  for (int it = 0; t < ntiles; t += gridDim.x, ++it) {
    const int buf = it & 1;
    RW_T(tq_a);
    // the window DMA of this tile was issued BEFORE the previous tile's output stores; vmcnt counts
    // stores too and retires in order, so leaving exactly those stores outstanding (4 bf16 / 8 fp32
    // store instructions per wave and tile) waits for the window without draining the stores
    if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (POOL) {   // one pooled pixel x 8 channels per lane: 1 bf16 / 2 fp32 store instructions
      if (a.out_f32) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else if (a.out_f32) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");   // window landed; everyone is done with the other buffer
    // pooled form: this lane's residual leaves BEFORE the next window's DMA pieces, so that waiting
    // for it in the epilogue (vmcnt retires in order) does not drain them
    uint4 res_pool = make_uint4(0u, 0u, 0u, 0u);
    if (POOL && a.residual) {
      const int q0 = (int)fdiv((uint32_t)t, a.dTx);
      const int tx0 = t - q0 * a.tiles_x;
      const int n0 = (int)fdiv((uint32_t)q0, a.dTy);
      const int ty0 = q0 - n0 * a.tiles_y;
      const int oy = ((ty0 * TH) >> 1) + wm, ox = ((tx0 * TW) >> 1) + rl;
      res_pool = *reinterpret_cast<const uint4*>(
          a.residual + ((int64_t)(n0 * (a.Ho >> 1) + oy) * (a.Wo >> 1) + ox) * a.Co + co);
    }
    if (t + (int)gridDim.x < ntiles) stage(buf ^ 1, t + gridDim.x);
    const unsigned char* Hb = smem + buf * RW_HB + hb0;
#ifdef CG_CONV_TIMING
    RW_T(tq_b); tq_wait += tq_b - tq_a;
#endif

    f32x16_t acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ri = tap / 3, si = tap % 3;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(
              Hb + ((i + ri) * PITCH + si) * 128 + colx[si][kk]);
          if (RELU) af = hc_relu(af);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[tap][kk], af, acc[i], 0, 0, 0);
        }
      }
      // fragment reads are not hoisted across taps: 144 of the 256 registers hold the weights
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef CG_CONV_TIMING
    RW_T(tq_a); tq_mma += tq_a - tq_b;
#endif
    // ---- epilogue of this tile (wave-private staging: no workgroup barrier) ----
    const int q = (int)fdiv((uint32_t)t, a.dTx);
    const int tx = t - q * a.tiles_x;
    const int n = (int)fdiv((uint32_t)q, a.dTy);
    const int ty = q - n * a.tiles_y;
    if constexpr (POOL) {
      // vertical pairs in registers, horizontal pairs from staging rows 2 x2 / 2 x2 + 1; lane ->
      // (pooled column x2 = lane / 4, channel group g8): bias before the pooling average is the same
      // as after it, the residual lives at the pooled resolution
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
        *reinterpret_cast<float4*>(Sw + frow * SP + (qq * 8 + 4 * half) * 4) =
            make_float4(acc[0][qq * 4 + 0] + acc[1][qq * 4 + 0], acc[0][qq * 4 + 1] + acc[1][qq * 4 + 1],
                        acc[0][qq * 4 + 2] + acc[1][qq * 4 + 2], acc[0][qq * 4 + 3] + acc[1][qq * 4 + 3]);
      __builtin_amdgcn_wave_barrier();
      const int x2 = rl;   // 0..15
      const unsigned char* r0 = Sw + (2 * x2) * SP + g8 * 32;
      const float4 a0 = *reinterpret_cast<const float4*>(r0);
      const float4 a1 = *reinterpret_cast<const float4*>(r0 + 16);
      const float4 b0 = *reinterpret_cast<const float4*>(r0 + SP);
      const float4 b1 = *reinterpret_cast<const float4*>(r0 + SP + 16);
      float v[8] = {0.25f * (a0.x + b0.x) + bv[0], 0.25f * (a0.y + b0.y) + bv[1],
                    0.25f * (a0.z + b0.z) + bv[2], 0.25f * (a0.w + b0.w) + bv[3],
                    0.25f * (a1.x + b1.x) + bv[4], 0.25f * (a1.y + b1.y) + bv[5],
                    0.25f * (a1.z + b1.z) + bv[6], 0.25f * (a1.w + b1.w) + bv[7]};
      const int oy = ((ty * TH) >> 1) + wm, ox = ((tx * TW) >> 1) + x2;
      const int64_t o = ((int64_t)(n * (a.Ho >> 1) + oy) * (a.Wo >> 1) + ox) * a.Co + co;
      if (a.residual) {
        float rv[8];
        unpack8_bf16(res_pool, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      if (a.out_f32) {
        float* op = reinterpret_cast<float*>(a.out) + o;
        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.out) + o) = pack8_bf16(v);
      }
      __builtin_amdgcn_wave_barrier();
#ifdef CG_CONV_TIMING
      RW_T(tq_b); tq_epi += tq_b - tq_a;
#endif
      continue;
    }
    const float osc = a.out_scale;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
        *reinterpret_cast<float4*>(Sw + frow * SP + (qq * 8 + 4 * half) * 4) =
            make_float4(acc[i][qq * 4 + 0], acc[i][qq * 4 + 1], acc[i][qq * 4 + 2],
                        acc[i][qq * 4 + 3]);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 32 / RPI; ++k) {
        const int row = rl + RPI * k;   // x within the tile row
        const float4 lo = *reinterpret_cast<const float4*>(Sw + row * SP + g8 * 32);
        const float4 hi = *reinterpret_cast<const float4*>(Sw + row * SP + g8 * 32 + 16);
        const int oy = ty * TH + wm * 2 + i, ox = tx * TW + row;
        const int64_t o = ((int64_t)(n * a.Ho + oy) * a.Wo + ox) * a.Co + co;
        float v[8] = {lo.x * osc + bv[0], lo.y * osc + bv[1], lo.z * osc + bv[2], lo.w * osc + bv[3],
                      hi.x * osc + bv[4], hi.y * osc + bv[5], hi.z * osc + bv[6], hi.w * osc + bv[7]};
        if (a.self_gate) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (!(v[e] > 0.f)) v[e] *= a.slope_out;
        }
        if (a.gate_out) {
          float gv[8];
          unpack8_bf16(*reinterpret_cast<const uint4*>(a.gate_out + o), gv);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (!(gv[e] > 0.f)) v[e] *= a.slope_out;
        }
        if (a.residual) {
          float rv[8];
          unpack8_bf16(*reinterpret_cast<const uint4*>(a.residual + o), rv);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rv[e];
        }
        if (a.out_f32) {
          float* op = reinterpret_cast<float*>(a.out) + o;
          *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.out) + o) = pack8_bf16(v);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
#ifdef CG_CONV_TIMING
    RW_T(tq_b); tq_epi += tq_b - tq_a;
#endif
  }
#ifdef CG_CONV_TIMING
  if (a.tdbg && tid == 0) {   // per workgroup: total, wait (window + barrier), MFMA loop, epilogue cycles
    unsigned long long* d = a.tdbg + (size_t)blockIdx.x * 8;
    d[0] = __builtin_amdgcn_s_memtime() - tq_t0; d[1] = tq_wait; d[2] = tq_mma; d[3] = tq_epi;
  }
#endif
#undef RW_T
}

// -------------------------------------------------------------------------------------------
// Zero-insertion ("unpool", resnet_ops.py:35-56) + 3x3 convolution with ALL FOUR output phases in
// one workgroup.  hconv_kernel runs the phases as separate workgroups (blockIdx.y): phase (0,0) has
// ONE tap, (0,1) / (1,0) two, (1,1) four, so a phase workgroup of a 128-channel layer loops over 2-8
// K-slices and spends most of its life in set-up, first-load latency and epilogue (293-580 TFLOP/s on
// the generator's up-convolutions, profiles/r03_dstep_launches.txt).  Here a workgroup owns an 8 x 16
// LOW-RESOLUTION tile x 64 output channels and all four phases of it (16 x 32 output pixels): the
// input window (9 x 17 pixels, shared by every phase) is staged once per 64-channel block, the nine
// taps' weight slabs stream through a three-deep ring, and tap (r, s) accumulates into the phase
// (r != 1, s != 1) it belongs to, reading the window at shift (r == 2, s == 2).
//  * 4 waves, each 32 low-resolution pixels x 64 channels x 4 phases = 128 accumulator registers; two
//    workgroups per CU;
//  * window double-buffered (the next channel block lands during the nine taps of the current one),
//    weight ring waits are counted (vmcnt 2 / 8), one barrier per tap;
//  * a wave owns ALL 64 channels of its pixels, so every output pixel is written as full 128-byte
//    rows (a channel-split wave would write 64-byte halves of lines that are 2 pixels apart);
//  * FUSE == 1: batch-norm (+ReLU) prologue on the staged window and per-channel statistics of the
//    stored output, one statistics row per workgroup tile (all phases together).
// -------------------------------------------------------------------------------------------
constexpr int HU_TW = 16, HU_TH = 8, HU_PITCH = HU_TW + 2;
constexpr int HU_ROWS = (HU_TH + 1) * HU_PITCH;        // 162 window rows (pixels)
constexpr int HU_SLOTS = 6;                            // window pieces per wave: 4 x 6 = 24 pieces
constexpr int HU_HB = 4 * HU_SLOTS * 1024;             // one window buffer (the last 3 pieces are dummies)
constexpr int HU_RING = 3;
constexpr int HU_SLAB = 64 * 128;                      // 64 out-channels x 64 k, bf16

template <bool RELU, int FUSE>
__global__ __launch_bounds__(256, 2) void hup_kernel(HConvArgs a) {
  constexpr int LDS_BYTES = 2 * HU_HB + HU_RING * HU_SLAB;
  constexpr int TAB_OFF = LDS_BYTES;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES + 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, half = lane >> 5;

  const int wg = hc_xcd_remap(blockIdx.x, gridDim.x);
  const int st = (int)fdiv((uint32_t)wg, a.dNt);
  const int nt = wg - st * a.ntiles;
  const int t1 = (int)fdiv((uint32_t)st, a.dTx);
  const int tx = st - t1 * a.tiles_x;
  const int n = (int)fdiv((uint32_t)t1, a.dTy);
  const int ty = t1 - n * a.tiles_y;
  const int n0 = nt * 64;
  const int nk = 9 * a.cblocks;

  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bt =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.bt, 0, a.bt_bytes, 0x00020000);

  // ---- staging descriptors ----
  uint32_t hvoff[HU_SLOTS];
  int hc8[HU_SLOTS];
  const int iy0 = ty * HU_TH, ix0 = tx * HU_TW;
  const bool ragged_ci = (a.Ci & 63) != 0;
#pragma unroll
  for (int j = 0; j < HU_SLOTS; ++j) {
    const int row = (wave + 4 * j) * 8 + (lane >> 3);
    const int hy = row / HU_PITCH, hx = row - hy * HU_PITCH;
    const int c = (lane & 7) ^ ((hx >> 1) & 7);
    const int iy = iy0 + hy, ix = ix0 + hx;
    const bool ok = row < HU_ROWS && hx <= HU_TW && iy < a.Hin && ix < a.Win;
    hvoff[j] = ok ? (uint32_t)((((n * a.Hin + iy) * a.Win + ix) * a.Ci + c * 8) * 2) : HC_OOB;
    hc8[j] = c * 8;
  }
  uint32_t bvoff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave * 2 + j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    bvoff[j] = (n0 + row) < a.Co ? (uint32_t)(((n0 + row) * a.Kp + c * 8) * 2) : HC_OOB;
  }
  auto issue_halo = [&](int cb) {
    const int crem = a.Ci - cb * 64;
    unsigned char* hbuf = smem + (cb & 1) * HU_HB;
#pragma unroll
    for (int j = 0; j < HU_SLOTS; ++j) {
      const uint32_t vo = (ragged_ci && hc8[j] >= crem) ? HC_OOB : hvoff[j];
      hc_dma16(rs_in, vo, (uint32_t)(cb * 128), hbuf + (wave + 4 * j) * 1024);
    }
  };
  auto issue_b = [&](int s) {   // slab of K-slice s = (channel block s / 9, tap s % 9)
    const int cb = s / 9, tap = s - cb * 9;
    unsigned char* bs = smem + 2 * HU_HB + (s % HU_RING) * HU_SLAB;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      hc_dma16(rs_bt, bvoff[j], (uint32_t)((tap * a.Ci + cb * 64) * 2), bs + (wave * 2 + j) * 1024);
  };
  const bool bnp = FUSE == 1 && a.bn_mean != nullptr;
  // the block's coefficient rows by 4-byte LDS-DMAs of wave 0 (hconv_kernel)
  auto fetch_bn_table = [&](int cb) {
    if (bnp && wave == 0) {
      float* tab = reinterpret_cast<float*>(smem + TAB_OFF);
      const int ch = min(cb * 64 + lane, a.Ci - 1);
      const int64_t pidx = a.bn_per_sample ? (int64_t)n * a.Ci + ch : ch;
      const int64_t sidx = a.bn_stat_group > 0 ? (int64_t)(n / a.bn_stat_group) * a.Ci + ch : ch;
      hc_glds4(a.bn_mean + sidx, tab);
      hc_glds4(a.bn_var + sidx, tab + 64);
      if (a.bn_gamma) hc_glds4(a.bn_gamma + pidx, tab + 128);
      if (a.bn_beta) hc_glds4(a.bn_beta + pidx, tab + 192);
    }
  };
  if (bnp && tid < 64) {
    float* tab = reinterpret_cast<float*>(smem + TAB_OFF);
    if (!a.bn_gamma) tab[128 + tid] = 1.f;
    if (!a.bn_beta) tab[192 + tid] = 0.f;
  }
  auto bn_transform = [&](int cb) {   // as hconv_kernel: in place, padding pixels stay zero
    const float* tab = reinterpret_cast<const float*>(smem + TAB_OFF);
    const int crem = a.Ci - cb * 64;
    unsigned char* hbuf = smem + (cb & 1) * HU_HB;
    const int c8 = (tid & 7) * 8;   // one channel group per thread: coefficients in registers
    if (c8 >= crem) return;
    float cm[8], cr[8], cg[8], cbt[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      cm[e] = tab[c8 + e];
      cr[e] = rsqrtf(tab[64 + c8 + e] + a.bn_eps);
      cg[e] = tab[128 + c8 + e];
      cbt[e] = tab[192 + c8 + e];
    }
    constexpr int NIT = (HU_ROWS + 31) / 32, NB = 3;   // batches of three rows (hconv_kernel)
#pragma unroll
    for (int b0 = 0; b0 < NIT; b0 += NB) {
      uint4* pp[NB];
      uint4 raw[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int row = (tid >> 3) + 32 * (b0 + b);
        const int hy = row / HU_PITCH, hx = row - hy * HU_PITCH;
        const bool ok = b0 + b < NIT && row < HU_ROWS && hx <= HU_TW && iy0 + hy < a.Hin &&
                        ix0 + hx < a.Win;
        pp[b] = ok ? reinterpret_cast<uint4*>(hbuf + (row * 8 + ((tid & 7) ^ ((hx >> 1) & 7))) * 16)
                   : nullptr;
        if (pp[b]) raw[b] = *pp[b];
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (!pp[b]) continue;
        float v[8];
        unpack8_bf16(raw[b], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = (v[e] - cm[e]) * cr[e];
          t = t * cg[e] + cbt[e];
          v[e] = fmaxf(t, 0.f);
        }
        *pp[b] = pack8_bf16(v);
      }
    }
  };

  fetch_bn_table(0);   // (first: the counted waits below leave the LATEST DMAs in flight)
  issue_halo(0);
  issue_b(0);
  if (nk > 1) issue_b(1);

  // ---- fragment addressing: pixel p = wave * 32 + frow -> (y, x) of the 8 x 16 tile ----
  const int py = (wave * 32 + frow) >> 4, px = frow & 15;
  int aoff[2][2];   // [dy][dx] byte offset of the pixel's window row; chunk swizzle per dx
  int aswz[2];
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) aoff[dy][dx] = ((py + dy) * HU_PITCH + px + dx) * 128;
#pragma unroll
  for (int dx = 0; dx < 2; ++dx) aswz[dx] = ((px + dx) >> 1) & 7;
  int bko[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) bko[kk] = frow * 128 + (((kk * 2 + half) ^ ((frow >> 1) & 7)) << 4);

  f32x16_t acc[4][2];   // [phase][32-channel tile]
#pragma unroll
  for (int ph = 0; ph < 4; ++ph)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[ph][j][v] = 0.f;

  const bool have_j1 = a.Co - n0 > 32;   // workgroup-uniform
  int slot = 0;
  // one channel block (nine taps).  J1: the tile's second 32 output channels exist (Co = 96: not in
  // the second tile); NKK: k-steps with real input channels (2 in the ragged last block of
  // Ci % 64 == 32); both chosen outside the loop (no branch around the MFMAs inside it)
  auto block = [&](int cb, auto j1_c, auto nkk_c) {
    constexpr bool J1 = decltype(j1_c)::value != 0;
    constexpr int NKK = decltype(nkk_c)::value;
    const unsigned char* Hb = smem + (cb & 1) * HU_HB;
    const bool next_cb = cb + 1 < a.cblocks;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int s = cb * 9 + tap;
      // own pieces of slab s (and, on tap 0, of this block's window) have landed: what may still be
      // in flight are the later slab (2 instructions) and, on tap 1, the next window (6) issued
      // after tap 0's barrier
      if (tap == 1 && next_cb) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (s + 1 < nk) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (bnp && tap == 0) {
        __syncthreads();
        bn_transform(cb);
        __syncthreads();
      } else {
        asm volatile("s_barrier" ::: "memory");
      }
      if (tap == 0 && next_cb) {   // the other window buffer was last read a whole block ago
        // (the table is read by bn_transform(cb + 1), nine barriers from here; its loads go first:
        // the compiler waits for them before the LDS stores, and vmcnt retires in order)
        fetch_bn_table(cb + 1);
        issue_halo(cb + 1);
      }
      if (s + 2 < nk) issue_b(s + 2);   // into the slot every wave finished before this barrier

      const int tr = tap / 3, ts = tap % 3;
      const int ph = (tr != 1 ? 2 : 0) + (ts != 1 ? 1 : 0);
      const int dy = tr == 2 ? 1 : 0, dx = ts == 2 ? 1 : 0;
      const unsigned char* Bs = smem + 2 * HU_HB + slot * HU_SLAB;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(Hb + aoff[dy][dx] +
                                                         (((kk * 2 + half) ^ aswz[dx]) << 4));
        if (RELU) af = hc_relu(af);
        const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(Bs + bko[kk]);
        acc[ph][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, af, acc[ph][0], 0, 0, 0);
        if constexpr (J1) {
          const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(Bs + bko[kk] + 32 * 128);
          acc[ph][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, af, acc[ph][1], 0, 0, 0);
        }
      }
      slot = slot + 1 == HU_RING ? 0 : slot + 1;
    }
  };
  auto run_blocks = [&](auto j1_c) {
    const int nfull = ragged_ci ? a.cblocks - 1 : a.cblocks;
    int cb = 0;
    for (; cb < nfull; ++cb) block(cb, j1_c, std::integral_constant<int, 4>());
    for (; cb < a.cblocks; ++cb) block(cb, j1_c, std::integral_constant<int, 2>());
  };
  if (have_j1) run_blocks(std::integral_constant<int, 1>());
  else run_blocks(std::integral_constant<int, 0>());

  // ---- epilogue: per-wave staging (32 pixels x 64 channels fp32), 8 lanes finish one pixel's 64
  // channels: 128-byte rows ----
  constexpr int SP = 64 * 4 + 16;
  static_assert(4 * 32 * SP <= LDS_BYTES, "epilogue staging does not fit");
  unsigned char* Sw = smem + wave * (32 * SP);
  const int g8 = lane & 7, rl = lane >> 3;
  const int co = n0 + g8 * 8;
  const bool co_ok = co < a.Co;
  float bv[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = s1[e] = s2[e] = 0.f;
  if (a.bias && co_ok) {
    const float4 b0 = *reinterpret_cast<const float4*>(a.bias + co);
    const float4 b1 = *reinterpret_cast<const float4*>(a.bias + co + 4);
    bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
    bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
  }
  __syncthreads();   // window / ring images are dead
  const float osc = a.out_scale;
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(Sw + frow * SP + (j * 32 + q * 8 + 4 * half) * 4) =
            make_float4(acc[ph][j][q * 4 + 0], acc[ph][j][q * 4 + 1], acc[ph][j][q * 4 + 2],
                        acc[ph][j][q * 4 + 3]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = rl + 8 * k;
      const float4 lo = *reinterpret_cast<const float4*>(Sw + row * SP + g8 * 32);
      const float4 hi = *reinterpret_cast<const float4*>(Sw + row * SP + g8 * 32 + 16);
      if (!co_ok) continue;
      const int p = wave * 32 + row;
      const int oy = (ty * HU_TH + (p >> 4)) * 2 + (ph >> 1), ox = (tx * HU_TW + (p & 15)) * 2 + (ph & 1);
      const int64_t o = ((int64_t)(n * a.Ho + oy) * a.Wo + ox) * a.Co + co;
      float v[8] = {lo.x * osc + bv[0], lo.y * osc + bv[1], lo.z * osc + bv[2], lo.w * osc + bv[3],
                    hi.x * osc + bv[4], hi.y * osc + bv[5], hi.z * osc + bv[6], hi.w * osc + bv[7]};
      if (a.self_gate) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (!(v[e] > 0.f)) v[e] *= a.slope_out;
      }
      if (a.gate_out) {
        float gv[8];
        unpack8_bf16(*reinterpret_cast<const uint4*>(a.gate_out + o), gv);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (!(gv[e] > 0.f)) v[e] *= a.slope_out;
      }
      if (a.residual) {
        float rv[8];
        unpack8_bf16(*reinterpret_cast<const uint4*>(a.residual + o), rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      if (a.out_f32) {
        float* op = reinterpret_cast<float*>(a.out) + o;
        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        const uint4 pk = pack8_bf16(v);
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.out) + o) = pk;
        if (FUSE == 1 && a.stats) unpack8_bf16(pk, v);   // statistics of the STORED values
      }
      if (FUSE == 1 && a.stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s1[e] += v[e];
          s2[e] += v[e] * v[e];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (FUSE == 1 && a.stats) {   // wave-uniform
    // lane = rl * 8 + g8: the two lanes of a 16-lane DPP row with the same g8 by one row rotation, the
    // four rows of a wave and the four waves through LDS in a fixed order (as hconv_kernel)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1[e] += hc_row_ror(s1[e], 8);
      s2[e] += hc_row_ror(s2[e], 8);
    }
    __syncthreads();   // every wave is done with its staging rows
    float* sreg = reinterpret_cast<float*>(smem);   // [4 waves][4 DPP rows][2][64]
    if ((lane & 15) < 8) {
      const int dr = lane >> 4;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sreg[((wave * 4 + dr) * 2 + 0) * 64 + g8 * 8 + e] = s1[e];
        sreg[((wave * 4 + dr) * 2 + 1) * 64 + g8 * 8 + e] = s2[e];
      }
    }
    __syncthreads();
    if (tid < 64 && n0 + tid < a.Co) {
      float t1s = 0.f, t2s = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 16; ++w4) {
        t1s += sreg[(w4 * 2 + 0) * 64 + tid];
        t2s += sreg[(w4 * 2 + 1) * 64 + tid];
      }
      a.stats[(int64_t)st * 2 * a.Co + n0 + tid] = t1s;
      a.stats[(int64_t)st * 2 * a.Co + a.Co + n0 + tid] = t2s;
    }
  }
}

// -------------------------------------------------------------------------------------------
// Weight gradient of the same convolutions (3x3, unit stride, 'SAME'): dw[tap][ci][co] =
// sum_pixels x[pixel + tap][ci] * dy[pixel][co]  (tf.gradients of arch_ops.conv2d w.r.t. the kernel,
// arch_ops.py:559-573).  A workgroup owns a 64-channel x 64-out-channel block of ALL 9 taps and a
// range of 256-pixel tiles; per tile the input window with its halo (42.5 KiB) and the dy tile
// (32 KiB) are staged once (double-buffered, buffer_load ... lds, padding by the bounds check) and
// every tap's operand is a shifted hardware-transpose read (ds_read_b64_tr_b16) of that window.
// What bounded the round-1 kernel was VALU issue (13 VALU instructions per MFMA: per-tap address
// arithmetic; PMC in profiles/r02_pmc_wgrad.txt), so here every LDS address is a per-lane base
// register plus a compile-time immediate: the 16-byte chunk swizzle byte ^= (row & 2) << 5 only
// depends on (row mod 4), and tap shifts / k-step offsets are constants, so four base registers
// (one per residue) cover all 9 taps x 16 k-steps.
// 8 waves = 2 (channel halves) x 2 (out-channel halves) x 2 (tap groups: taps 0-4 / taps 5-8 + the
// bias-gradient column sums), 5 accumulator tiles per wave, one workgroup per CU.
// -------------------------------------------------------------------------------------------
struct HWgradArgs {
  const bf16_t* in;
  const bf16_t* dy;
  float* out;    // dw (splits == 1) or partials [splits][K*Co]
  float* bias;   // nullptr, dbias (splits == 1) or partials [splits][Co]
  int N, Hin, Win, Ci, Co, pt, pl;
  int K, ntiles;
  int tiles_x, tiles_y, nslices, slices_per_split;
  int accumulate;
  int dy_up;         // dy is [N, H/2, W/2, Co]: the gradient of a 2x2 average pooling behind the
  float out_scale;   // convolution (its nearest-neighbour up-sampling times out_scale = 1/4)
  int asm_dma;       // LDS-DMA from inline asm (cg_dma16_asm): keeps hipcc from draining the next
                     // slice's prefetch in front of the transposed LDS reads (CGAMD_ASM_DMA, A/B)
  FastDiv dNt, dTx, dTy;
};

typedef __attribute__((ext_vector_type(4))) short hc_s16x4_t;
typedef __attribute__((address_space(3))) hc_s16x4_t* hc_tr_ptr;
typedef __attribute__((address_space(3))) unsigned char* hc_lds_ptr;

__device__ __forceinline__ bf16x8_t hc_tr_read2(hc_lds_ptr p) {
  const hc_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((hc_tr_ptr)p);
  const hc_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((hc_tr_ptr)(p + 512));
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

template <bool RELU, int TWL, bool ASMDMA>
__global__ __launch_bounds__(512, 2) void hwgrad_kernel(HWgradArgs a) {
  constexpr int TW = 1 << TWL, TH = 256 >> TWL, PITCH = TW + 2;
  constexpr int HROWS = (TH + 2) * PITCH;
  constexpr int X_BYTES = HC_HALO_BYTES, Y_BYTES = 256 * 128, BUF = X_BYTES + Y_BYTES;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = (int)fdiv((uint32_t)blockIdx.x, a.dNt);
  const int nt = blockIdx.x - cb * a.ntiles;
  const int c0 = nt * 64;
  // wave roles: tap group tg, channel half wk, out-channel half wn.  Waves w and w + 4 share a SIMD
  // (cyclic placement), so the role bit that is EMPTY in this workgroup -- the second channel half of
  // a ragged last channel block (Ci = 96: 64 + 32), the second out-channel half of a ragged last
  // out-channel tile -- is taken from wave >> 2: every SIMD then carries one working and one idle
  // wave and the block's matrix work halves, instead of two SIMDs idling beside two full ones
  const bool ragged_k = cb * 64 + 32 >= a.Ci, ragged_n = c0 + 32 >= a.Co;   // workgroup-uniform
  int tg, wk, wn;
  if (ragged_k) {
    wk = wave >> 2; tg = (wave >> 1) & 1; wn = wave & 1;
  } else if (ragged_n) {
    wn = wave >> 2; tg = (wave >> 1) & 1; wk = wave & 1;
  } else {
    tg = wave >> 2; wk = (wave >> 1) & 1; wn = wave & 1;
  }
  const bool idle = (ragged_k && wk == 1) || (ragged_n && wn == 1);   // wave-uniform

  // ---- staging descriptors (relative to the tile's origin; the origin goes into the descriptor
  // base).  Halo piece p = wave + 8 j: row 8 p + (lane >> 3); 16-byte chunk c of row r lives at chunk
  // c ^ (((r >> 1) & 1) << 2), so LDS chunk (lane & 7) holds source chunk (lane & 7) ^ that
  uint32_t hrel[HC_HSLOTS];
  int hyx[HC_HSLOTS];
#pragma unroll
  for (int j = 0; j < HC_HSLOTS; ++j) {
    const int row = (wave + 8 * j) * 8 + (lane >> 3);
    const int hy = row / PITCH, hx = row - hy * PITCH;
    const int c = (lane & 7) ^ (((row >> 1) & 1) << 2);
    const bool ok = row < HROWS && (cb * 64 + c * 8) < a.Ci;
    hrel[j] = (uint32_t)(((hy * a.Win + hx) * a.Ci + c * 8) * 2);
    hyx[j] = ok ? (hy | (hx << 16)) : 0x7fff7fff;
  }
  uint32_t yrel[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = (wave * 4 + j) * 8 + (lane >> 3);
    const int y = p >> TWL, x = p & (TW - 1);
    const int c = (lane & 7) ^ (((p >> 1) & 1) << 2);
    const int us = a.dy_up;
    yrel[j] = (c0 + c * 8) < a.Co
                  ? (uint32_t)((((y >> us) * (a.Win >> us) + (x >> us)) * a.Co + c0 + c * 8) * 2)
                  : HC_OOB;
  }
  // staging of one slice, in two steps so that the inline-asm form can spread the pieces over the
  // k-steps of the slice being multiplied: prepare() = the slice's scalars, piece(i) = one 1-KiB DMA
  // (i < HC_HSLOTS: window piece wave + 8 i, then the 4 dy pieces of this wave)
  struct StageCtx {
    int iy0, ix0;
    const bf16_t* xo;
    const bf16_t* yo;
    cg_i32x4_t rxa, rya;
    uint32_t xb_addr, yb_addr;
  };
  constexpr int NPIECE = HC_HSLOTS + 4;
  auto stage_prepare = [&](StageCtx& c, int buf, int sl) {
    // slice -> (image, tile row, tile column): wave-uniform
    const int t1 = (int)fdiv((uint32_t)sl, a.dTx);
    const int tx = sl - t1 * a.tiles_x;
    const int n = (int)fdiv((uint32_t)t1, a.dTy);
    const int ty = t1 - n * a.tiles_y;
    c.iy0 = ty * TH - a.pt;
    c.ix0 = tx * TW - a.pl;
    c.xo = a.in + (((int64_t)n * a.Hin + c.iy0) * a.Win + c.ix0) * a.Ci + cb * 64;
    const int us = a.dy_up;   // tile origins are even
    c.yo = a.dy + (((int64_t)n * (a.Hin >> us) + ((ty * TH) >> us)) * (a.Win >> us) +
                   ((tx * TW) >> us)) * a.Co;
    if constexpr (ASMDMA) {
      c.rxa = cg_make_rsrc(c.xo, 0x7fffffffu);
      c.rya = cg_make_rsrc(c.yo, 0x7fffffffu);
      c.xb_addr = cg_lds_addr(smem + buf * BUF);
      c.yb_addr = c.xb_addr + X_BYTES;
    }
  };
  auto stage_piece = [&](const StageCtx& c, int i) {   // ASMDMA only; i is a compile-time constant
    if (i < HC_HSLOTS) {
      if (wave + 8 * i < HC_HALO_PIECES) {
        const int hy = hyx[i] & 0xffff, hx = hyx[i] >> 16;
        const bool ok = (unsigned)(c.iy0 + hy) < (unsigned)a.Hin &&
                        (unsigned)(c.ix0 + hx) < (unsigned)a.Win;
        cg_dma16_asm(c.rxa, ok ? hrel[i] : HC_OOB, 0u, c.xb_addr + (wave + 8 * i) * 1024);
      }
    } else {
      const int j = i - HC_HSLOTS;
      cg_dma16_asm(c.rya, yrel[j], 0u, c.yb_addr + (wave * 4 + j) * 1024);
    }
  };
  auto stage = [&](int buf, int sl) {
    StageCtx c;
    stage_prepare(c, buf, sl);
    if constexpr (ASMDMA) {
#pragma unroll
      for (int i = 0; i < NPIECE; ++i) stage_piece(c, i);
    } else {
      unsigned char* Xb = smem + buf * BUF;
      unsigned char* Yb = Xb + X_BYTES;
      const __amdgpu_buffer_rsrc_t rx =
          __builtin_amdgcn_make_buffer_rsrc((void*)c.xo, 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t ry =
          __builtin_amdgcn_make_buffer_rsrc((void*)c.yo, 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int j = 0; j < HC_HSLOTS; ++j) {
        if (wave + 8 * j < HC_HALO_PIECES) {
          const int hy = hyx[j] & 0xffff, hx = hyx[j] >> 16;
          const bool ok = (unsigned)(c.iy0 + hy) < (unsigned)a.Hin &&
                          (unsigned)(c.ix0 + hx) < (unsigned)a.Win;
          hc_dma16(rx, ok ? hrel[j] : HC_OOB, 0, Xb + (wave + 8 * j) * 1024);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) hc_dma16(ry, yrel[j], 0, Yb + (wave * 4 + j) * 1024);
    }
  };

  // ---- transpose-read addressing: for k-step ks (16 pixels) this lane supplies pixel
  // ks*16 + prow (lo read) and + 4 (hi read: +512 B, the same tile row), 4 channels at tcolb
  const int l16 = lane & 15;
  const int prow = (lane >> 5) * 8 + (l16 >> 2);
  const int tcolb = (((lane >> 4) & 1) * 16 + (l16 & 3) * 4) * 2;
  const int xcolb = wk * 64 + tcolb, ycolb = wn * 64 + tcolb;
  int xb0[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) xb0[c] = prow * 128 + (xcolb ^ (((prow + c) & 2) << 5));
  const int yb0 = X_BYTES + prow * 128 + (ycolb ^ ((prow & 2) << 5));

  f32x16_t acc[5];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
  f32x16_t accb;
#pragma unroll
  for (int v = 0; v < 16; ++v) accb[v] = 0.f;
  const bool want_bias = a.bias != nullptr && cb == 0 && wk == 0 && tg == 1;   // wave-uniform
  const s16x8_t ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);

  const int sbeg = blockIdx.y * a.slices_per_split;
  const int send = min(a.nslices, sbeg + a.slices_per_split);
  if (sbeg < send) stage(0, sbeg);

  hc_lds_ptr lds = (hc_lds_ptr)smem;
  // one 256-pixel slice; ROLE 0 / 1: tap group of a working wave, 2: idle wave (stages only).  The
  // role is chosen outside the loop: a branch around the MFMAs inside it costs accumulator copies
  auto slice = [&](int sl, auto role_c) {
    constexpr int ROLE = decltype(role_c)::value;
    const int buf = (sl - sbeg) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (RELU) {
      // ReLU of the input operand: once per staged window, in LDS, by the wave that staged the piece
      // (hconv_kernel), instead of on each of the 9 x 16 transposed fragment reads
#pragma unroll
      for (int j = 0; j < HC_HSLOTS; ++j)
        if (wave + 8 * j < HC_HALO_PIECES) {
          bf16x8_t* p = reinterpret_cast<bf16x8_t*>(smem + buf * BUF + (wave + 8 * j) * 1024 + lane * 16);
          *p = hc_relu(*p);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");   // slice landed for everybody; the other buffer is free
    // the next slice: all pieces at once (builtin DMA: hipcc waits for them before the first
    // transposed read anyway), or spread over the first NPIECE k-steps below (inline-asm DMA: the 8
    // waves no longer stall in the DMA issue queue together while the matrix pipes idle)
    const bool more = sl + 1 < send;
    StageCtx nctx;
    if constexpr (ASMDMA) {
      if (more) stage_prepare(nctx, buf ^ 1, sl + 1);
    } else {
      if (more) stage(buf ^ 1, sl + 1);
    }
    const int boff = buf * BUF;
    int xb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) xb[c] = xb0[c] + boff;
    const int yb = yb0 + boff;
    auto compute = [&](auto tgc) {
      constexpr int TG = decltype(tgc)::value;
      constexpr int NT = TG == 0 ? 5 : 4;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if constexpr (ASMDMA) {
          if (ks < NPIECE && more) stage_piece(nctx, ks);
        }
        const int kc = TWL == 5 ? (ks >> 1) * PITCH + (ks & 1) * 16 : ks * PITCH;
        const bf16x8_t yf = hc_tr_read2(lds + yb + ks * 2048);
        if (TG == 1 && want_bias)
          accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, yf, accb, 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const int tap = TG * 5 + tt;
          const int sh = kc + (tap / 3) * PITCH + (tap % 3);
          const bf16x8_t xf = hc_tr_read2(lds + xb[sh & 3] + sh * 128);
          acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, yf, acc[tt], 0, 0, 0);
        }
      }
    };
    if constexpr (ROLE == 2) {   // no operand of this wave's tiles is real: it only stages its share
      if constexpr (ASMDMA) {
        if (more) {
#pragma unroll
          for (int i = 0; i < NPIECE; ++i) stage_piece(nctx, i);
        }
      }
    } else {
      compute(role_c);
    }
  };
  if (idle) {
    for (int sl = sbeg; sl < send; ++sl) slice(sl, std::integral_constant<int, 2>());
  } else if (tg == 0) {
    for (int sl = sbeg; sl < send; ++sl) slice(sl, std::integral_constant<int, 0>());
  } else {
    for (int sl = sbeg; sl < send; ++sl) slice(sl, std::integral_constant<int, 1>());
  }

  // ---- write-out: acc[tt][v] = dw[tap][ci = (v & 3) + 8 (v >> 2) + 4 (lane >> 5)][co = lane & 31]
  const bool direct = (gridDim.y == 1);
  float* outp = a.out + (direct ? 0 : (int64_t)blockIdx.y * a.K * a.Co);
  const int co = c0 + wn * 32 + (lane & 31);
  if (co < a.Co) {
#pragma unroll
    for (int tt = 0; tt < 5; ++tt) {
      const int tap = tg * 5 + tt;
      if (tap < 9) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int ch = cb * 64 + wk * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
          if (ch >= a.Ci) continue;
          const int64_t o = ((int64_t)tap * a.Ci + ch) * a.Co + co;
          if (direct && a.accumulate)
            outp[o] += acc[tt][v] * a.out_scale;
          else
            outp[o] = acc[tt][v] * a.out_scale;
        }
      }
    }
    // every row of accb holds the column sums; row 0 lives in accb[0] of lanes 0..31
    if (want_bias && lane < 32) {
      float* bp = a.bias + (direct ? 0 : (int64_t)blockIdx.y * a.Co) + co;
      *bp = (direct && a.accumulate) ? *bp + accb[0] * a.out_scale : accb[0] * a.out_scale;
    }
  }
}


// -------------------------------------------------------------------------------------------
// Image-input ("stem") 3x3 convolutions, Ci <= 3 (the first convolutions of every discriminator:
// resnet5.py:118-121, resnet_cifar.py:136-139, resnet_biggan.py:372-377).  ~1 % of the FLOPs but HBM-
// bound on the activation they write (forward) or read (weight gradient): 268 MB at 128x128x64 and
// batch 128.  The round-1 kernels gathered the im2col rows from global memory element by element
// (1.9 TB/s forward, 1.2 TB/s weight gradient); here the input window of a 256-pixel tile is staged
// in LDS once (2 KiB), the K = 9 Ci <= 27 im2col elements of a fragment are 2-byte LDS reads at
// per-lane constant offsets, the weights live in registers for the whole workgroup, and a
// workgroup walks several tiles.
// -------------------------------------------------------------------------------------------
struct WStemArgs {
  const bf16_t* in;
  const bf16_t* bt;    // forward: [Co][32] bf16 (k = (r*3 + s)*Ci + c, zero-padded)
  const bf16_t* dy;    // weight gradient: [N,H,W,Co] bf16
  void* out;           // forward: [N,H,W,Co];  weight gradient: partials [splits][K*Co + Co] fp32
  const float* bias;
  int N, H, W, Ci, Co;
  int tiles_x, tiles_y, ntiles, tiles_per_wg;
  int relu_in, out_f32, self_gate, want_bias;
  float slope_out;
  int pool;    // forward: 2x2 average pooling of the output in the epilogue (Co = 64 / 128)
  int dy_up;   // weight gradient: dy is the pooled-resolution gradient (see HWgradArgs)
  FastDiv dTx, dTy;
};

constexpr int WS_WIN_MAX = 1024;   // window elements: 10 x 34 x 3 = 1020, 18 x 18 x 3 = 972
constexpr int WS_CI = 3;           // RGB inputs (K = 27, padded to 32)

// window of tile (n, ty, tx) -> LDS (u16 [TH + 2][(TW + 2) Ci]), zero outside the image
template <int TWL>
__device__ __forceinline__ void ws_load_window(const WStemArgs& a, int n, int ty, int tx,
                                               bf16_t* win, int tid, int nthreads) {
  constexpr int TW = 1 << TWL, TH = 256 >> TWL;
  constexpr int WC = (TW + 2) * WS_CI, WE = (TH + 2) * WC;
  const int rowlen = a.W * WS_CI;
  for (int e = tid; e < WE; e += nthreads) {
    const int wy = e / WC, wc = e - wy * WC;
    const int iy = ty * TH - 1 + wy, ic = (tx * TW - 1) * WS_CI + wc;
    bf16_t v = 0;
    if ((unsigned)iy < (unsigned)a.H && (unsigned)ic < (unsigned)rowlen) {
      v = a.in[((int64_t)n * a.H + iy) * rowlen + ic];
      if (a.relu_in && (v & 0x8000)) v = 0;
    }
    win[e] = v;
  }
}

// the same window through registers: issue() starts branch-free loads from clamped addresses (a
// guarded load compiles to a branch with a wait on the value right behind it -- serial round trips
// that also drain whatever DMA or stores the wave has in flight), commit() writes them to LDS later
template <int TWL>
struct WsWindowRegs {
  static constexpr int TW = 1 << TWL, TH = 256 >> TWL;
  static constexpr int WC = (TW + 2) * WS_CI, WE = (TH + 2) * WC, WL = (WE + 255) / 256;
  bf16_t v[WL];
  uint32_t ok;
  __device__ __forceinline__ void issue(const WStemArgs& a, int n, int ty, int tx, int tid) {
    const int rowlen = a.W * WS_CI;
    ok = 0;
#pragma unroll
    for (int k = 0; k < WL; ++k) {
      const int e = tid + 256 * k;
      const int wy = e / WC, wc = e - wy * WC;
      const int iy = ty * TH - 1 + wy, ic = (tx * TW - 1) * WS_CI + wc;
      const bool in = e < WE && (unsigned)iy < (unsigned)a.H && (unsigned)ic < (unsigned)rowlen;
      ok |= in ? (1u << k) : 0u;
      v[k] = a.in[in ? ((int64_t)n * a.H + iy) * rowlen + ic : 0];
    }
  }
  __device__ __forceinline__ void commit(const WStemArgs& a, bf16_t* win, int tid) const {
#pragma unroll
    for (int k = 0; k < WL; ++k) {
      bf16_t x = (ok >> k) & 1u ? v[k] : (bf16_t)0;
      if (a.relu_in && (x & 0x8000)) x = 0;
      if (tid + 256 * k < WE) win[tid + 256 * k] = x;
    }
  }
};

// forward: 4 waves, each 64 pixels x all CT*32 output channels of a 256-pixel tile
template <int CT, int TWL, bool POOL>
__global__ __launch_bounds__(256) void wstem_fwd_kernel(WStemArgs a) {
  constexpr int TW = 1 << TWL, TH = 256 >> TWL;
  constexpr int CO = CT * 32;
  constexpr int SP = CO * 4 + 16;   // epilogue staging row pitch (bytes)
  constexpr int G8 = CO / 8;        // 8-channel groups per pixel
  __shared__ __attribute__((aligned(16))) bf16_t win2[2][WS_WIN_MAX + 8];
  __shared__ __attribute__((aligned(16))) unsigned char stg[4 * 32 * SP];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, half = lane >> 5;
  constexpr int WC = (TW + 2) * WS_CI, K = 9 * WS_CI;

  // weights: B fragments for every channel tile and both k-steps stay in registers
  bf16x8_t bfr[CT][2];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      bfr[ct][kk] = *reinterpret_cast<const bf16x8_t*>(a.bt + (ct * 32 + frow) * 32 + kk * 16 + half * 8);
  // this lane's 16 im2col offsets (elements, relative to the pixel's window origin); k >= K reads
  // the pixel's own tap-(0,0) element: its weight is zero and it belongs to the true window
  int koff[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int k = (j >> 3) * 16 + half * 8 + (j & 7);
    const int r = k / (3 * WS_CI), rem = k - r * 3 * WS_CI;
    koff[j] = k < K ? r * WC + rem : 0;
  }
  float bv[8];
  const int g8 = lane % G8;
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = a.bias ? a.bias[g8 * 8 + e] : 0.f;

  // The window of the NEXT tile is loaded into registers right after the barrier that publishes
  // the current one -- before this tile's output stores are issued.  vmcnt retires in order, so a
  // window load issued behind the stores (round 2: load, barrier, compute, store, loop) made every
  // tile wait for the previous tile's 32 KiB of stores to reach memory: 15 us per tile and
  // workgroup, 2.1 TB/s of a 6.9 TB/s write path.  Two window buffers: one barrier per tile.
  constexpr int WE = (TH + 2) * WC, WL = (WE + 255) / 256;
  const int rowlen = a.W * WS_CI;
  bf16_t wreg[WL];
  uint32_t wok = 0;   // bit k: element k of this thread lies inside the image
  auto window_load = [&](int t) {
    const int q = (int)fdiv((uint32_t)t, a.dTx);
    const int tx = t - q * a.tiles_x;
    const int n = (int)fdiv((uint32_t)q, a.dTy);
    const int ty = q - n * a.tiles_y;
    wok = 0;
    // unconditional loads from clamped addresses (a guarded load becomes a branch with a wait on
    // the value right behind it: four serial round trips, each draining the previous tile's stores)
#pragma unroll
    for (int k = 0; k < WL; ++k) {
      const int e = tid + 256 * k;
      const int wy = e / WC, wc = e - wy * WC;
      const int iy = ty * TH - 1 + wy, ic = (tx * TW - 1) * WS_CI + wc;
      const bool ok = e < WE && (unsigned)iy < (unsigned)a.H && (unsigned)ic < (unsigned)rowlen;
      wok |= ok ? (1u << k) : 0u;
      wreg[k] = a.in[ok ? ((int64_t)n * a.H + iy) * rowlen + ic : 0];
    }
  };
  const int t0 = blockIdx.x * a.tiles_per_wg;
  const int t1e = min(a.ntiles, t0 + a.tiles_per_wg);
  if (t0 < t1e) window_load(t0);
  for (int t = t0; t < t1e; ++t) {
    const int q = (int)fdiv((uint32_t)t, a.dTx);
    const int tx = t - q * a.tiles_x;
    const int n = (int)fdiv((uint32_t)q, a.dTy);
    const int ty = q - n * a.tiles_y;
    bf16_t* win = win2[(t - t0) & 1];
#pragma unroll
    for (int k = 0; k < WL; ++k) {
      bf16_t v = (wok >> k) & 1u ? wreg[k] : (bf16_t)0;
      if (a.relu_in && (v & 0x8000)) v = 0;
      if (tid + 256 * k < WE) win[tid + 256 * k] = v;
    }
    __syncthreads();   // window published; every wave is past its reads of the OTHER buffer's last use
    if (t + 1 < t1e) window_load(t + 1);

    f32x16_t acc[2][CT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][ct][v] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = wave * 64 + i * 32 + frow;
      const int y = p >> TWL, x = p & (TW - 1);
      const bf16_t* wp = win + y * WC + x * WS_CI;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        union { bf16x8_t f; bf16_t h[8]; } af;
#pragma unroll
        for (int e = 0; e < 8; ++e) af.h[e] = wp[koff[kk * 8 + e]];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          acc[i][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ct][kk], af.f, acc[i][ct], 0, 0, 0);
      }
    }
    // epilogue: per-wave staging (see hconv_kernel), 32 pixels per pass
    unsigned char* Sw = stg + wave * (32 * SP);
    if constexpr (POOL) {
      // pooled form (as hconv_kernel's): the wave's 64 pixels are two tile rows (8x32 tiles: rows
      // wave*2 + i -> vertical neighbours are acc[0] / acc[1] of the SAME lane, summed in registers,
      // one staging pass) or four half rows (16x16 tiles: staged rows r and r + 16 of a pass are
      // vertical neighbours); every lane then finishes pooled pixels from 2 (4) staged rows
      constexpr int NPASS = TWL == 5 ? 1 : 2;
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            float4 t4 = make_float4(acc[i][ct][qq * 4 + 0], acc[i][ct][qq * 4 + 1],
                                    acc[i][ct][qq * 4 + 2], acc[i][ct][qq * 4 + 3]);
            if (TWL == 5) {
              t4.x += acc[1][ct][qq * 4 + 0]; t4.y += acc[1][ct][qq * 4 + 1];
              t4.z += acc[1][ct][qq * 4 + 2]; t4.w += acc[1][ct][qq * 4 + 3];
            }
            *reinterpret_cast<float4*>(Sw + frow * SP + (ct * 32 + qq * 8 + 4 * half) * 4) = t4;
          }
        __builtin_amdgcn_wave_barrier();
        constexpr int NPX = TWL == 5 ? 16 : 8;   // pooled pixels per pass
#pragma unroll
        for (int it = lane; it < NPX * G8; it += 64) {
          const int x2 = it / G8, gg = it - x2 * G8;
          const unsigned char* r0 = Sw + (2 * x2) * SP + gg * 32;
          float v[8];
          {
            const float4 a0 = *reinterpret_cast<const float4*>(r0);
            const float4 a1 = *reinterpret_cast<const float4*>(r0 + 16);
            const float4 b0 = *reinterpret_cast<const float4*>(r0 + SP);
            const float4 b1 = *reinterpret_cast<const float4*>(r0 + SP + 16);
            v[0] = a0.x + b0.x; v[1] = a0.y + b0.y; v[2] = a0.z + b0.z; v[3] = a0.w + b0.w;
            v[4] = a1.x + b1.x; v[5] = a1.y + b1.y; v[6] = a1.z + b1.z; v[7] = a1.w + b1.w;
          }
          if (TWL == 4) {
            const float4 a0 = *reinterpret_cast<const float4*>(r0 + 16 * SP);
            const float4 a1 = *reinterpret_cast<const float4*>(r0 + 16 * SP + 16);
            const float4 b0 = *reinterpret_cast<const float4*>(r0 + 17 * SP);
            const float4 b1 = *reinterpret_cast<const float4*>(r0 + 17 * SP + 16);
            v[0] += a0.x + b0.x; v[1] += a0.y + b0.y; v[2] += a0.z + b0.z; v[3] += a0.w + b0.w;
            v[4] += a1.x + b1.x; v[5] += a1.y + b1.y; v[6] += a1.z + b1.z; v[7] += a1.w + b1.w;
          }
          const int y2 = TWL == 4 ? wave * 2 + i : wave;
          const int64_t o = (((int64_t)n * (a.H >> 1) + ((ty * TH) >> 1) + y2) * (a.W >> 1) +
                             ((tx * TW) >> 1) + x2) * a.Co + gg * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = 0.25f * v[e] + (a.bias ? a.bias[gg * 8 + e] : 0.f);
          if (a.out_f32) {
            float* op = reinterpret_cast<float*>(a.out) + o;
            *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
          } else {
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.out) + o) = pack8_bf16(v);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      continue;   // next tile
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          *reinterpret_cast<float4*>(Sw + frow * SP + (ct * 32 + qq * 8 + 4 * half) * 4) =
              make_float4(acc[i][ct][qq * 4 + 0], acc[i][ct][qq * 4 + 1], acc[i][ct][qq * 4 + 2],
                          acc[i][ct][qq * 4 + 3]);
      __builtin_amdgcn_wave_barrier();
      for (int it = lane; it < 32 * G8; it += 64) {
        const int row = it / G8;   // it % G8 == g8 when 64 % G8 == 0; general otherwise
        const int gg = it - row * G8;
        const float4 lo = *reinterpret_cast<const float4*>(Sw + row * SP + gg * 32);
        const float4 hi = *reinterpret_cast<const float4*>(Sw + row * SP + gg * 32 + 16);
        const int p = wave * 64 + i * 32 + row;
        const int y = p >> TWL, x = p & (TW - 1);
        const int64_t o = (((int64_t)n * a.H + ty * TH + y) * a.W + tx * TW + x) * a.Co + gg * 8;
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (gg == g8) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bv[e];
        } else if (a.bias) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += a.bias[gg * 8 + e];
        }
        if (a.self_gate) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (!(v[e] > 0.f)) v[e] *= a.slope_out;
        }
        if (a.out_f32) {
          float* op = reinterpret_cast<float*>(a.out) + o;
          *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.out) + o) = pack8_bf16(v);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// weight gradient: block = (64 output channels, split); per 256-pixel tile the window (LDS, as
// above) and the dy tile (DMA, hwgrad's transpose-read image) are staged, 4 waves split the 16
// k-steps; dw^T fragments are 8 two-byte LDS reads at constant offsets.  Partials per split
// [K*Co + Co] (dw then dbias), reduced by the caller.
template <int TWL, bool ASMDMA>
__global__ __launch_bounds__(256) void wstem_wgrad_kernel(WStemArgs a) {
  constexpr int TW = 1 << TWL, TH = 256 >> TWL, CI = WS_CI;
  constexpr int WC = (TW + 2) * CI, K = 9 * CI;
  constexpr int Y_BYTES = 256 * 128;
  __shared__ __attribute__((aligned(1024))) unsigned char ysm[2 * Y_BYTES];
  __shared__ __attribute__((aligned(16))) bf16_t win[2][WS_WIN_MAX + 8];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c0 = blockIdx.x * 64;
  const int K_Co = K * a.Co;

  // dy staging: 32 pieces of 8 pixels, 8 per wave (see hwgrad_kernel)
  uint32_t yrel[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int p = (wave * 8 + j) * 8 + (lane >> 3);
    const int y = p >> TWL, x = p & (TW - 1);
    const int c = (lane & 7) ^ (((p >> 1) & 1) << 2);
    const int us = a.dy_up;
    yrel[j] = (c0 + c * 8) < a.Co
                  ? (uint32_t)((((y >> us) * (a.W >> us) + (x >> us)) * a.Co + c0 + c * 8) * 2)
                  : HC_OOB;
  }
  auto stage = [&](int buf, int t) {
    const int q = (int)fdiv((uint32_t)t, a.dTx);
    const int tx = t - q * a.tiles_x;
    const int n = (int)fdiv((uint32_t)q, a.dTy);
    const int ty = q - n * a.tiles_y;
    const int us = a.dy_up;
    const bf16_t* yo = a.dy + (((int64_t)n * (a.H >> us) + ((ty * TH) >> us)) * (a.W >> us) +
                               ((tx * TW) >> us)) * a.Co;
    if constexpr (ASMDMA) {
      const cg_i32x4_t rya = cg_make_rsrc(yo, 0x7fffffffu);
      const uint32_t yaddr = cg_lds_addr(ysm + buf * Y_BYTES);
#pragma unroll
      for (int j = 0; j < 8; ++j) cg_dma16_asm(rya, yrel[j], 0u, yaddr + (wave * 8 + j) * 1024);
    } else {
      const __amdgpu_buffer_rsrc_t ry =
          __builtin_amdgcn_make_buffer_rsrc((void*)yo, 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        hc_dma16(ry, yrel[j], 0, ysm + buf * Y_BYTES + (wave * 8 + j) * 1024);
    }
  };
  // the window goes through registers: its loads leave BEFORE the dy DMA of the same tile and are
  // written to LDS after the current tile's MFMA work (loading it straight to LDS behind the DMA made
  // every tile wait for its successor's 32 KiB of dy before multiplying: no overlap at all)
  WsWindowRegs<TWL> wreg;
  auto window_issue = [&](int t) {
    const int q = (int)fdiv((uint32_t)t, a.dTx);
    const int tx = t - q * a.tiles_x;
    const int n = (int)fdiv((uint32_t)q, a.dTy);
    const int ty = q - n * a.tiles_y;
    wreg.issue(a, n, ty, tx, tid);
  };

  // A^T fragment of k-step ks: lane -> im2col element k = lane & 31, pixels ks*16 + half*8 + e
  const int kq = lane & 31, half = lane >> 5;
  const int kr = kq / (3 * CI), krem = kq - kr * 3 * CI;
  const int abase = (kq < K ? kr * WC + krem : 0) + half * 8 * CI;
  // dy fragments (transpose reads, hwgrad_kernel's addressing)
  const int l16 = lane & 15;
  const int prow = half * 8 + (l16 >> 2);
  const int tcolb = (((lane >> 4) & 1) * 16 + (l16 & 3) * 4) * 2;
  int yb0[2];   // the 16-byte chunk swizzle flips bit 6 of the byte column, i.e. the channel tile
#pragma unroll
  for (int j = 0; j < 2; ++j) yb0[j] = prow * 128 + ((j * 64 + tcolb) ^ ((prow & 2) << 5));

  f32x16_t acc[2], accb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      acc[j][v] = 0.f;
      accb[j][v] = 0.f;
    }
  const s16x8_t ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);

  const int t0 = blockIdx.y * a.tiles_per_wg;
  const int t1e = min(a.ntiles, t0 + a.tiles_per_wg);
  if (t0 < t1e) {
    window_issue(t0);
    stage(0, t0);
    wreg.commit(a, win[0], tid);
  }
  hc_lds_ptr ylds = (hc_lds_ptr)ysm;
  for (int t = t0; t < t1e; ++t) {
    const int buf = (t - t0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < t1e) {
      window_issue(t + 1);
      stage(buf ^ 1, t + 1);
    }
    const bf16_t* wb = win[buf] + abase;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int ks = wave * 4 + kk;
      const int wo = TWL == 5 ? (ks >> 1) * WC + (ks & 1) * 16 * CI : ks * WC;
      union { bf16x8_t f; bf16_t h[8]; } af;
#pragma unroll
      for (int e = 0; e < 8; ++e) af.h[e] = wb[wo + e * CI];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8_t yf = hc_tr_read2(ylds + buf * Y_BYTES + yb0[j] + ks * 2048);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.f, yf, acc[j], 0, 0, 0);
        if (a.want_bias) accb[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, yf, accb[j], 0, 0, 0);
      }
    }
    if (t + 1 < t1e) wreg.commit(a, win[buf ^ 1], tid);   // last read a whole tile ago
  }
  // ---- cross-wave sum through LDS, then one partial per (split): dw[k][co] rows k < K, dbias
  __syncthreads();
  float* red = reinterpret_cast<float*>(ysm);   // [4 waves][64 lanes][33] floats
  {
    float* mine = red + (wave * 64 + lane) * 33;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) mine[j * 16 + v] = acc[j][v];
    mine[32] = 0.f;
  }
  float* redb = red + 4 * 64 * 33;              // [4 waves][2][32] column sums (row 0 of accb)
  if (lane < 32) {
    redb[(wave * 2 + 0) * 32 + lane] = accb[0][0];
    redb[(wave * 2 + 1) * 32 + lane] = accb[1][0];
  }
  __syncthreads();
  float* outp = reinterpret_cast<float*>(a.out) + (int64_t)blockIdx.y * (K_Co + a.Co);
  if (wave == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int co = c0 + j * 32 + (lane & 31);
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int k = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[(w * 64 + lane) * 33 + j * 16 + v];
        if (k < K && co < a.Co) outp[(int64_t)k * a.Co + co] = a.dy_up ? 0.25f * s : s;
      }
    }
  } else if (wave == 1 && a.want_bias) {
    const int j = lane >> 5, co = c0 + j * 32 + (lane & 31);
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += redb[(w * 2 + j) * 32 + (lane & 31)];
    if (co < a.Co) outp[K_Co + co] = a.dy_up ? 0.25f * s : s;
  }
}

int hc_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// tile width log2 for a per-phase output grid of Hp x Wp, or 0 if the grid does not tile
int hc_tile_log(int Hp, int Wp) {
  if (Wp >= 32 && (Wp % 32) == 0 && (Hp % 8) == 0) return 5;
  if (Wp == 16 && (Hp % 16) == 0) return 4;
  return 0;
}

}  // namespace

#ifdef CG_CONV_TIMING
static unsigned long long* g_hconv_tdbg = nullptr;
extern "C" void cg_debug_set_hconv_timing_buffer(void* p) { g_hconv_tdbg = (unsigned long long*)p; }
#endif

// out-channel tile: 128 (two 32-channel MFMA tiles per wave) unless that leaves at most
// CGAMD_HCONV_BN64_MAX workgroups (half a chip of the 128 x 16x16 x 128 layers): 64-channel tiles
// then double the grid, which buys more than the wider tile saves on a K loop of 18 slices
static int hc_pick_bn(const cgConvGeom* g) {
  static const int bn64_max = hc_env("CGAMD_HCONV_BN64_MAX", 160);
  if (g->Co <= 64) return 64;
  const int Hp = g->Ho / g->U, Wp = g->Wo / g->U;
  const int64_t wgs128 = (int64_t)g->N * (Hp * Wp / 256) * cdiv(g->Co, 128) * g->U * g->U;
  return wgs128 <= bn64_max ? 64 : 128;
}

// all-phase zero-insertion kernel (hup_kernel): 3x3 on a x2 zero-inserted input, 8 x 16 tiles
// Up to CGAMD_HUP_MAX_CO (128) output channels: every 64-channel tile re-stages (and, fused,
// re-normalises) the window, which costs more than the per-phase kernel's set-up from 4 tiles on
// (measured, batch 64: 64^2 -> 128^2 x 64: 121 / 145 -> 71 / 86 us plain / fused; 32^2 -> 64^2 x 128:
// 55 / 77 -> 51 / 68; 16^2 -> 32^2 x 256: 40 / 45 -> 40 / 62)
static bool hup_geom_ok(const cgConvGeom* g) {
  static const int enabled = hc_env("CGAMD_HUP", 1);
  static const int max_co = hc_env("CGAMD_HUP_MAX_CO", 128);
  return enabled && (enabled > 1 || g->Co <= max_co) && g->U == 2 && g->S == 1 && g->kh == 3 && g->kw == 3 && g->pt == 1 && g->pl == 1 &&
         g->Ho == 2 * g->Hin && g->Wo == 2 * g->Win && (g->Hin % HU_TH) == 0 &&
         (g->Win % HU_TW) == 0 && (g->Ci % 32) == 0 && (g->Co % 8) == 0 && g->Co >= 64;
}
static bool hconv_rw_geom_ok(const cgConvGeom* g);
static void hconv_rw_launch_ex(const cgConvGeom* g, const void* in, const void* bt, void* out,
                               int out_is_f32, const float* bias, const void* gate_in,
                               const void* gate_out, float slope_out, const void* residual,
                               int pool, int in_up, float out_scale, hipStream_t st);

static bool hc_geom_ok(const cgConvGeom* g, bool narrow) {
  if (g->S != 1 || (g->U != 1 && g->U != 2)) return false;
  if (g->kh > 3 || g->kw > 3) return false;
  if (g->Ci % 32 != 0) return false;
  if (narrow) {   // RGB outputs: plain 3x3 only (hconv_kernel<64, *, *, 3>)
    if (g->Co >= 8 || g->U != 1 || g->kh != 3 || g->kw != 3) return false;
  } else if (g->Co % 8 != 0 || g->Co < 64) {
    return false;
  }
  if (g->Ho % g->U || g->Wo % g->U) return false;
  const int Hp = g->Ho / g->U, Wp = g->Wo / g->U;
  if (Hp != g->Hin || Wp != g->Win) return false;   // 'SAME' unit-stride geometry only
  if (!hc_tile_log(Hp, Wp)) return false;
  if ((int64_t)g->N * g->Hin * g->Win * g->Ci * 2 >= (1ll << 31)) return false;
  if ((int64_t)g->Co * (((int64_t)g->kh * g->kw * g->Ci + 7) & ~7ll) * 2 >= (1ll << 31)) return false;
  return true;
}

bool cg_hconv_geom_ok(const cgConvGeom* g) { return hc_geom_ok(g, false); }

// hconv_kernel<64, *, 3, 0>: four whole 8x8 images per tile.  Only where the one-tap kernel is the
// alternative (sconv takes the small grids) and the K loop is long enough to pay for a 50 KiB window
static bool hc_mi_use(const cgConvGeom* g) {
  static const int enabled = hc_env("CGAMD_HCONV_MI", 1);
  static const int min_wgs = hc_env("CGAMD_HCONV_MI_MIN", 192);
  static const int min_ci = hc_env("CGAMD_HCONV_MI_CI", 256);
  // measured against the one-tap kernel's 64-row tiles only (ResNet5: 256 / 512 channels, 256
  // workgroups: profiles/r04_hconv_8x8_ab.txt); BigGAN's 8x8 x 1536 layers run 128x128 one-tap tiles
  // at ~0.89 PFLOP/s and stay there until that comparison exists
  static const int max_ci = hc_env("CGAMD_HCONV_MI_CI_MAX", 1024);
  if (!enabled || g->Ci > max_ci) return false;
  if (g->S != 1 || g->U != 1 || g->kh != 3 || g->kw != 3 || g->pt != 1 || g->pl != 1) return false;
  if (g->Hin != 8 || g->Win != 8 || g->Ho != 8 || g->Wo != 8) return false;
  if ((g->Ci % 32) != 0 || (g->Co % 8) != 0 || g->Co < 64 || g->Ci < min_ci) return false;
  if ((int64_t)g->N * 64 * g->Ci * 2 >= (1ll << 31)) return false;
  if ((int64_t)g->Co * 9 * g->Ci * 2 >= (1ll << 31)) return false;
  return (int64_t)cdiv(g->N, 4) * cdiv(g->Co, 64) >= min_wgs;
}

bool cg_hconv_narrow(const cgConvGeom* g) { return g->Co < 8; }
bool cg_hconv_narrow_ok(const cgConvGeom* g) { return hc_geom_ok(g, true); }

bool cg_hconv_supported(const cgConvGeom* g, const void* in, const void* gate_in, float slope_in) {
  static const int enabled = hc_env("CGAMD_HCONV", 1);
  static const int narrow_on = hc_env("CGAMD_HCONV_NARROW", 1);
  static const int min_wgs = hc_env("CGAMD_HCONV_MIN", 100);
  const bool narrow = g->Co < 8;
  if (enabled && hc_mi_use(g)) return !(gate_in && !(gate_in == in && slope_in == 0.f));
  if (!enabled || (narrow && !narrow_on) || !hc_geom_ok(g, narrow)) return false;
  if (gate_in && !(gate_in == in && slope_in == 0.f)) return false;
  const int Hp = g->Ho / g->U, Wp = g->Wo / g->U;
  const int bn = hc_pick_bn(g);
  const int64_t wgs = (int64_t)g->N * (Hp * Wp / 256) * cdiv(g->Co, bn) * g->U * g->U;
  return wgs >= min_wgs;
}

int cg_hconv_stats_phases(const cgConvGeom* g) { return hup_geom_ok(g) ? 1 : g->U * g->U; }

int cg_hconv_stats_rows(const cgConvGeom* g) {
  if (hup_geom_ok(g)) return g->N * (g->Hin / HU_TH) * (g->Win / HU_TW);   // one row per tile
  const int Hp = g->Ho / g->U, Wp = g->Wo / g->U;
  return g->U * g->U * g->N * (Hp * Wp / 256);
}

void cg_hconv_launch(const cgConvGeom* g, const void* in, const void* bt, void* out,
                     int out_is_f32, const float* bias, const void* gate_in, const void* gate_out,
                     float slope_out, const void* residual, hipStream_t st) {
  cg_hconv_launch_fused(g, in, bt, out, out_is_f32, bias, gate_in, gate_out, slope_out, residual,
                        nullptr, st);
}

void cg_hconv_launch_fused(const cgConvGeom* g, const void* in, const void* bt, void* out,
                           int out_is_f32, const float* bias, const void* gate_in,
                           const void* gate_out, float slope_out, const void* residual,
                           const cgConvFusion* fu, hipStream_t st) {
  // 64 -> 64 channels with a pooled epilogue / a pooled-resolution input and no batch-norm fusion:
  // the register-resident-weight kernel (B0 conv2 of the ResNet5 discriminator and its data gradient)
  static const int rw_fused = hc_env("CGAMD_HCONV_RW_FUSED", 1);
  static const int rw_min_tiles = hc_env("CGAMD_HCONV_RW_MIN", 512);
  if (rw_fused && fu && (fu->pool_out || fu->in_up) && !(fu->pool_out && fu->in_up) &&
      !fu->bn_mean && !fu->stats_out && hconv_rw_geom_ok(g) &&
      (int64_t)g->N * (g->Ho / 4) * (g->Wo / 32) >= rw_min_tiles &&
      !(fu->pool_out && gate_out)) {
    hconv_rw_launch_ex(g, in, bt, out, out_is_f32, bias, gate_in, gate_out, slope_out, residual,
                       fu->pool_out ? 1 : 0, fu->in_up ? 1 : 0,
                       fu->out_scale != 0.f ? fu->out_scale : 1.f, st);
    return;
  }
  const bool mi = hc_mi_use(g);   // (8x8 maps: no fused form reaches this launcher)
  const bool hup = !mi && hup_geom_ok(g) && !(fu && (fu->pool_out || fu->in_up));
  HConvArgs a;
  a.bn_mean = fu ? fu->bn_mean : nullptr;
  a.bn_var = fu ? fu->bn_var : nullptr;
  a.bn_gamma = fu ? fu->bn_gamma : nullptr;
  a.bn_beta = fu ? fu->bn_beta : nullptr;
  a.bn_eps = fu ? fu->bn_eps : 0.f;
  a.bn_per_sample = fu ? fu->bn_per_sample : 0;
  a.bn_stat_group = fu ? fu->bn_stat_group : 0;
  a.stats = fu ? fu->stats_out : nullptr;
  a.pool = fu ? fu->pool_out : 0;
  a.in_up = fu ? fu->in_up : 0;
  a.out_scale = (fu && fu->out_scale != 0.f) ? fu->out_scale : 1.f;
  a.in = (const bf16_t*)in;
  a.bt = (const bf16_t*)bt;
  a.out = out;
  a.bias = bias;
  a.self_gate = (gate_out != nullptr && gate_out == out);
  a.gate_out = a.self_gate ? nullptr : (const bf16_t*)gate_out;
  a.residual = (const bf16_t*)residual;
  a.N = g->N; a.Hin = g->Hin; a.Win = g->Win; a.Ci = g->Ci;
  a.Ho = g->Ho; a.Wo = g->Wo; a.Co = g->Co; a.kh = g->kh; a.kw = g->kw;
  a.U = g->U; a.pt = g->pt; a.pl = g->pl;
  a.Kp = (g->kh * g->kw * g->Ci + 7) & ~7;
  a.cblocks = cdiv(g->Ci, 64);
  a.in_bytes = (uint32_t)(((int64_t)g->N * g->Hin * g->Win * g->Ci * 2) >> (a.in_up ? 2 : 0));
  a.bt_bytes = (uint32_t)((int64_t)g->Co * a.Kp * 2);
  const int Hp = g->Ho / g->U, Wp = g->Wo / g->U;
  const int twl = hc_tile_log(Hp, Wp);
  const int TW = 1 << twl, TH = 256 >> twl;
  a.tiles_x = mi ? 1 : Wp / TW;
  a.tiles_y = mi ? 1 : Hp / TH;
  const int bn = mi ? 64 : hc_pick_bn(g);
  a.ntiles = cdiv(g->Co, bn);
  a.out_f32 = out_is_f32;
  a.slope_out = slope_out;
  a.dNt = make_fastdiv(a.ntiles);
  a.dTx = make_fastdiv(a.tiles_x);
  a.dTy = make_fastdiv(a.tiles_y);
#ifdef CG_CONV_TIMING
  a.tdbg = g_hconv_tdbg;
#endif
  const bool relu = gate_in != nullptr && a.bn_mean == nullptr;   // the BN prologue includes the ReLU
  if (mi) {
    dim3 mgrid(cdiv(g->N, 4) * a.ntiles, 1);
    CgProfScope prof(CG_PROF_HCONV_64, g, st);
    if (relu) hconv_kernel<64, true, 3, 0><<<mgrid, 512, 0, st>>>(a);
    else hconv_kernel<64, false, 3, 0><<<mgrid, 512, 0, st>>>(a);
    return;
  }
  const bool narrow = g->Co < 8;
  // (narrow outputs: cg_gconv only comes here without gate tensor / residual; no fused form)
  if (hup) {
    a.tiles_x = g->Win / HU_TW;
    a.tiles_y = g->Hin / HU_TH;
    a.ntiles = cdiv(g->Co, 64);
    a.dNt = make_fastdiv(a.ntiles);
    a.dTx = make_fastdiv(a.tiles_x);
    a.dTy = make_fastdiv(a.tiles_y);
    dim3 ugrid(g->N * a.tiles_y * a.tiles_x * a.ntiles);
    CgProfScope prof(CG_PROF_HCONV_64, g, st);
    if (a.bn_mean || a.stats) {
      if (relu) hup_kernel<true, 1><<<ugrid, 256, 0, st>>>(a);
      else hup_kernel<false, 1><<<ugrid, 256, 0, st>>>(a);
    } else {
      if (relu) hup_kernel<true, 0><<<ugrid, 256, 0, st>>>(a);
      else hup_kernel<false, 0><<<ugrid, 256, 0, st>>>(a);
    }
    return;
  }
  dim3 grid(g->N * a.tiles_y * a.tiles_x * a.ntiles, g->U * g->U);
  CgProfScope prof(bn == 128 ? CG_PROF_HCONV_128 : CG_PROF_HCONV_64, g, st);
#define HC_LAUNCH2(BN_, TWL_, FUSE_)                                                     \
  do {                                                                                   \
    if (relu) hconv_kernel<BN_, true, TWL_, FUSE_><<<grid, 512, 0, st>>>(a);             \
    else hconv_kernel<BN_, false, TWL_, FUSE_><<<grid, 512, 0, st>>>(a);                 \
  } while (0)
#define HC_LAUNCH(BN_, TWL_)                                                             \
  do {                                                                                   \
    if (BN_ == 64 && narrow) HC_LAUNCH2(64, TWL_, 3);                                    \
    else if (a.pool) HC_LAUNCH2(BN_, TWL_, 2);                                           \
    else if (a.bn_mean || a.stats) HC_LAUNCH2(BN_, TWL_, 1);                             \
    else HC_LAUNCH2(BN_, TWL_, 0);                                                       \
  } while (0)
  if (bn == 128) {
    if (twl == 5) HC_LAUNCH(128, 5);
    else HC_LAUNCH(128, 4);
  } else {
    if (twl == 5) HC_LAUNCH(64, 5);
    else HC_LAUNCH(64, 4);
  }
#undef HC_LAUNCH2
#undef HC_LAUNCH
}

// ---- weight gradient ----
namespace {
struct HWgradPlan {
  int twl, tiles_x, tiles_y, nslices, tiles, splits, sps;
};
bool hwgrad_geom_ok(const cgConvGeom* g) {
  return g->S == 1 && g->U == 1 && g->kh == 3 && g->kw == 3 && g->Ho == g->Hin &&
         g->Wo == g->Win && g->pt == 1 && g->pl == 1 && (g->Ci % 32) == 0 && (g->Co % 8) == 0 &&
         g->Co >= 32 && hc_tile_log(g->Ho, g->Wo) != 0 &&
         (int64_t)12 * g->Win * (g->Ci > g->Co ? g->Ci : g->Co) * 2 < (1ll << 31);
}
HWgradPlan hwgrad_plan(const cgConvGeom* g) {
  HWgradPlan p;
  p.twl = hc_tile_log(g->Ho, g->Wo);
  const int TW = 1 << p.twl, TH = 256 >> p.twl;
  p.tiles_x = g->Wo / TW;
  p.tiles_y = g->Ho / TH;
  p.nslices = g->N * p.tiles_y * p.tiles_x;
  p.tiles = cdiv(g->Ci, 64) * cdiv(g->Co, 64);
  // one workgroup per CU (150 KB of LDS each): every split costs a K x Co fp32 partial.  512
  // workgroups -- two rounds of half-size work items -- when each still walks >= 96 slices (BigGAN at
  // 256 per GPU: 128^2 x 96 ... 16^2 x 768 on 512 images): measured, hwgrad 29.5 -> 23.6 ms per step
  // there (603 -> 770 TFLOP/s), step 194.9 -> 188.7 ms -- finer items even out the tail of a launch
  // whose (ci, co) tiles are unevenly filled (96 = 64 + 32 channels); with fewer slices per workgroup
  // the extra partials cost more than that buys (cifar 6.67 -> 7.15 ms, ResNet5 D-step 5.12 ->
  // 5.29 ms at 512 everywhere: profiles/r05_hwgrad_blocks_ab.txt).
  // CGAMD_HWGRAD_BLOCKS > 0 fixes the target.
  static const int forced = hc_env("CGAMD_HWGRAD_BLOCKS", 0);
  const int target = forced > 0 ? forced : ((int64_t)p.nslices * p.tiles >= 512 * 96 ? 512 : 256);
  int s = cdiv(target, p.tiles);
  if (s > p.nslices) s = p.nslices;
  if (s < 1) s = 1;
  p.sps = cdiv(p.nslices, s);
  p.splits = cdiv(p.nslices, p.sps);
  return p;
}
}  // namespace

bool cg_hwgrad_supported(const cgConvGeom* g, const void* in, const void* gate_in, float slope_in,
                         const void* gate_dy) {
  static const int enabled = hc_env("CGAMD_HWGRAD", 1);
  static const int min_work = hc_env("CGAMD_HWGRAD_MIN", 128);
  if (!enabled || gate_dy || !hwgrad_geom_ok(g)) return false;
  if (gate_in && !(gate_in == in && slope_in == 0.f)) return false;
  const HWgradPlan p = hwgrad_plan(g);
  return (int64_t)p.nslices * p.tiles >= min_work;
}

size_t cg_hwgrad_workspace_bytes(const cgConvGeom* g) {
  if (!hwgrad_geom_ok(g)) return 0;
  const HWgradPlan p = hwgrad_plan(g);
  const size_t K = (size_t)9 * g->Ci;
  return align_up(p.splits > 1 ? (size_t)p.splits * (K + 1) * g->Co * sizeof(float) : 256, 256);
}

void cg_hwgrad_launch(const cgConvGeom* g, const void* in, const void* gate_in, const void* dy,
                      float* dw, int accumulate, float* dbias, void* ws, hipStream_t st) {
  cg_hwgrad_launch_pooled(g, in, gate_in, dy, 0, dw, accumulate, dbias, ws, st);
}

void cg_hwgrad_launch_pooled(const cgConvGeom* g, const void* in, const void* gate_in,
                             const void* dy, int dy_pooled, float* dw, int accumulate,
                             float* dbias, void* ws, hipStream_t st) {
  const HWgradPlan p = hwgrad_plan(g);
  HWgradArgs h;
  h.dy_up = dy_pooled ? 1 : 0;
  h.out_scale = dy_pooled ? 0.25f : 1.f;
  static const int asm_dma = hc_env("CGAMD_ASM_DMA", 1);
  h.asm_dma = asm_dma;
  h.in = (const bf16_t*)in;
  h.dy = (const bf16_t*)dy;
  h.N = g->N; h.Hin = g->Hin; h.Win = g->Win; h.Ci = g->Ci; h.Co = g->Co;
  h.pt = g->pt; h.pl = g->pl;
  h.K = 9 * g->Ci;
  h.ntiles = cdiv(g->Co, 64);
  h.tiles_x = p.tiles_x; h.tiles_y = p.tiles_y; h.nslices = p.nslices;
  h.slices_per_split = p.sps;
  h.accumulate = accumulate;
  h.dNt = make_fastdiv(h.ntiles);
  h.dTx = make_fastdiv(p.tiles_x);
  h.dTy = make_fastdiv(p.tiles_y);
  float* wsf = (float*)ws;
  const size_t KC = (size_t)h.K * g->Co;
  h.out = p.splits == 1 ? dw : wsf;
  h.bias = !dbias ? nullptr : (p.splits == 1 ? dbias : wsf + (size_t)p.splits * KC);
  dim3 grid(p.tiles, p.splits);
  CgProfScope prof(CG_PROF_HWGRAD, g, st);
  const bool relu = gate_in != nullptr;
#define HW_LAUNCH(TWL_)                                                               \
  do {                                                                                \
    if (asm_dma) {                                                                    \
      if (relu) hwgrad_kernel<true, TWL_, true><<<grid, 512, 0, st>>>(h);             \
      else hwgrad_kernel<false, TWL_, true><<<grid, 512, 0, st>>>(h);                 \
    } else {                                                                          \
      if (relu) hwgrad_kernel<true, TWL_, false><<<grid, 512, 0, st>>>(h);            \
      else hwgrad_kernel<false, TWL_, false><<<grid, 512, 0, st>>>(h);                \
    }                                                                                 \
  } while (0)
  if (p.twl == 5) HW_LAUNCH(5);
  else HW_LAUNCH(4);
#undef HW_LAUNCH
  if (p.splits > 1)
    cg_split_reduce4_pair(wsf, (int64_t)(KC / 4), dw, wsf + (size_t)p.splits * KC, g->Co / 4,
                          dbias, p.splits, accumulate, st);
}

// ---- image-input (stem) 3x3 convolutions ----
namespace {
bool wstem_geom_ok(const cgConvGeom* g) {
  return g->S == 1 && g->U == 1 && g->kh == 3 && g->kw == 3 && g->Ci == 3 && g->pt == 1 &&
         g->pl == 1 && g->Ho == g->Hin && g->Wo == g->Win && hc_tile_log(g->Ho, g->Wo) != 0 &&
         (int64_t)g->N * g->Ho * g->Wo < (1ll << 31);
}
void wstem_fill(const cgConvGeom* g, WStemArgs* a) {
  const int twl = hc_tile_log(g->Ho, g->Wo);
  const int TW = 1 << twl, TH = 256 >> twl;
  a->N = g->N; a->H = g->Hin; a->W = g->Win; a->Ci = g->Ci; a->Co = g->Co;
  a->tiles_x = g->Wo / TW;
  a->tiles_y = g->Ho / TH;
  a->ntiles = g->N * a->tiles_x * a->tiles_y;
  a->dTx = make_fastdiv(a->tiles_x);
  a->dTy = make_fastdiv(a->tiles_y);
}
int wstem_wgrad_splits(const cgConvGeom* g, int* tiles_per_split) {
  WStemArgs a;
  wstem_fill(g, &a);
  int s = a.ntiles < 512 ? a.ntiles : 512;
  const int tps = cdiv(a.ntiles, s);
  *tiles_per_split = tps;
  return cdiv(a.ntiles, tps);
}
}  // namespace

bool cg_wstem_conv_supported(const cgConvGeom* g, const void* in, const void* out,
                             const void* gate_in, float slope_in, const void* gate_out,
                             const void* residual) {
  static const int enabled = hc_env("CGAMD_WSTEM", 1);
  if (!enabled || !wstem_geom_ok(g)) return false;
  if (g->Co != 64 && g->Co != 96 && g->Co != 128) return false;
  if ((gate_out && gate_out != out) || residual) return false;
  if (gate_in && !(gate_in == in && slope_in == 0.f)) return false;
  return true;
}

void cg_wstem_conv_launch(const cgConvGeom* g, const void* in, const void* bt, void* out,
                          int out_is_f32, const float* bias, const void* gate_in,
                          const void* gate_out, float slope_out, hipStream_t st) {
  cg_wstem_conv_launch_pool(g, in, bt, out, out_is_f32, bias, gate_in, gate_out, slope_out, 0, st);
}

void cg_wstem_conv_launch_pool(const cgConvGeom* g, const void* in, const void* bt, void* out,
                               int out_is_f32, const float* bias, const void* gate_in,
                               const void* gate_out, float slope_out, int pool, hipStream_t st) {
  WStemArgs a;
  wstem_fill(g, &a);
  a.pool = pool;
  a.dy_up = 0;
  a.in = (const bf16_t*)in; a.bt = (const bf16_t*)bt; a.dy = nullptr; a.out = out; a.bias = bias;
  a.relu_in = gate_in != nullptr; a.out_f32 = out_is_f32; a.self_gate = gate_out != nullptr;
  a.want_bias = 0; a.slope_out = slope_out;
  // persistent workgroups (4 per CU) once there are enough tiles: the window prefetch needs a loop
  static const int wg_target = hc_env("CGAMD_WSTEM_WGS", 1024);
  a.tiles_per_wg = a.ntiles >= 2 * wg_target ? cdiv(a.ntiles, wg_target) : (a.ntiles >= 2048 ? 2 : 1);
  const int grid = cdiv(a.ntiles, a.tiles_per_wg);
  const int twl = hc_tile_log(g->Ho, g->Wo);
  CgProfScope prof(CG_PROF_STEM_FWD, g, st);
#define WS_FWD(CT_)                                                             \
  do {                                                                          \
    if (a.pool) {                                                               \
      if (twl == 5) wstem_fwd_kernel<CT_, 5, true><<<grid, 256, 0, st>>>(a);    \
      else wstem_fwd_kernel<CT_, 4, true><<<grid, 256, 0, st>>>(a);             \
    } else {                                                                    \
      if (twl == 5) wstem_fwd_kernel<CT_, 5, false><<<grid, 256, 0, st>>>(a);   \
      else wstem_fwd_kernel<CT_, 4, false><<<grid, 256, 0, st>>>(a);            \
    }                                                                           \
  } while (0)
  if (g->Co == 64) WS_FWD(2);
  else if (g->Co == 96) WS_FWD(3);
  else WS_FWD(4);
#undef WS_FWD
}

bool cg_wstem_wgrad_supported(const cgConvGeom* g, const void* in, const void* gate_in,
                              float slope_in, const void* gate_dy) {
  static const int enabled = hc_env("CGAMD_WSTEM", 1);
  if (!enabled || !wstem_geom_ok(g) || gate_dy || (g->Co % 8) != 0 || g->Co < 32) return false;
  if ((int64_t)10 * g->Win * g->Co * 2 >= (1ll << 31)) return false;
  if (gate_in && !(gate_in == in && slope_in == 0.f)) return false;
  return true;
}

size_t cg_wstem_wgrad_workspace_bytes(const cgConvGeom* g) {
  if (!wstem_geom_ok(g)) return 0;
  int tps;
  const int splits = wstem_wgrad_splits(g, &tps);
  return align_up((size_t)splits * ((size_t)27 * g->Co + g->Co) * sizeof(float), 256);
}

// partial layout [splits][K*Co + Co]; the caller (cg_conv_fast.hip) owns the strided reduce
void cg_wstem_wgrad_launch(const cgConvGeom* g, const void* in, const void* gate_in,
                           const void* dy, int want_bias, void* ws, int* splits_out,
                           hipStream_t st) {
  cg_wstem_wgrad_launch_pooled(g, in, gate_in, dy, 0, want_bias, ws, splits_out, st);
}

void cg_wstem_wgrad_launch_pooled(const cgConvGeom* g, const void* in, const void* gate_in,
                                  const void* dy, int dy_pooled, int want_bias, void* ws,
                                  int* splits_out, hipStream_t st) {
  WStemArgs a;
  wstem_fill(g, &a);
  a.pool = 0;
  a.dy_up = dy_pooled ? 1 : 0;
  int tps;
  const int splits = wstem_wgrad_splits(g, &tps);
  a.in = (const bf16_t*)in; a.bt = nullptr; a.dy = (const bf16_t*)dy; a.out = ws; a.bias = nullptr;
  a.relu_in = gate_in != nullptr; a.out_f32 = 1; a.self_gate = 0; a.slope_out = 0.f;
  a.want_bias = want_bias;
  a.tiles_per_wg = tps;
  dim3 grid(cdiv(g->Co, 64), splits);
  static const int asm_dma = hc_env("CGAMD_ASM_DMA", 1);
  if (asm_dma) {
    if (hc_tile_log(g->Ho, g->Wo) == 5) wstem_wgrad_kernel<5, true><<<grid, 256, 0, st>>>(a);
    else wstem_wgrad_kernel<4, true><<<grid, 256, 0, st>>>(a);
  } else {
    if (hc_tile_log(g->Ho, g->Wo) == 5) wstem_wgrad_kernel<5, false><<<grid, 256, 0, st>>>(a);
    else wstem_wgrad_kernel<4, false><<<grid, 256, 0, st>>>(a);
  }
  *splits_out = splits;
}

// ---- 64 -> 64 channel 3x3: register-resident weights ----
bool cg_hconv_rw_supported(const cgConvGeom* g, const void* in, const void* gate_in,
                           float slope_in) {
  static const int enabled = hc_env("CGAMD_HCONV_RW", 1);
  static const int min_tiles = hc_env("CGAMD_HCONV_RW_MIN", 512);
  if (!enabled) return false;
  if (g->S != 1 || g->U != 1 || g->kh != 3 || g->kw != 3 || g->Ci != 64 || g->Co != 64) return false;
  if (g->Ho != g->Hin || g->Wo != g->Win || g->pt != 1 || g->pl != 1) return false;
  if ((g->Wo % 32) != 0 || (g->Ho % 4) != 0) return false;
  if (gate_in && !(gate_in == in && slope_in == 0.f)) return false;
  if ((int64_t)6 * g->Win * 64 * 2 >= (1ll << 31)) return false;
  return (int64_t)g->N * (g->Ho / 4) * (g->Wo / 32) >= min_tiles;
}

static bool hconv_rw_geom_ok(const cgConvGeom* g) {
  if (g->S != 1 || g->U != 1 || g->kh != 3 || g->kw != 3 || g->Ci != 64 || g->Co != 64) return false;
  if (g->Ho != g->Hin || g->Wo != g->Win || g->pt != 1 || g->pl != 1) return false;
  if ((g->Wo % 32) != 0 || (g->Ho % 4) != 0) return false;
  if ((int64_t)6 * g->Win * 64 * 2 >= (1ll << 31)) return false;
  return true;
}

void cg_hconv_rw_launch(const cgConvGeom* g, const void* in, const void* bt, void* out,
                        int out_is_f32, const float* bias, const void* gate_in,
                        const void* gate_out, float slope_out, const void* residual,
                        hipStream_t st) {
  hconv_rw_launch_ex(g, in, bt, out, out_is_f32, bias, gate_in, gate_out, slope_out, residual, 0, 0,
                     1.f, st);
}

static void hconv_rw_launch_ex(const cgConvGeom* g, const void* in, const void* bt, void* out,
                               int out_is_f32, const float* bias, const void* gate_in,
                               const void* gate_out, float slope_out, const void* residual,
                               int pool, int in_up, float out_scale, hipStream_t st) {
  HConvArgs a;
  memset(&a, 0, sizeof(a));
  a.pool = pool;
  a.in_up = in_up;
  a.out_scale = out_scale;
  a.in = (const bf16_t*)in;
  a.bt = (const bf16_t*)bt;
  a.out = out;
  a.bias = bias;
  a.self_gate = (gate_out != nullptr && gate_out == out);
  a.gate_out = a.self_gate ? nullptr : (const bf16_t*)gate_out;
  a.residual = (const bf16_t*)residual;
  a.N = g->N; a.Hin = g->Hin; a.Win = g->Win; a.Ci = g->Ci;
  a.Ho = g->Ho; a.Wo = g->Wo; a.Co = g->Co; a.kh = 3; a.kw = 3;
  a.U = 1; a.pt = g->pt; a.pl = g->pl;
  a.Kp = 9 * 64;
  a.cblocks = 1;
  a.tiles_x = g->Wo / 32;
  a.tiles_y = g->Ho / 4;
  a.ntiles = 1;
  a.out_f32 = out_is_f32;
  a.slope_out = slope_out;
  a.dNt = make_fastdiv(1);
  a.dTx = make_fastdiv(a.tiles_x);
  a.dTy = make_fastdiv(a.tiles_y);
  const int ntiles = g->N * a.tiles_y * a.tiles_x;
  const int grid = ntiles < 512 ? ntiles : 512;
#ifdef CG_CONV_TIMING
  a.tdbg = g_hconv_tdbg;
#endif
  CgProfScope prof(CG_PROF_HCONV_64, g, st);
  if (pool) {
    if (gate_in) hconv_rw_kernel<true, true><<<grid, 256, 0, st>>>(a);
    else hconv_rw_kernel<false, true><<<grid, 256, 0, st>>>(a);
  } else {
    if (gate_in) hconv_rw_kernel<true, false><<<grid, 256, 0, st>>>(a);
    else hconv_rw_kernel<false, false><<<grid, 256, 0, st>>>(a);
  }
}
