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
//  * epilogue through LDS (bias, activation, gate, residual) writing whole 16-byte channel groups.
#include "cg_conv_fast.h"

#include <stdlib.h>

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
  FastDiv dNt, dTx, dTy;
};

__device__ __forceinline__ int hc_xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

__device__ __forceinline__ void hc_dma16(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff,
                                         unsigned char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)lds_wave_base, 16, voff, soff, 0, 0);
}

__device__ __forceinline__ bf16x8_t hc_relu(bf16x8_t v) {
  s16x8_t s = __builtin_bit_cast(s16x8_t, v);
  const s16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
  s = __builtin_elementwise_max(s, z);
  return __builtin_bit_cast(bf16x8_t, s);
}

// BN: output channels per workgroup (128 or 64); TWL: log2 of the tile width (5: 8x32, 4: 16x16)
template <int BN, bool RELU, int TWL>
__global__ __launch_bounds__(512, 4) void hconv_kernel(HConvArgs a) {
  constexpr int TW = 1 << TWL, TH = 256 >> TWL, PITCH = TW + 2;
  constexpr int TN = BN / 64;         // 32-channel MFMA tiles per wave (2 waves along channels)
  constexpr int BJ = BN / 64;         // weight staging pieces per wave and K-slice
  constexpr int B_BYTES = BN * 128;   // one K-slice of weights: BN rows x 64 k x 2 B
  constexpr int LDC = BN + 4;         // epilogue staging row (floats)
  constexpr int LDS_BYTES = HC_HALO_BYTES + 2 * B_BYTES;
  static_assert(128 * LDC * 4 <= LDS_BYTES, "epilogue staging does not fit");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31, half = lane >> 5;

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
  const int npieces = (HH * PITCH + 7) >> 3;             // <= HC_HALO_PIECES

  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bt =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.bt, 0, a.bt_bytes, 0x00020000);

  // ---- staging descriptors: byte offsets, computed once ----
  // halo piece p = wave + 8 j covers halo rows 8 p .. 8 p + 7; lane -> row 8 p + (lane >> 3), LDS
  // chunk (lane & 7) which must hold source chunk (lane & 7) ^ ((hx >> 1) & 7)
  uint32_t hvoff[HC_HSLOTS];
  {
    const int iy0 = ty * TH + bh, ix0 = tx * TW + bw;
#pragma unroll
    for (int j = 0; j < HC_HSLOTS; ++j) {
      const int row = (wave + 8 * j) * 8 + (lane >> 3);
      const int hy = row / PITCH, hx = row - hy * PITCH;
      const int c = (lane & 7) ^ ((hx >> 1) & 7);
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = hy < HH && hx < HWID && (unsigned)iy < (unsigned)a.Hin &&
                      (unsigned)ix < (unsigned)a.Win;
      hvoff[j] = ok ? (uint32_t)((((n * a.Hin + iy) * a.Win + ix) * a.Ci + c * 8) * 2) : HC_OOB;
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
#pragma unroll
    for (int j = 0; j < HC_HSLOTS; ++j)
      if (wave + 8 * j < npieces)
        hc_dma16(rs_in, hvoff[j], (uint32_t)(cb * 128), smem + (wave + 8 * j) * 1024);
  };
  auto issue_b = [&](int slot, int koff) {
#pragma unroll
    for (int j = 0; j < BJ; ++j)
      hc_dma16(rs_bt, bvoff[j], (uint32_t)(koff * 2),
               smem + HC_HALO_BYTES + slot * B_BYTES + (wave * BJ + j) * 1024);
  };

  // ---- fragment addressing ----
  // pixel p = wm*64 + i*32 + frow of the tile -> (y, x); halo row of its tap-(0,0) input pixel
  int hb[2], hx0[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = wm * 64 + i * 32 + frow;
    const int y = p >> TWL, x = p & (TW - 1);
    hb[i] = (y * PITCH + x) * 128;
    hx0[i] = x;
  }
  // weights: row = wn*(BN/2) + j*32 + frow, chunk (kk*2 + half) ^ ((frow >> 1) & 7)
  int bko[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
    bko[kk] = (wn * (BN / 2) + frow) * 128 + (((kk * 2 + half) ^ ((frow >> 1) & 7)) << 4);

  f32x16_t acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  // ---- main loop: K-slice = (channel block, tap); the halo is staged once per channel block ----
  issue_halo(0);
  issue_b(0, ((r0 * a.kw) + s0) * a.Ci);
  int tap = 0, cb = 0, ri = 0, si = 0;
  for (int it = 0; it < nk; ++it) {
    // this wave's pieces of slice `it` (and of the halo, on a block's first tap) have landed; after
    // the barrier so have everybody's, and every wave is done with the weight slot restaged below
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
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

    const unsigned char* Bs = smem + HC_HALO_BYTES + (it & 1) * B_BYTES;
    const int tshift = (ri * PITCH + si) * 128;
    int abase[2], aswz[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      abase[i] = hb[i] + tshift;
      aswz[i] = ((hx0[i] + si) >> 1) & 7;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8_t af[2], bfr[TN];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const bf16x8_t*>(smem + abase[i] +
                                                   (((kk * 2 + half) ^ aswz[i]) << 4));
        if (RELU) af[i] = hc_relu(af[i]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8_t*>(Bs + bko[kk] + j * 32 * 128);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (ntap == 0 && ncb < a.cblocks) {
      // channel block finished: every wave is done with the halo image before it is overwritten
      asm volatile("s_barrier" ::: "memory");
      issue_halo(ncb);
    }
    tap = ntap;
    cb = ncb;
    ri = nri;
    si = nsi;
  }

  // ---- epilogue through LDS in two passes of 128 pixels: accumulators (fp32) -> LDS, then every
  // thread finishes 8 consecutive channels of one pixel (bias, activation, gate, residual) and
  // writes 16 (bf16) / 32 (fp32) contiguous bytes ----
  constexpr int C8 = BN / 8;
  float* Cs = reinterpret_cast<float*>(smem);
  const int c8 = tid & (C8 - 1);
  const int co = n0 + c8 * 8;
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = 0.f;
  if (a.bias && co < a.Co) {
    const float4 b0 = *reinterpret_cast<const float4*>(a.bias + co);
    const float4 b1 = *reinterpret_cast<const float4*>(a.bias + co + 4);
    bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
    bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
  }
  for (int h = 0; h < 2; ++h) {
    __syncthreads();
    if ((wm >> 1) == h) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = wn * (BN / 2) + j * 32 + q * 8 + 4 * half;
            const int row = (wm & 1) * 64 + i * 32 + frow;
            *reinterpret_cast<float4*>(Cs + row * LDC + col) =
                make_float4(acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2],
                            acc[i][j][q * 4 + 3]);
          }
    }
    __syncthreads();
    if (co < a.Co) {
      for (int row = tid / C8; row < 128; row += 512 / C8) {
        const int p = h * 128 + row;
        const int y = p >> TWL, x = p & (TW - 1);
        const int oy = (ty * TH + y) * a.U + ph, ox = (tx * TW + x) * a.U + pw;
        const int64_t o = ((int64_t)(n * a.Ho + oy) * a.Wo + ox) * a.Co + co;
        const float4 lo = *reinterpret_cast<const float4*>(Cs + row * LDC + c8 * 8);
        const float4 hi = *reinterpret_cast<const float4*>(Cs + row * LDC + c8 * 8 + 4);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bv[e];
        if (a.self_gate) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (!(v[e] > 0.f)) v[e] *= a.slope_out;
        }
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

bool cg_hconv_supported(const cgConvGeom* g, const void* in, const void* gate_in, float slope_in) {
  static const int enabled = hc_env("CGAMD_HCONV", 1);
  static const int min_wgs = hc_env("CGAMD_HCONV_MIN", 200);
  if (!enabled) return false;
  if (g->S != 1 || (g->U != 1 && g->U != 2)) return false;
  if (g->kh > 3 || g->kw > 3) return false;
  if (g->Ci % 64 != 0 || g->Co % 8 != 0 || g->Co < 64) return false;
  if (g->Ho % g->U || g->Wo % g->U) return false;
  const int Hp = g->Ho / g->U, Wp = g->Wo / g->U;
  if (Hp != g->Hin || Wp != g->Win) return false;   // 'SAME' unit-stride geometry only
  const int twl = hc_tile_log(Hp, Wp);
  if (!twl) return false;
  if (gate_in && !(gate_in == in && slope_in == 0.f)) return false;
  if ((int64_t)g->N * g->Hin * g->Win * g->Ci * 2 >= (1ll << 31)) return false;
  if ((int64_t)g->Co * (((int64_t)g->kh * g->kw * g->Ci + 7) & ~7ll) * 2 >= (1ll << 31)) return false;
  const int bn = g->Co <= 64 ? 64 : 128;
  const int64_t wgs = (int64_t)g->N * (Hp * Wp / 256) * cdiv(g->Co, bn) * g->U * g->U;
  return wgs >= min_wgs;
}

void cg_hconv_launch(const cgConvGeom* g, const void* in, const void* bt, void* out,
                     int out_is_f32, const float* bias, const void* gate_in, const void* gate_out,
                     float slope_out, const void* residual, hipStream_t st) {
  HConvArgs a;
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
  a.cblocks = g->Ci / 64;
  a.in_bytes = (uint32_t)((int64_t)g->N * g->Hin * g->Win * g->Ci * 2);
  a.bt_bytes = (uint32_t)((int64_t)g->Co * a.Kp * 2);
  const int Hp = g->Ho / g->U, Wp = g->Wo / g->U;
  const int twl = hc_tile_log(Hp, Wp);
  const int TW = 1 << twl, TH = 256 >> twl;
  a.tiles_x = Wp / TW;
  a.tiles_y = Hp / TH;
  const int bn = g->Co <= 64 ? 64 : 128;
  a.ntiles = cdiv(g->Co, bn);
  a.out_f32 = out_is_f32;
  a.slope_out = slope_out;
  a.dNt = make_fastdiv(a.ntiles);
  a.dTx = make_fastdiv(a.tiles_x);
  a.dTy = make_fastdiv(a.tiles_y);
  const bool relu = gate_in != nullptr;
  dim3 grid(g->N * a.tiles_y * a.tiles_x * a.ntiles, g->U * g->U);
  CgProfScope prof(bn == 128 ? CG_PROF_HCONV_128 : CG_PROF_HCONV_64, g, st);
#define HC_LAUNCH(BN_, TWL_)                                                        \
  do {                                                                              \
    if (relu) hconv_kernel<BN_, true, TWL_><<<grid, 512, 0, st>>>(a);               \
    else hconv_kernel<BN_, false, TWL_><<<grid, 512, 0, st>>>(a);                   \
  } while (0)
  if (bn == 128) {
    if (twl == 5) HC_LAUNCH(128, 5);
    else HC_LAUNCH(128, 4);
  } else {
    if (twl == 5) HC_LAUNCH(64, 5);
    else HC_LAUNCH(64, 4);
  }
#undef HC_LAUNCH
}
