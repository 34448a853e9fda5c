// Small-map 3x3 convolution for gfx950: unit-stride 'SAME' 3x3 filters on feature maps whose
// launches cannot fill the chip with 256-pixel tiles (4x4 ... 16x16 maps, or few images).
// Contract and reference call sites: include/cgamd.h (cg_gconv: arch_ops.conv2d, arch_ops.py:559-573,
// and its data gradient through the adjoint geometry); this file only adds a faster kernel behind
// the same entry point.  The layers it serves: the 8x8 / 16x16 blocks of the ResNet-CIFAR
// discriminator (resnet_cifar.py:119-167; 72 launches per train step), blocks B4 / B5 of the
// ResNet5 discriminator (resnet5.py:99-145: 8x8 and 4x4 maps, 512 channels) and the first generator
// blocks.
//
// Why another kernel (profiles/r03_small_conv_ab.txt): on these shapes the K loop of the tiled
// kernels is short and every step of it is serial -- wait for the slice, workgroup barrier, issue the
// next slice, multiply -- so a launch costs 20-45 us for 2-10 GFLOP.  Here
//  * a workgroup owns 64 output pixels (one 8x8 tile, or four 4x4 images) x 64 output channels, so
//    a 128-image 8x8 layer with 128 channels is 256 workgroups -- one per CU;
//  * the 4 compute waves split K, not the tile: wave w multiplies the K units u = w (mod 4), a unit
//    being 32 channels of one filter tap, into its own full 64x64 accumulator.  Each wave streams
//    ITS OWN weight units through a private LDS ring with counted vmcnt waits, so the K loop has no
//    workgroup barrier and no wave ever waits for another wave's loads; fragment traffic is 1 KiB
//    of LDS reads per MFMA (a 64x64 wave tile), half of what a 2x2 wave grid of 32x32 tiles needs;
//  * the input window with its halo (10x10 pixels x 64 channels = 12.5 KiB) is staged once per
//    64-channel block by a FIFTH wave that does nothing else (its own vmcnt queue: the compute
//    waves' counted waits never include a halo piece), double-buffered, one s_barrier per channel
//    block hands a window over;
//  * LDS images are lane-linear (buffer_load ... lds), swizzled on the source side: window rows are
//    128 B with 16-byte chunk c at c ^ (((x >> 1) & 3) | ((y & 1) << 2)) (8x8 tiles; x, y = halo
//    coordinates), weight-unit rows are 64 B with chunk c at c ^ ((row >> 2) & 3): every
//    ds_read_b128 lane group hits 16 distinct 16-byte bank groups for every tap shift (checked
//    exhaustively by tests/test_host_boundary.py::test_small_conv_swizzle_is_conflict_free);
//  * the four partial accumulators are summed through LDS in a fixed order (deterministic), then the
//    usual epilogue: bias, output gate, residual, one rounding to bf16, 16-byte stores.
#include "cg_conv_fast.h"

#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void sc_lds_void_t;
constexpr uint32_t SC_OOB = 0x80000000u;   // voffset of a lane that must read zeros (bounds check)

struct SConvArgs {
  const bf16_t* in;
  const bf16_t* bt;
  void* out;
  const float* bias;
  const bf16_t* gate_out;
  const bf16_t* residual;
  uint32_t in_bytes, bt_bytes;
  int N, H, W, Ci, Co;
  int Kp, cblocks;
  int tiles_x, tiles_y, ntiles;
  int out_f32, self_gate;
  float slope_out;
  FastDiv dNt, dTx, dTy;
};

__device__ __forceinline__ int sc_xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

__device__ __forceinline__ void sc_dma16(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff,
                                         unsigned char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (sc_lds_void_t*)lds_wave_base, 16, voff, soff, 0, 0);
}

__device__ __forceinline__ bf16x8_t sc_relu(bf16x8_t v) {
  s16x8_t s = __builtin_bit_cast(s16x8_t, v);
  const s16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
  s = __builtin_elementwise_max(s, z);
  return __builtin_bit_cast(bf16x8_t, s);
}

// TWL: log2 of the tile edge: 3 = one 8x8 tile of one image, 2 = four whole 4x4 images
template <int TWL, bool RELU>
__global__ __launch_bounds__(320) void sconv_kernel(SConvArgs a) {
  constexpr int TW = 1 << TWL, TH = TW, NI = 64 / (TW * TH), P = TW + 2, HH = TH + 2;
  constexpr int IMG_ROWS = HH * P;                 // window rows of one image: 100 / 36
  constexpr int HROWS = NI * IMG_ROWS;             // 100 / 144
  constexpr int HPIECES = (HROWS + 7) / 8;         // 1-KiB pieces of 8 rows: 13 / 18
  constexpr int HSLOT = HPIECES * 1024;
  constexpr int UNIT = 4096;                       // one weight unit: 64 out-channels x 32 k x 2 B
  constexpr int RING = 3;
  constexpr int B_OFF = 2 * HSLOT;
  constexpr int LDS_MAIN = B_OFF + 4 * RING * UNIT;
  constexpr int SP = 64 * 4 + 16;                  // epilogue staging row pitch (bytes)
  constexpr int STAGE = 4 * 64 * SP;
  constexpr int LDS_BYTES = LDS_MAIN > STAGE ? LDS_MAIN : STAGE;
  constexpr int XM = TWL == 3 ? 3 : 1;             // swizzle: x bits
  constexpr int YB = TWL == 3 ? 2 : 1;             //          position of the y-parity bit
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- workgroup -> (pixel tile, out-channel tile), XCD-contiguous ----
  const int wg = sc_xcd_remap(blockIdx.x, gridDim.x);
  const int ptile = (int)fdiv((uint32_t)wg, a.dNt);
  const int nt = wg - ptile * a.ntiles;
  int n0, ty = 0, tx = 0;
  if (TWL == 3) {
    const int t1 = (int)fdiv((uint32_t)ptile, a.dTx);
    tx = ptile - t1 * a.tiles_x;
    n0 = (int)fdiv((uint32_t)t1, a.dTy);
    ty = t1 - n0 * a.tiles_y;
  } else {
    n0 = ptile * NI;
  }
  const int co0 = nt * 64;
  const int ncb = a.cblocks;

  if (wave == 4) {
    // ================= loader wave: the input windows, one per 64-channel block =================
    const __amdgpu_buffer_rsrc_t rs_in =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
    uint32_t hv[HPIECES];
#pragma unroll
    for (int j = 0; j < HPIECES; ++j) {
      const int row = j * 8 + (lane >> 3);
      const int il = row / IMG_ROWS, rem = row - il * IMG_ROWS;
      const int hy = rem / P, hx = rem - hy * P;
      const int g = ((hx >> 1) & XM) | ((hy & 1) << YB) | (TWL == 2 ? ((il & 1) << 2) : 0);
      const int c = (lane & 7) ^ g;
      const int n = n0 + il, iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = row < HROWS && n < a.N && (unsigned)iy < (unsigned)a.H &&
                      (unsigned)ix < (unsigned)a.W;
      hv[j] = ok ? (uint32_t)((((n * a.H + iy) * a.W + ix) * a.Ci + c * 8) * 2) : SC_OOB;
    }
    auto issue_halo = [&](int slot, int cb) {
#pragma unroll
      for (int j = 0; j < HPIECES; ++j)
        sc_dma16(rs_in, hv[j], (uint32_t)(cb * 128), smem + slot * HSLOT + j * 1024);
    };
    issue_halo(0, 0);
    if (ncb > 1) {
      issue_halo(1, 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HPIECES) : "memory");   // window 0 landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");                            // start
    for (int cb = 0; cb + 1 < ncb; ++cb) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // window cb + 1 landed
      asm volatile("s_barrier" ::: "memory");                          // everyone is done with window cb
      if (cb + 2 < ncb) issue_halo(cb & 1, cb + 2);
    }
    asm volatile("s_barrier" ::: "memory");                            // epilogue barriers
    asm volatile("s_barrier" ::: "memory");
    return;
  }

  // ===================== compute waves: K units u = wave (mod 4) =====================
  const int frow = lane & 31, half = lane >> 5;
  const __amdgpu_buffer_rsrc_t rs_bt =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.bt, 0, a.bt_bytes, 0x00020000);
  uint32_t bv[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = p * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    bv[p] = (uint32_t)(((co0 + row) * a.Kp + c * 8) * 2);
  }
  unsigned char* Bring = smem + B_OFF + wave * (RING * UNIT);
  // unit g -> (channel block, tap, half): k offset of its weight unit in a row of bt
  auto issue_b = [&](int slot, int cb, int q) {
    const int koff = (q >> 1) * a.Ci + cb * 64 + (q & 1) * 32;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      sc_dma16(rs_bt, bv[p], (uint32_t)(koff * 2), Bring + slot * UNIT + p * 1024);
  };
  const int total = 18 * ncb;
  // the first two units leave before the rest of the set-up
  {
    issue_b(0, 0, wave);
    const int g1 = wave + 4;           // < 18: same channel block
    if (g1 < total) issue_b(1, 0, g1);
  }

  // ---- fragment addressing ----
  int ab[2], px[2], py[2], gil[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = i * 32 + frow;
    const int il = p >> (2 * TWL), y = (p >> TWL) & (TH - 1), x = p & (TW - 1);
    ab[i] = (il * IMG_ROWS + y * P + x) * 128;
    px[i] = x;
    py[i] = y;
    gil[i] = TWL == 2 ? ((il & 1) << 2) : 0;
  }
  int bo[2];
#pragma unroll
  for (int k2 = 0; k2 < 2; ++k2) bo[k2] = frow * 64 + (((k2 * 2 + half) ^ ((frow >> 2) & 3)) << 4);

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  asm volatile("s_barrier" ::: "memory");   // start: window 0 is in LDS

  int cb = 0, q = wave, slot = 0;
  for (int g = wave; g < total; g += 4) {
    // B(g) landed (the unit issued after it may still be in flight)
    if (g + 4 < total) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
      // prefetch the unit two ahead into the slot of the unit just finished
      int q2 = q + 8, cb2 = cb;
      if (q2 >= 18) {
        q2 -= 18;
        ++cb2;
      }
      int s2 = slot + 2;
      if (s2 >= RING) s2 -= RING;
      if (g + 8 < total) issue_b(s2, cb2, q2);
    }
    const int tap = q >> 1, h = q & 1;
    const int ri = (tap * 11) >> 5, si = tap - 3 * ri;
    const unsigned char* As = smem + (cb & 1) * HSLOT + (ri * P + si) * 128;
    const unsigned char* Bs = Bring + slot * UNIT;
    int gsw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
      gsw[i] = (((px[i] + si) >> 1) & XM) | (((py[i] + ri) & 1) << YB) | gil[i];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const int kc = (h * 2 + k2) * 2 + half;
      bf16x8_t af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const bf16x8_t*>(As + ab[i] + ((kc ^ gsw[i]) << 4));
        if (RELU) af[i] = sc_relu(af[i]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8_t*>(Bs + j * 2048 + bo[k2]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    // next unit of this wave
    q += 4;
    if (++slot == RING) slot = 0;
    if (q >= 18) {
      q -= 18;
      ++cb;
      // channel block finished: the loader's next window is complete behind this barrier, and the
      // window just used may be overwritten
      if (cb < ncb) asm volatile("s_barrier" ::: "memory");
    }
  }

  // ---- epilogue: the 4 partial tiles are summed through LDS (fixed order), wave w finishes the
  // pixels 16 w .. 16 w + 15: 8 consecutive channels per lane, one rounding to bf16 ----
  const int g8 = lane & 7, rl = lane >> 3;
  const int co = co0 + g8 * 8;
  float bv8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv8[e] = 0.f;
  if (a.bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(a.bias + co);
    const float4 b1 = *reinterpret_cast<const float4*>(a.bias + co + 4);
    bv8[0] = b0.x; bv8[1] = b0.y; bv8[2] = b0.z; bv8[3] = b0.w;
    bv8[4] = b1.x; bv8[5] = b1.y; bv8[6] = b1.z; bv8[7] = b1.w;
  }
  asm volatile("s_barrier" ::: "memory");   // every wave is done with the windows and its ring
  unsigned char* Sw = smem + wave * (64 * SP);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
        *reinterpret_cast<float4*>(Sw + (i * 32 + frow) * SP + (j * 32 + qq * 8 + 4 * half) * 4) =
            make_float4(acc[i][j][qq * 4 + 0], acc[i][j][qq * 4 + 1], acc[i][j][qq * 4 + 2],
                        acc[i][j][qq * 4 + 3]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int row = wave * 16 + rl + 8 * k;   // pixel of the tile
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const unsigned char* src = smem + m * (64 * SP) + row * SP + g8 * 32;
      const float4 lo = *reinterpret_cast<const float4*>(src);
      const float4 hi = *reinterpret_cast<const float4*>(src + 16);
      v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w;
      v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
    }
    const int il = row >> (2 * TWL), y = (row >> TWL) & (TH - 1), x = row & (TW - 1);
    const int n = n0 + il;
    if (n >= a.N) continue;
    const int oy = ty * TH + y, ox = tx * TW + x;
    const int64_t o = ((int64_t)(n * a.H + oy) * a.W + ox) * a.Co + co;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += bv8[e];
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
}


// -------------------------------------------------------------------------------------------
// Weight gradient on small maps: dw[tap][ci][co] = sum_pixels x[pixel + tap][ci] * dy[pixel][co]
// (tf.gradients of arch_ops.conv2d w.r.t. the kernel, arch_ops.py:559-573) for 3x3 'SAME' filters
// when the whole batch has at most a few thousand pixels (N*H*W <= 8192: the 4x4 / 8x8 blocks).
// The tiled weight-gradient kernels split the pixel sum over workgroups to fill the chip and write
// fp32 partials (32 x the 0.59 MB gradient of a ResNet-CIFAR 8x8 layer: 6.3 x its algorithmic HBM
// traffic, profiles/r02_pmc_traffic.json) that a second launch reduces.  Here nothing is split
// across workgroups and nothing but dw is written:
//  * a workgroup owns ONE tap x 64 channels x 64 out-channels and walks ALL pixels; the grid is
//    9 * (Ci/64) * (Co/64) workgroups per layer, and SEVERAL layers travel in one launch (job table by
//    value in the kernel arguments: cg_gwgrad_multi), so the six 8x8 weight gradients of a
//    ResNet-CIFAR discriminator backward pass are one launch of 216 workgroups;
//  * the 4 waves split the pixels, not the tile: wave w multiplies the 16-pixel k-steps ks = w (mod
//    4) into its own 64x64 accumulator from a PRIVATE LDS ring (2 KiB of x rows + 2 KiB of dy rows
//    per k-step, buffer_load ... lds, counted vmcnt): no workgroup barrier in the loop; operands are
//    transposed out of the row-major LDS image by ds_read_b64_tr_b16 (row swizzle byte ^= (row & 2)
//    << 5, as hwgrad_kernel);
//  * the four accumulators are summed through LDS in a fixed order (deterministic) and written
//    (or added, `accumulate`) straight into dw; the bias gradient is one more MFMA per k-step with
//    an all-ones operand in the workgroups of tap 0 / channel tile 0.
// The kernel is bound by the 64 B/clk/CU fill path (256 B of operands per pixel for a 64x64 tile):
// x and dy are re-read 9 * Co/64 and 9 * Ci/64 times, from L2 -- they are a few MB on these layers.
// -------------------------------------------------------------------------------------------
struct SwJob {
  const bf16_t* x;
  const bf16_t* dy;
  float* dw;
  float* db;        // nullptr = no bias gradient
  uint32_t x_bytes, dy_bytes;
  int H, W, wl, hl; // map size and its logs (powers of two, W >= 4, H*W >= 16)
  int Ci, Co, nks;  // nks = N*H*W / 16 k-steps
  int relu, accumulate;
  int tiles_ci, tiles_co;
  int wg_begin;     // first workgroup of this job
};
constexpr int SW_MAX_JOBS = 24;
struct SwArgs {
  SwJob job[SW_MAX_JOBS];
  int njobs;
};

typedef __attribute__((ext_vector_type(4))) short sc_s16x4_t;
typedef __attribute__((address_space(3))) sc_s16x4_t* sc_tr_ptr;
typedef __attribute__((address_space(3))) unsigned char* sc_lds_ptr;

__device__ __forceinline__ bf16x8_t sc_tr_read2(sc_lds_ptr p) {
  const sc_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sc_tr_ptr)p);
  const sc_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sc_tr_ptr)(p + 512));
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

struct SwFrag {
  bf16x8_t x0, x1, y0, y1;
};

// the pixel loop of one wave: k-steps ks = wave, wave + 4, ... (cnt of them).  Ring of 4 k-step
// images; the fragments of step i + 1 are read while step i is multiplied, and step i + 4 is staged
// into the slot step i has just left.
template <bool RELU, bool BIAS, typename StageFn>
__device__ __forceinline__ void sw_pixel_loop(int cnt, int wave, sc_lds_ptr ring, int xa, int xb,
                                              StageFn stage, f32x16_t (&acc)[2][2],
                                              f32x16_t (&accb)[2]) {
  constexpr int KSTEP = 4096;
  const s16x8_t ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);
  auto read_frags = [&](SwFrag& f, int slot) {
    sc_lds_ptr so = ring + slot * KSTEP;
    f.x0 = sc_tr_read2(so + xa);
    f.x1 = sc_tr_read2(so + xb);
    f.y0 = sc_tr_read2(so + 2048 + xa);
    f.y1 = sc_tr_read2(so + 2048 + xb);
    if (RELU) {
      f.x0 = sc_relu(f.x0);
      f.x1 = sc_relu(f.x1);
    }
  };
  // step j of this wave has landed when only the steps staged after it are still in flight: steps
  // 1..3 behind step 0 (the prologue stages four), steps j + 1, j + 2 behind any later one (step
  // j + 3 is staged right after step j's fragments are read) -- fewer at the end of the loop
  auto wait_landed = [&](int j) {
    int later = cnt - 1 - j;
    const int depth = j == 0 ? 3 : 2;
    if (later > depth) later = depth;
    if (later >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto multiply = [&](const SwFrag& f) {
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.y0, f.x0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.y1, f.x0, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.y0, f.x1, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.y1, f.x1, acc[1][1], 0, 0, 0);
    if (BIAS) {
      accb[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.y0, ones, accb[0], 0, 0, 0);
      accb[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.y1, ones, accb[1], 0, 0, 0);
    }
  };
  if (cnt <= 0) return;
  SwFrag F0, F1;
  wait_landed(0);
  read_frags(F0, 0);
  int i = 0;
  // pairs of steps with a successor each
  for (; i + 2 < cnt; i += 2) {
    wait_landed(i + 1);
    read_frags(F1, (i + 1) & 3);
    if (i + 4 < cnt) stage(i & 3, wave + 4 * (i + 4));
    multiply(F0);
    wait_landed(i + 2);
    read_frags(F0, (i + 2) & 3);
    if (i + 5 < cnt) stage((i + 1) & 3, wave + 4 * (i + 5));
    multiply(F1);
  }
  // tail: one or two steps left, F0 holds step i
  if (i + 1 < cnt) {
    wait_landed(i + 1);
    read_frags(F1, (i + 1) & 3);
    multiply(F0);
    multiply(F1);
  } else {
    multiply(F0);
  }
}

__global__ __launch_bounds__(256) void swgrad_kernel(SwArgs a) {
  constexpr int KSTEP = 4096;                // one k-step: 16 pixels x (64 ch of x + 64 ch of dy)
  constexpr int RING = 4;
  constexpr int SP = 64 * 4 + 16;
  constexpr int STAGE = 4 * 64 * SP;         // four 64 x 64 fp32 tiles
  constexpr int LDS_BYTES = STAGE + 1024;    // + [4 waves][64] bias partials
  static_assert(4 * RING * KSTEP <= LDS_BYTES, "ring does not fit");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- workgroup -> job -> (tap, channel tile, out-channel tile) ----
  int ji = 0;
  for (int j = 1; j < a.njobs; ++j)
    if ((int)blockIdx.x >= a.job[j].wg_begin) ji = j;
  const SwJob& jb = a.job[ji];
  const int local = blockIdx.x - jb.wg_begin;
  const int per_tap = jb.tiles_ci * jb.tiles_co;
  const int tap = local / per_tap;
  const int rem = local - tap * per_tap;
  const int cit = rem / jb.tiles_co, cot = rem - cit * jb.tiles_co;
  const int ci0 = cit * 64, co0 = cot * 64;
  const int dr = tap / 3 - 1, ds = tap % 3 - 1;
  const int H = jb.H, W = jb.W, wl = jb.wl, hl = jb.hl, Ci = jb.Ci, Co = jb.Co, nks = jb.nks;

  const cg_i32x4_t rs_x = cg_make_rsrc(jb.x, jb.x_bytes);
  const cg_i32x4_t rs_y = cg_make_rsrc(jb.dy, jb.dy_bytes);

  // ---- staging: piece j of a k-step covers its pixels 8 j .. 8 j + 7 (128-byte rows), lane ->
  // row 8 j + (lane >> 3), LDS chunk (lane & 7) <- source chunk (lane & 7) ^ (((row >> 1) & 1) << 2)
  int xl[2], yl[2];
  uint32_t xlc[2], ylc[2];   // lane-constant byte offsets
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = j * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (((row >> 1) & 1) << 2);
    // pixel `row` of a k-step: W >= 16: same image row; W = 8: two rows; W = 4: four rows
    xl[j] = wl >= 4 ? row : (row & (W - 1));
    yl[j] = wl >= 4 ? 0 : (row >> wl);
    xlc[j] = (uint32_t)(((yl[j] * W + xl[j]) * Ci + ci0 + c * 8) * 2);
    ylc[j] = (uint32_t)((row * Co + co0 + c * 8) * 2);
  }
  unsigned char* ring = smem + wave * (RING * KSTEP);
  const uint32_t ring_addr = cg_lds_addr(ring);
  auto stage = [&](int slot, int ks) {
    // first pixel of the k-step: m0 = 16 ks -> (n, y0, x0), all wave-uniform
    const int m0 = ks * 16;
    const int x0 = m0 & (W - 1);
    const int r0 = m0 >> wl;
    const int y0 = r0 & (H - 1), n = r0 >> hl;
    const int sbase = (((n * H + y0 + dr) * W + x0 + ds) * Ci) * 2;
    const uint32_t dst = ring_addr + (uint32_t)(slot * KSTEP);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = (unsigned)(y0 + yl[j] + dr) < (unsigned)H &&
                      (unsigned)(x0 + xl[j] + ds) < (unsigned)W;
      const uint32_t vo = ok ? (uint32_t)(sbase + (int)xlc[j]) : SC_OOB;
      cg_dma16_asm_m0(rs_x, vo, 0u, dst + j * 1024);
    }
    const uint32_t ybase = (uint32_t)(m0 * Co * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) cg_dma16_asm_m0(rs_y, ybase + ylc[j], 0u, dst + 2048 + j * 1024);
  };

  // ---- transpose-read addressing inside a k-step image: channel sub-tile 0 at xa, sub-tile 1 at
  // xa ^ 64 (the row swizzle flips the same bit); the dy image follows 2048 bytes behind ----
  const int l16 = lane & 15;
  const int prow = (lane >> 5) * 8 + (l16 >> 2);
  const int tcolb = (((lane >> 4) & 1) * 16 + (l16 & 3) * 4) * 2;
  const int xa = prow * 128 + (tcolb ^ ((prow & 2) << 5));
  const int xb = xa ^ 64;

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
  f32x16_t accb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) accb[j][v] = 0.f;
  const bool want_bias = jb.db != nullptr && tap == 0 && cit == 0;   // wave-uniform
  const bool relu = jb.relu != 0;

  const int cnt = nks > wave ? (nks - wave + 3) >> 2 : 0;
#pragma unroll
  for (int d = 0; d < 4; ++d)
    if (d < cnt) stage(d, wave + 4 * d);
  sc_lds_ptr lring = (sc_lds_ptr)ring;
  if (want_bias) {
    if (relu) sw_pixel_loop<true, true>(cnt, wave, lring, xa, xb, stage, acc, accb);
    else sw_pixel_loop<false, true>(cnt, wave, lring, xa, xb, stage, acc, accb);
  } else {
    if (relu) sw_pixel_loop<true, false>(cnt, wave, lring, xa, xb, stage, acc, accb);
    else sw_pixel_loop<false, false>(cnt, wave, lring, xa, xb, stage, acc, accb);
  }

  // ---- epilogue: acc[i][j][v] = dw[ci = i*32 + (lane & 31)][co = j*32 + (v&3) + 8 (v>>2) + 4 half];
  // the four partial tiles are summed through LDS in a fixed order, wave w finishes rows 16 w ..
  const int frow = lane & 31, half = lane >> 5;
  __syncthreads();   // every wave is done with its ring
  unsigned char* Sw = smem + wave * (64 * SP);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
        *reinterpret_cast<float4*>(Sw + (i * 32 + frow) * SP + (j * 32 + qq * 8 + 4 * half) * 4) =
            make_float4(acc[i][j][qq * 4 + 0], acc[i][j][qq * 4 + 1], acc[i][j][qq * 4 + 2],
                        acc[i][j][qq * 4 + 3]);
  float* sb = reinterpret_cast<float*>(smem + STAGE);
  if (want_bias && frow == 0) {
    // every column of accb holds the row sums: column 0 lives in lanes 0 and 32
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v)
        sb[wave * 64 + j * 32 + (v & 3) + 8 * (v >> 2) + 4 * half] = accb[j][v];
  }
  __syncthreads();
  const int g8 = lane & 7, rl = lane >> 3;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int row = wave * 16 + rl + 8 * k;   // channel of the tile
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const unsigned char* src = smem + m * (64 * SP) + row * SP + g8 * 32;
      const float4 lo = *reinterpret_cast<const float4*>(src);
      const float4 hi = *reinterpret_cast<const float4*>(src + 16);
      v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w;
      v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
    }
    float* op = jb.dw + ((int64_t)tap * Ci + ci0 + row) * Co + co0 + g8 * 8;
    float4* o4 = reinterpret_cast<float4*>(op);
    if (jb.accumulate) {
      const float4 p0 = o4[0], p1 = o4[1];
      v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w;
      v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
    }
    o4[0] = make_float4(v[0], v[1], v[2], v[3]);
    o4[1] = make_float4(v[4], v[5], v[6], v[7]);
  }
  if (want_bias && tid < 64) {
    float s = sb[tid] + sb[64 + tid] + sb[128 + tid] + sb[192 + tid];
    float* bp = jb.db + co0 + tid;
    *bp = jb.accumulate ? *bp + s : s;
  }
}

int sc_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// 3: 8x8 tiles, 2: four 4x4 images per tile, 0: not covered
int sc_tile_log(const cgConvGeom* g) {
  if (g->Hin == 4 && g->Win == 4) return 2;
  if ((g->Hin % 8) == 0 && (g->Win % 8) == 0) return 3;
  return 0;
}

}  // namespace

bool cg_sconv_geom_ok(const cgConvGeom* g) {
  if (g->S != 1 || g->U != 1 || g->kh != 3 || g->kw != 3 || g->pt != 1 || g->pl != 1) return false;
  if (g->Ho != g->Hin || g->Wo != g->Win) return false;
  if ((g->Ci % 64) != 0 || (g->Co % 64) != 0) return false;
  if (!sc_tile_log(g)) return false;
  if ((int64_t)g->N * g->Hin * g->Win * g->Ci * 2 >= (1ll << 31)) return false;
  if ((int64_t)g->Co * 9 * g->Ci * 2 >= (1ll << 31)) return false;
  return true;
}

// Policy (scripts/check_small_conv.py, profiles/r03_small_conv_ab.txt): every workgroup streams its
// 64 out-channel weight rows once per 64 pixels, which costs as much fill-path time as the tile's
// MFMA work, so the kernel wins where the tiled kernels are latency-bound -- at most ~1.5 of its own
// workgroups per CU: 1.8-1.9 x on the 128 x 8x8 x 128 / 64 x 8x8 x 256 / 128 x 4x4 x 512 layers --
// and loses on larger grids (0.8 x at 1024 workgroups).  CGAMD_SCONV: 0 = off, 1 = policy
// (default), 2 = wherever the geometry fits.
bool cg_sconv_supported(const cgConvGeom* g, const void* in, const void* gate_in, float slope_in) {
  static const int mode = sc_env("CGAMD_SCONV", 1);
  static const int max_wgs = sc_env("CGAMD_SCONV_MAX", 384);
  if (!mode || !cg_sconv_geom_ok(g)) return false;
  if (gate_in && !(gate_in == in && slope_in == 0.f)) return false;
  if (mode == 2) return true;
  const int64_t pix = (int64_t)g->N * g->Ho * g->Wo;
  return cdiv(pix, 64) * (g->Co / 64) <= max_wgs;
}

void cg_sconv_launch(const cgConvGeom* g, const void* in, const void* bt, void* out, int out_is_f32,
                     const float* bias, const void* gate_in, const void* gate_out, float slope_out,
                     const void* residual, hipStream_t st) {
  SConvArgs a;
  a.in = (const bf16_t*)in;
  a.bt = (const bf16_t*)bt;
  a.out = out;
  a.bias = bias;
  a.self_gate = (gate_out != nullptr && gate_out == out);
  a.gate_out = a.self_gate ? nullptr : (const bf16_t*)gate_out;
  a.residual = (const bf16_t*)residual;
  a.N = g->N; a.H = g->Hin; a.W = g->Win; a.Ci = g->Ci; a.Co = g->Co;
  a.Kp = 9 * g->Ci;
  a.cblocks = g->Ci / 64;
  a.in_bytes = (uint32_t)((int64_t)g->N * g->Hin * g->Win * g->Ci * 2);
  a.bt_bytes = (uint32_t)((int64_t)g->Co * a.Kp * 2);
  const int twl = sc_tile_log(g);
  a.tiles_x = twl == 3 ? g->Win / 8 : 1;
  a.tiles_y = twl == 3 ? g->Hin / 8 : 1;
  a.ntiles = g->Co / 64;
  a.out_f32 = out_is_f32;
  a.slope_out = slope_out;
  a.dNt = make_fastdiv(a.ntiles);
  a.dTx = make_fastdiv(a.tiles_x);
  a.dTy = make_fastdiv(a.tiles_y);
  const int ptiles = twl == 3 ? g->N * a.tiles_y * a.tiles_x : cdiv(g->N, 4);
  const int grid = ptiles * a.ntiles;
  CgProfScope prof(CG_PROF_SCONV, g, st);
  const bool relu = gate_in != nullptr;
  if (twl == 3) {
    if (relu) sconv_kernel<3, true><<<grid, 320, 0, st>>>(a);
    else sconv_kernel<3, false><<<grid, 320, 0, st>>>(a);
  } else {
    if (relu) sconv_kernel<2, true><<<grid, 320, 0, st>>>(a);
    else sconv_kernel<2, false><<<grid, 320, 0, st>>>(a);
  }
}

// ---- small-map weight gradient ----
namespace {
int sw_log2(int x) {
  int l = 0;
  while ((1 << l) < x) ++l;
  return ((1 << l) == x) ? l : -1;
}
}  // namespace

bool cg_swgrad_geom_ok(const cgConvGeom* g) {
  if (g->S != 1 || g->U != 1 || g->kh != 3 || g->kw != 3 || g->pt != 1 || g->pl != 1) return false;
  if (g->Ho != g->Hin || g->Wo != g->Win) return false;
  if ((g->Ci % 64) != 0 || (g->Co % 64) != 0) return false;
  const int wl = sw_log2(g->Win), hl = sw_log2(g->Hin);
  if (wl < 2 || hl < 0 || g->Hin * g->Win < 16) return false;
  if (wl < 4 && (g->Hin << wl) % 16 != 0) return false;
  if ((int64_t)g->N * g->Hin * g->Win * (g->Ci > g->Co ? g->Ci : g->Co) * 2 >= (1ll << 31)) return false;
  return true;
}

// Policy (profiles/r03_small_conv_ab.txt): every workgroup walks ALL pixels, so the kernel is for
// small batches of pixels: up to 4096 on its own, up to 8192 inside a grouped launch (where the
// other layers of the group fill the CUs a 36-workgroup layer leaves idle).
// CGAMD_SWGRAD: 0 = off, 1 = policy (default), 2 = wherever the geometry fits.
bool cg_swgrad_supported(const cgConvGeom* g, const void* in, const void* gate_in, float slope_in,
                         const void* gate_dy, bool grouped) {
  static const int mode = sc_env("CGAMD_SWGRAD", 1);
  static const int max_pixels = sc_env("CGAMD_SWGRAD_MAX", 4096);
  static const int max_pixels_grouped = sc_env("CGAMD_SWGRAD_GROUP_MAX", 8192);
  if (!mode || gate_dy || !cg_swgrad_geom_ok(g)) return false;
  if (gate_in && !(gate_in == in && slope_in == 0.f)) return false;
  if (mode == 2) return true;
  return (int64_t)g->N * g->Hin * g->Win <= (grouped ? max_pixels_grouped : max_pixels);
}

// launches `n` (<= SW_MAX_JOBS) weight gradients as ONE grid
void cg_swgrad_launch_multi(const cgConvGeom* const* geoms, const void* const* ins,
                            const int* relus, const void* const* dys, float* const* dws,
                            const int* accumulates, float* const* dbs, int n, hipStream_t st) {
  SwArgs a;
  memset(&a, 0, sizeof(a));
  int wgs = 0;
  for (int i = 0; i < n; ++i) {
    const cgConvGeom* g = geoms[i];
    SwJob& j = a.job[i];
    j.x = (const bf16_t*)ins[i];
    j.dy = (const bf16_t*)dys[i];
    j.dw = dws[i];
    j.db = dbs[i];
    j.x_bytes = (uint32_t)((int64_t)g->N * g->Hin * g->Win * g->Ci * 2);
    j.dy_bytes = (uint32_t)((int64_t)g->N * g->Hin * g->Win * g->Co * 2);
    j.H = g->Hin; j.W = g->Win; j.wl = sw_log2(g->Win); j.hl = sw_log2(g->Hin);
    j.Ci = g->Ci; j.Co = g->Co;
    j.nks = g->N * g->Hin * g->Win / 16;
    j.relu = relus[i];
    j.accumulate = accumulates[i];
    j.tiles_ci = g->Ci / 64;
    j.tiles_co = g->Co / 64;
    j.wg_begin = wgs;
    wgs += 9 * j.tiles_ci * j.tiles_co;
  }
  a.njobs = n;
  swgrad_kernel<<<wgs, 256, 0, st>>>(a);
}

void cg_swgrad_launch(const cgConvGeom* g, const void* in, const void* gate_in, const void* dy,
                      float* dw, int accumulate, float* dbias, hipStream_t st) {
  const int relu = gate_in != nullptr;
  CgProfScope prof(CG_PROF_SWGRAD, g, st);
  const void* ins[1] = {in};
  const void* dys[1] = {dy};
  float* dws[1] = {dw};
  float* dbs[1] = {dbias};
  cg_swgrad_launch_multi(&g, ins, &relu, dys, dws, &accumulate, dbs, 1, st);
}
