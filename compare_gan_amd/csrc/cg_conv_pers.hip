// Persistent halo-staged 3x3 convolution for gfx950 ("pconv"): the deep-pipelined successor of
// hconv_kernel (cg_conv_halo.hip) for unit-stride 3x3 'SAME' filters on maps that tile into 16x32
// pixel tiles (32x32 ... 128x128 and larger).
// Contract and reference call sites: include/cgamd.h (cg_gconv / cg_gconv_fused: arch_ops.conv2d,
// arch_ops.py:559-573, the blocks of resnet5.py:99-145 / resnet_biggan.py:223-302 and their data
// gradients through the adjoint geometry); this file only adds a faster kernel behind the same
// entry points.
//
// Why (profiles/r02_hconv_workgroup_timeline.txt): hconv_kernel waits for `vmcnt(0)` and
// a workgroup barrier before every 64-channel x 1-tap K slice, stages the halo of the next channel
// block only after the current block's last MFMA, and pays descriptors + first-load latency +
// epilogue once per 256-pixel workgroup (about a fifth of a workgroup's life on the 128-channel
// layers).  Here
//  * ONE workgroup per CU lives for the whole launch and walks (tile, out-channel tile) items; the
//    input window with its halo (18 x 34 pixels x 64 channels = 76.5 KiB) is double-buffered in LDS,
//    so the window of the NEXT channel block -- or of the next item -- is in flight (LDS-DMA,
//    buffer_load ... lds, spread one 1-KiB piece per half K-slice) while the current one is
//    multiplied: no set-up, no first-load latency, no exposed halo latency after the first item;
//  * the weights do NOT go through LDS: a wave loads its own B fragments straight into registers
//    from a fragment-ordered image (cg_weight_frag_elems: every wave load is 1 KiB contiguous), three
//    half-slices deep with counted `vmcnt`, so the K loop has no barrier at all -- ONE s_barrier per
//    64-channel block (9 taps = 288 MFMAs per wave) hands the window buffers over;
//  * a wave owns 128 pixels x 64 out-channels (4 x 2 MFMA tiles of 32x32, 8 waves = 4 x 2), so one
//    ds_read_b128 of the window feeds two MFMAs: half the LDS bytes per MFMA of hconv_kernel's
//    64x64 wave tiles with LDS-staged weights;
//  * the per-wave LDS-staged epilogue (bias, gates, residual, pooling, batch-norm statistics: the
//    arithmetic of hconv_kernel, one rounding to bf16) runs in the window buffer that just died.
// LDS image and swizzle are hconv_kernel's: 128-byte rows (one pixel x 64 channels), 16-byte chunk c
// of a row at chunk c ^ ((halo_x >> 1) & 7).
#include "cg_conv_fast.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr uint32_t PC_OOB = 0x80000000u;   // voffset of a lane that must read zeros (bounds check)
constexpr int PC_TW = 32, PC_TH = 16, PC_PITCH = PC_TW + 2;
constexpr int PC_HROWS = (PC_TH + 2) * PC_PITCH;          // 612 window rows of 128 B
constexpr int PC_PIECES = (PC_HROWS + 7) / 8;             // 77 pieces of 1 KiB
constexpr int PC_HB = PC_PIECES * 1024;                   // one window buffer
constexpr int PC_SLOTS = (PC_PIECES + 7) / 8;             // 10 pieces per wave (waves 5..7: 9)
constexpr int PC_ROWSTEP = PC_PITCH * 128;                // LDS bytes between two tile rows

struct PConvArgs {
  const bf16_t* in;
  const bf16_t* btf;    // fragment-ordered weights [Co/32][cblocks][9][4][64 lanes][8]
  void* out;
  const float* bias;
  const bf16_t* gate_out;
  const bf16_t* residual;
  uint32_t btf_bytes;
  int N, H, W, Ci, Co;
  int cblocks, nslices, cotiles;   // 64-channel blocks, 9 * cblocks, ceil(Co / 32)
  int tiles_x, tiles_y, ntiles, nitems;
  int out_f32, self_gate;
  float slope_out;
  // fusions of cgConvFusion (cgamd.h), as in hconv_kernel
  const float* bn_mean;
  const float* bn_var;
  const float* bn_gamma;
  const float* bn_beta;
  float bn_eps;
  int bn_per_sample;
  int bn_stat_group;
  float* stats;
  int pool, in_up;
  float out_scale;
  FastDiv dNt, dTx, dTy;
#ifdef CG_CONV_TIMING
  unsigned long long* tdbg;   // 8 words per workgroup + 4 per wave (scripts/pconv_timeline.py)
#endif
  int prio;  // s_setprio level of the younger half of the waves (4..7), 0..3
  int dbg;   // CGAMD_PCONV_DBG, measurement only (results are WRONG when set): ablation bits 1 no
             // window DMA, 2 no weight wait, 4 no weight loads, 8 no output stores, 16 no window
             // fragment reads, 32 no MFMA

};
#ifdef CG_CONV_TIMING
#define PC_NOW() __builtin_amdgcn_s_memtime()
#define PC_TDBG(slot, val)                                              \
  do {                                                                  \
    if (a.tdbg && tid == 0) a.tdbg[blockIdx.x * 8 + (slot)] = (val);    \
  } while (0)
#else
#define PC_NOW() 0ull
#define PC_TDBG(slot, val) do {} while (0)
#endif
#if defined(CG_CONV_TIMING) || defined(CG_CONV_ABLATE)
#define PC_DBG(bit) (a.dbg & (bit))
#else
#define PC_DBG(bit) 0
#endif

__device__ __forceinline__ int pc_xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

__device__ __forceinline__ bf16x8_t pc_relu(bf16x8_t v) {
  s16x8_t s = __builtin_bit_cast(s16x8_t, v);
  const s16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
  s = __builtin_elementwise_max(s, z);
  return __builtin_bit_cast(bf16x8_t, s);
}

// ---- wave-private weight fragments: buffer_load_dwordx4 from inline asm, counted waits by hand ----
// (hipcc waits vmcnt(0) for any ordinary load that shares the queue with an LDS-DMA,
// cdna_hip_programming.md section 5; from asm the loads are invisible to that pass.)  The value is
// NOT valid until pc_wait<N> names the register again.
typedef cg_i32x4_t pc_frag_t;   // 8 bf16 of a weight fragment, as the 4 dwords the load delivers
template <int IMM>
__device__ __forceinline__ void pc_bload(pc_frag_t& dst, uint32_t voff, cg_i32x4_t rs, uint32_t soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4"
               : "=&v"(dst)
               : "v"(voff), "s"(rs), "s"(soff), "n"(IMM));
}
// at most N vector-memory operations of this wave still in flight; the fragments named here may be
// consumed after it (the "+v" ties keep hipcc from hoisting their MFMAs above the wait)
template <int N>
__device__ __forceinline__ void pc_wait(pc_frag_t& b0, pc_frag_t& b1) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b0), "+v"(b1) : "n"(N));
}
template <int N>
__device__ __forceinline__ void pc_wait(pc_frag_t& b0, pc_frag_t& b1, pc_frag_t& b2, pc_frag_t& b3) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "n"(N));
}
// the LDS-DMA of a window piece; no "memory" clobber on purpose: it writes the window buffer nobody
// reads before the next block-end barrier (which has one), and the clobber would stop hipcc from
// scheduling the fragment reads of the current buffer across it
__device__ __forceinline__ void pc_dma16(cg_i32x4_t rs, uint32_t voff, uint32_t soff, uint32_t lds) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 4\n\t"
      "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rs), "s"(lds), "s"(soff));
}

template <int I, int N, class F>
__device__ __forceinline__ void pc_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    pc_static_for<I + 1, N>(f);
  }
}

// one (item, channel block) of a workgroup's walk: everything wave-uniform
struct PcBlock {
  int q, cb;          // item, 64-channel block
  int valid;          // 0: past the end (its loads are issued as all-zero reads: the counts stay exact)
  int n, ty, tx, nt;  // image, tile row / column, out-channel tile
  int st;             // spatial tile index (statistics row)
};

// ---- the epilogue of one (tile, out-channel tile) item, shared by the persistent and the
// two-workgroups-per-CU kernel: every wave stages its own MI x 32 pixels x WCO channels accumulator
// tile through a private LDS region (fp32, one tile row of 32 pixels per pass), then each lane
// finishes 8 consecutive channels of one pixel (bias, activation, gate, residual, pooling: one
// rounding to bf16) and writes 16 (bf16) / 32 (fp32) contiguous bytes; batch-norm statistics of the
// STORED values per item in a fixed order (hconv_kernel's arithmetic).  NW waves = WM x WN; tile rows
// wm * MI + i; TH = tile height.  Ends with the waves' LDS traffic complete but NO barrier.
template <int BN, int WM, int WN, int MI, int NJ, int FUSE, int TH, int STAGE_BYTES>
__device__ __forceinline__ void pc_epilogue(const PConvArgs& a, f32x16_t (&acc)[MI][NJ],
                                            unsigned char* stage, int tid, int lane, int wave, int wm,
                                            int wn, int frow, int half, int n, int ty, int tx, int nt,
                                            int st) {
  constexpr int NW = WM * WN;
    constexpr int WCO = BN / WN;           // channels per wave
    constexpr int SP = WCO * 4 + 16;       // staging row pitch in bytes (+16: conflict-free b128 writes)
    constexpr int G8 = WCO / 8;            // 8-channel groups per row
    constexpr int RPI = 64 / G8;           // rows per sweep of the 64 lanes
    static_assert(NW * 32 * SP + NW * 2 * WCO * 4 <= STAGE_BYTES, "per-wave epilogue staging does not fit");
    unsigned char* Sw = stage + wave * (32 * SP);
    const int n0 = nt * BN;
    const int g8 = lane & (G8 - 1), rl = lane / G8;
    const int co = n0 + wn * WCO + g8 * 8;
    const bool co_ok = co < a.Co;          // Co % 8 == 0
    const float osc = a.out_scale;
    if constexpr (FUSE == 2) {
      // ---- pooled epilogue: tile rows wm*MI + 2p, + 2p + 1 are acc[2p] / acc[2p + 1] of the SAME
      // lane (summed in registers); horizontal pairs are neighbouring staging rows
#pragma unroll
      for (int p = 0; p < MI / 2; ++p) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            float4 t = make_float4(acc[2 * p][j][q4 * 4 + 0] + acc[2 * p + 1][j][q4 * 4 + 0],
                                   acc[2 * p][j][q4 * 4 + 1] + acc[2 * p + 1][j][q4 * 4 + 1],
                                   acc[2 * p][j][q4 * 4 + 2] + acc[2 * p + 1][j][q4 * 4 + 2],
                                   acc[2 * p][j][q4 * 4 + 3] + acc[2 * p + 1][j][q4 * 4 + 3]);
            *reinterpret_cast<float4*>(Sw + frow * SP + (j * 32 + q4 * 8 + 4 * half) * 4) = t;
          }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = lane; it < 16 * G8; it += 64) {
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
          const int cq = n0 + wn * WCO + gg * 8;
          if (cq >= a.Co) continue;
          const int oy = ty * (TH / 2) + wm * (MI / 2) + p, ox = tx * (PC_TW / 2) + x2;
          const int64_t o = ((int64_t)(n * (a.H >> 1) + oy) * (a.W >> 1) + ox) * a.Co + cq;
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
    } else {
      float bv[8], s1[8], s2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] = s1[e] = s2[e] = 0.f;
      if (a.bias && co_ok) {
        const float4 b0 = *reinterpret_cast<const float4*>(a.bias + co);
        const float4 b1 = *reinterpret_cast<const float4*>(a.bias + co + 4);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
        bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4)
            *reinterpret_cast<float4*>(Sw + frow * SP + (j * 32 + q4 * 8 + 4 * half) * 4) =
                make_float4(acc[i][j][q4 * 4 + 0], acc[i][j][q4 * 4 + 1], acc[i][j][q4 * 4 + 2],
                            acc[i][j][q4 * 4 + 3]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 32 / RPI; ++k) {
          const int row = rl + RPI * k;   // pixel column of tile row wm*MI + i
          const float4 lo = *reinterpret_cast<const float4*>(Sw + row * SP + g8 * 32);
          const float4 hi = *reinterpret_cast<const float4*>(Sw + row * SP + g8 * 32 + 16);
          if (!co_ok) continue;
          const int oy = ty * TH + wm * MI + i, ox = tx * PC_TW + row;
          const int64_t o = ((int64_t)(n * a.H + oy) * a.W + ox) * a.Co + co;
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
          if (PC_DBG(8)) continue;
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
        // lanes with the same g8 hold partial sums of the same 8 channels: butterfly over the
        // others, then the WM pixel-waves of a channel group are combined through LDS in a fixed
        // order (deterministic)
#pragma unroll
        for (int m = G8; m < 64; m <<= 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[e] += __shfl_xor(s1[e], m, 64);
            s2[e] += __shfl_xor(s2[e], m, 64);
          }
        }
        float* sreg = reinterpret_cast<float*>(stage + NW * 32 * SP);   // [8 waves][2][WCO]
        if (lane < G8) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            sreg[(wave * 2 + 0) * WCO + g8 * 8 + e] = s1[e];
            sreg[(wave * 2 + 1) * WCO + g8 * 8 + e] = s2[e];
          }
        }
        __syncthreads();
        if (tid < BN) {
          const int cw = tid / WCO, cc = tid - cw * WCO;
          float t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int m4 = 0; m4 < WM; ++m4) {
            t1 += sreg[((m4 * WN + cw) * 2 + 0) * WCO + cc];
            t2 += sreg[((m4 * WN + cw) * 2 + 1) * WCO + cc];
          }
          const int cch = n0 + tid;
          if (cch < a.Co) {
            a.stats[(int64_t)st * 2 * a.Co + cch] = t1;
            a.stats[(int64_t)st * 2 * a.Co + a.Co + cch] = t2;
          }
        }
      }
    }
}

// BN: out-channels per workgroup (128 / 64); WM: waves along the pixels (4: 128 x (BN/2) per wave,
// 2: 256 x (BN/4)); FUSE: 0 plain, 1 batch-norm prologue / statistics epilogue, 2 pooled epilogue
template <int BN, int WM, bool RELU, int FUSE, bool PIPE>
__global__ __launch_bounds__(512, 2) void pconv_kernel(PConvArgs a) {
  constexpr int WN = 8 / WM;
  constexpr int MI = PC_TH / WM;          // tile rows (= 32-pixel MFMA tiles) per wave
  constexpr int NJ = BN / (32 * WN);      // 32-channel MFMA tiles per wave
  constexpr int LB = 2 * NJ;              // fragment loads per half-slice (2 k-steps of 16)
  static_assert(NJ == 1 || NJ == 2, "wave tile");
  constexpr int TAB_OFF = 2 * PC_HB;      // [4][64] floats: mean, rstd, gamma, beta of a channel block
  constexpr int LDS_BYTES = 2 * PC_HB + (FUSE == 1 ? 1024 : 0);
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & 31, half = lane >> 5;
  const int G = gridDim.x;
  const int wgp = pc_xcd_remap(blockIdx.x, G);
  const uint32_t lds0 = cg_lds_addr(smem);
  const int us = a.in_up;   // 0 / 1: logical pixel (iy, ix) lives at (iy >> us, ix >> us)
  PC_TDBG(0, PC_NOW());
  PC_TDBG(5, __builtin_amdgcn_s_memrealtime());
  unsigned long long t_blocks = 0, t_epi = 0, n_blocks = 0, t_wait = 0, t_bar = 0;
  (void)t_blocks; (void)t_epi; (void)n_blocks; (void)t_wait; (void)t_bar;

  const cg_i32x4_t rs_b = cg_make_rsrc(a.btf, a.btf_bytes);

  auto make_block = [&](int q, int cb) {
    PcBlock b;
    b.q = q;
    b.cb = cb;
    b.valid = q < a.nitems;
    const int qq = b.valid ? q : 0;
    b.st = (int)fdiv((uint32_t)qq, a.dNt);
    b.nt = qq - b.st * a.ntiles;
    const int t1 = (int)fdiv((uint32_t)b.st, a.dTx);
    b.tx = b.st - t1 * a.tiles_x;
    b.n = (int)fdiv((uint32_t)t1, a.dTy);
    b.ty = t1 - b.n * a.tiles_y;
    b.q = __builtin_amdgcn_readfirstlane(b.q);
    b.st = __builtin_amdgcn_readfirstlane(b.st);
    b.nt = __builtin_amdgcn_readfirstlane(b.nt);
    b.tx = __builtin_amdgcn_readfirstlane(b.tx);
    b.ty = __builtin_amdgcn_readfirstlane(b.ty);
    b.n = __builtin_amdgcn_readfirstlane(b.n);
    return b;
  };
  auto next_block = [&](const PcBlock& b) {
    return b.cb + 1 < a.cblocks ? make_block(b.q, b.cb + 1) : make_block(b.q + G, 0);
  };

  // ---- window staging: piece p = wave + 8 j covers window rows 8 p .. 8 p + 7; lane -> row
  // 8 p + (lane >> 3), LDS chunk (lane & 7), which must hold source chunk (lane & 7) ^ ((hx >> 1) & 7).
  // The per-lane geometry is recomputed per piece (a dozen VALU operations per 288 MFMAs) instead of
  // living in 20 registers.  The descriptor base is the window origin of the tile, so offsets are
  // tile-independent and non-negative; padding = the bounds check (offset >= 0x80000000 -> zeros).
  auto window_rsrc = [&](const PcBlock& b) {
    const int py = ((b.ty * PC_TH) >> us) - 1, px = ((b.tx * PC_TW) >> us) - 1;
    const bf16_t* o = a.in + (((int64_t)b.n * (a.H >> us) + py) * (a.W >> us) + px) * a.Ci;
    return cg_make_rsrc(o, 0x7fffffffu);
  };
  // Per piece ONE register, computed once: the byte offset of this lane's 16-byte chunk relative to
  // the window origin (a multiple of 16, < 2^30), with the conditions that can make it padding in
  // its spare bits -- bit 0 / 1: first / last window row, bit 2 / 3: first / last window column
  // (outside the image when the tile touches that edge), bit 30: upper half of the 64 channels
  // (zero-filled in a ragged last channel block) -- or PC_OOB for the rows past the window.
  uint32_t prel[PC_SLOTS];
#pragma unroll
  for (int j = 0; j < PC_SLOTS; ++j) {
    const int row = (wave + 8 * j) * 8 + (lane >> 3);
    const int hy = row / PC_PITCH, hx = row - hy * PC_PITCH;
    const int c8 = ((lane & 7) ^ ((hx >> 1) & 7)) * 8;
    const uint32_t rel =
        (uint32_t)(((((hy + us) >> us) * (a.W >> us) + ((hx + us) >> us)) * a.Ci + c8) * 2);
    prel[j] = row < PC_HROWS ? (rel | (hy == 0 ? 1u : 0u) | (hy == PC_TH + 1 ? 2u : 0u) |
                                (hx == 0 ? 4u : 0u) | (hx == PC_TW + 1 ? 8u : 0u) |
                                (c8 >= 32 ? 0x40000000u : 0u))
                             : PC_OOB;
  }
  // the bits of prel that mean "padding" for a block: its tile's image edges, a ragged channel block,
  // everything when the block is past the end
  auto pad_mask = [&](const PcBlock& b) {
    uint32_t m = 0x80000000u | (b.ty == 0 ? 1u : 0u) | (b.ty == a.tiles_y - 1 ? 2u : 0u) | (b.tx == 0 ? 4u : 0u) |
                 (b.tx == a.tiles_x - 1 ? 8u : 0u) | (a.Ci - b.cb * 64 < 64 ? 0x40000000u : 0u);
    return b.valid ? m : 0xffffffffu;
  };
  auto halo_piece = [&](auto jc, uint32_t pmask, int cb, const cg_i32x4_t& rs, int buf) {
    constexpr int j = decltype(jc)::value;
    const int piece = wave + 8 * j;
    if (piece < PC_PIECES) {   // wave-uniform (false only for j = 9 on waves 5..7)
      const uint32_t pr = prel[j];
      const uint32_t vo = ((pr & pmask) != 0u || pmask == 0xffffffffu) ? PC_OOB : (pr & 0x3ffffff0u);
      if (!PC_DBG(1)) pc_dma16(rs, vo, (uint32_t)(cb * 128), lds0 + buf * PC_HB + piece * 1024);
    }
  };
  // ReLU of the input (arch_ops.py:595-597 / resnet_ops.py:165 in front of the convolution): applied
  // ONCE to the staged window, in LDS, by the wave that staged the piece, right after its own counted
  // wait has seen the piece land -- not on each of the 9 x 2 fragment reads that consume a chunk
  // (that was 2 VALU instructions per MFMA in a loop whose ceiling is instruction issue).  bf16 ReLU
  // = signed 16-bit max with 0; padding zeros stay zeros.
  auto relu_piece = [&](auto jc, int buf) {
    constexpr int j = decltype(jc)::value;
    const int piece = wave + 8 * j;
    if (piece < PC_PIECES) {
      bf16x8_t* p = reinterpret_cast<bf16x8_t*>(smem + buf * PC_HB + piece * 1024 + lane * 16);
      *p = pc_relu(*p);
    }
  };

  // ---- weight fragments of half-slice (slice s = cb * 9 + tap, k-steps 2 hh, 2 hh + 1) ----
  const uint32_t lane16 = (uint32_t)lane * 16u;
  // load `idx` (= j * 2 + k2) of a half-slice on its own: the main loop spreads the LB loads of a
  // half-slice over its row pairs (a burst of 4 queues up in front of the address unit)
  auto b_issue_one = [&](pc_frag_t (&dst)[NJ][2], const PcBlock& b, int tap, auto hhc, auto idxc) {
    constexpr int hh = decltype(hhc)::value, idx = decltype(idxc)::value;
    constexpr int j = idx >> 1, k2 = idx & 1;
    const int ct = b.nt * (BN / 32) + wn * NJ + j;
    const bool ok = b.valid && ct < a.cotiles;
    const uint32_t soff = (uint32_t)(((ok ? ct : 0) * a.nslices + b.cb * 9 + tap) * 4096);
    const uint32_t vo = ok ? lane16 : PC_OOB;
    if (!PC_DBG(4)) pc_bload<hh * 2048 + k2 * 1024>(dst[j][k2], vo, rs_b, soff);
  };
  auto b_issue = [&](pc_frag_t (&dst)[NJ][2], const PcBlock& b, int tap, auto hhc) {
    constexpr int hh = decltype(hhc)::value;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int ct = b.nt * (BN / 32) + wn * NJ + j;
      const bool ok = b.valid && ct < a.cotiles;
      const uint32_t soff = (uint32_t)(((ok ? ct : 0) * a.nslices + b.cb * 9 + tap) * 4096);
      const uint32_t vo = ok ? lane16 : PC_OOB;
      if (!PC_DBG(4)) {
        pc_bload<hh * 2048>(dst[j][0], vo, rs_b, soff);
        pc_bload<hh * 2048 + 1024>(dst[j][1], vo, rs_b, soff);
      }
    }
  };

  // ---- fused batch-norm prologue (hconv_kernel's, on the 18 x 34 window) ----
  const bool bnp = FUSE == 1 && a.bn_mean != nullptr;   // wave-uniform
  auto load_bn_table = [&](const PcBlock& b) {
    if (tid < 64) {
      float* tab = reinterpret_cast<float*>(smem + TAB_OFF);
      const int ch = min(b.cb * 64 + tid, a.Ci - 1);   // (a ragged last block only uses its first half)
      const int64_t pidx = a.bn_per_sample ? (int64_t)b.n * a.Ci + ch : ch;
      const int64_t sidx = a.bn_stat_group > 0 ? (int64_t)(b.n / a.bn_stat_group) * a.Ci + ch : ch;
      tab[tid] = a.bn_mean[sidx];
      tab[64 + tid] = rsqrtf(a.bn_var[sidx] + a.bn_eps);
      tab[128 + tid] = a.bn_gamma ? a.bn_gamma[pidx] : 1.f;
      tab[192 + tid] = a.bn_beta ? a.bn_beta[pidx] : 0.f;
    }
  };
  // the staged window is normalised in place (operation order of cg_bn_apply, arch_ops.py:306-312);
  // padding pixels stay zero: the padding applies to the BN output.  A thread keeps ONE 8-channel
  // group (tid & 7) for all its rows: 32 coefficients out of the LDS table once per block.
  auto bn_transform = [&](const PcBlock& b, int buf) {
    const float* tab = reinterpret_cast<const float*>(smem + TAB_OFF);
    const int crem = a.Ci - b.cb * 64;
    const int c8 = (tid & 7) * 8;
    if (c8 >= crem) return;   // zero-filled half of a ragged last channel block
    float cm[8], cr[8], cg[8], cbt[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      cm[e] = tab[c8 + e];
      cr[e] = tab[64 + c8 + e];
      cg[e] = tab[128 + c8 + e];
      cbt[e] = tab[192 + c8 + e];
    }
    unsigned char* base = smem + buf * PC_HB;
    for (int row = tid >> 3; row < PC_HROWS; row += 64) {
      const int hy = row / PC_PITCH, hx = row - hy * PC_PITCH;
      const int iy = b.ty * PC_TH - 1 + hy, ix = b.tx * PC_TW - 1 + hx;
      if (!((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)) continue;
      uint4* p = reinterpret_cast<uint4*>(base + (row * 8 + ((tid & 7) ^ ((hx >> 1) & 7))) * 16);
      float v[8];
      unpack8_bf16(*p, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = (v[e] - cm[e]) * cr[e];
        t = t * cg[e] + cbt[e];
        v[e] = fmaxf(t, 0.f);
      }
      *p = pack8_bf16(v);
    }
  };

  // ---- fragment addressing.  Pixel (tile row wm*MI + i, column frow): window row (y + r) * PITCH +
  // frow + s for tap (r, s); chunk (kk*2 + half) of it sits at slot ^ (((frow + s) >> 1) & 7).  Per
  // tap column s and k-step one per-lane register; tile row / tap row are immediates.
  // chunk slot of k-step 0 for tap column s, in bytes (k-step kk: ^ (kk << 5)); row base of s = 0
  int tsw[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) tsw[s] = (half ^ (((frow + s) >> 1) & 7)) << 4;
  const int rowbase = ((wm * MI) * PC_PITCH + frow) * 128;

  f32x16_t acc[MI][NJ];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
  };
  zero_acc();

  pc_frag_t Bq[3][NJ][2];   // ring of three half-slices of this wave's weight fragments

  // ---- prologue: the first window and the first two half-slices of weights ----
  PcBlock cur = make_block(wgp, 0);
  PcBlock nxt = next_block(cur);
  {
    const cg_i32x4_t rs0 = window_rsrc(cur);
    const uint32_t pm0 = pad_mask(cur);
    pc_static_for<0, PC_SLOTS>([&](auto jc) { halo_piece(jc, pm0, cur.cb, rs0, 0); });
    b_issue(Bq[0], cur, 0, std::integral_constant<int, 0>());
    b_issue(Bq[1], cur, 0, std::integral_constant<int, 1>());
    if (bnp) load_bn_table(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (RELU) pc_static_for<0, PC_SLOTS>([&](auto jc) { relu_piece(jc, 0); });   // own pieces: landed
    __syncthreads();
    if (bnp) {
      bn_transform(cur, 0);
      __syncthreads();
    }
  }
  PC_TDBG(1, PC_NOW());
  // the second-dispatched half of the waves loses every issue arbitration to the older half on its
  // SIMD (MI355X_MICROARCH.md, two waves per SIMD) and sets the pace of every block: a static
  // priority for it
  if (wave >= 4) {
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
  }
  int buf = 0;
  bool fresh = true;   // the weights of half-slices 0 and 1 have landed (prologue / after an epilogue)

  while (cur.valid) {
    const cg_i32x4_t rs_n = window_rsrc(nxt);
    const uint32_t pm_n = pad_mask(nxt);
    const int cb_n = nxt.cb;
    const int rb = rowbase + buf * PC_HB;

    // ---- 9 taps x 2 half-slices: no barrier, no drain ----
    const unsigned long long tb0 = PC_NOW();
    // window fragments of k-step kk of half-slice h for tile rows i0, i0 + 1
    auto a_read = [&](auto hc, auto k2c, auto i0c, bf16x8_t (&dst)[2]) {
      constexpr int h = decltype(hc)::value, k2 = decltype(k2c)::value, i0 = decltype(i0c)::value;
      constexpr int tap = h >> 1, hh = h & 1, r = tap / 3, s = tap % 3, kk = hh * 2 + k2;
      const int ad = rb + (tsw[s] ^ (kk << 5));
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (PC_DBG(16)) continue;
        dst[e] = *reinterpret_cast<const bf16x8_t*>(smem + ad + s * 128 + (i0 + e + r) * PC_ROWSTEP);
      }
    };
    bf16x8_t afp[2][2];   // PIPE: two fragment pairs in flight (read one pair ahead of the MFMAs)
    if constexpr (PIPE)
      a_read(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(),
             std::integral_constant<int, 0>(), afp[0]);
    pc_static_for<0, 18>([&](auto hc) {
      constexpr int h = decltype(hc)::value;
      constexpr int tap = h >> 1, hh = h & 1, r = tap / 3, s = tap % 3;
      constexpr int h2 = h + 2;
      // 1. weights two half-slices ahead (of this block, or of the next one): all at once here, or
      // (PIPE) one load per row pair below
      auto b_ahead = [&](auto idxc) {
        if constexpr (h2 < 18)
          b_issue_one(Bq[h2 % 3], cur, h2 >> 1, std::integral_constant<int, h2 & 1>(), idxc);
        else
          b_issue_one(Bq[h2 % 3], nxt, (h2 - 18) >> 1, std::integral_constant<int, h2 & 1>(), idxc);
      };
      if constexpr (!PIPE) pc_static_for<0, LB>(b_ahead);
      // 2. this half-slice's weights have landed: everything younger may stay in flight.  !PIPE: the
      // two half-slices just issued and the window pieces issued behind them (one per half-slice
      // 0..8; piece 9 is not issued by every wave and is not counted: a stricter wait for the
      // others).  PIPE: the last load of this half-slice left in the last row pair of half-slice
      // h - 2; behind it came the piece and the loads of half-slice h - 1
      constexpr int allow = PIPE ? LB + ((h - 1 >= 0 && h - 1 <= 8) ? 1 : 0)
                                 : 2 * LB + ((h - 2 >= 0 && h - 2 <= 8) ? 1 : 0) +
                                       ((h - 1 >= 0 && h - 1 <= 8) ? 1 : 0);
#ifdef CG_CONV_TIMING
      const unsigned long long tw0 = PC_NOW();
#endif
      if (!(fresh && h < 2) && !PC_DBG(2 | 4)) {
        if constexpr (NJ == 2)
          pc_wait<allow>(Bq[h % 3][0][0], Bq[h % 3][0][1], Bq[h % 3][1][0], Bq[h % 3][1][1]);
        else
          pc_wait<allow>(Bq[h % 3][0][0], Bq[h % 3][0][1]);
      }
#ifdef CG_CONV_TIMING
      t_wait += PC_NOW() - tw0;
#endif
      // 3. one piece of the next block's window
      if constexpr (h < PC_SLOTS) halo_piece(hc, pm_n, cb_n, rs_n, buf ^ 1);
      // 3b. ReLU of the next window: every piece of this wave is older than the weights the wait
      // above covered (issued at half-slices >= 10), so they have landed; two pieces per half-slice
      if constexpr (RELU && h >= 12 && h < 12 + PC_SLOTS / 2) {
        asm volatile("" ::: "memory");   // the LDS reads below stay behind the wait
        relu_piece(std::integral_constant<int, 2 * (h - 12)>(), buf ^ 1);
        relu_piece(std::integral_constant<int, 2 * (h - 12) + 1>(), buf ^ 1);
      }
      // 4. two k-steps
      if constexpr (PIPE) {
        // pairs of tile rows: (k-step, row pair) = MI pairs per half-slice; the fragments of pair
        // g + 1 are read before the MFMAs of pair g are issued (MI is even: pair 0 of every
        // half-slice uses set 0)
        constexpr int NP = MI;   // 2 k-steps x MI / 2 row pairs
        pc_static_for<0, NP>([&](auto gc) {
          constexpr int g = decltype(gc)::value;
          constexpr int k2 = g / (MI / 2), i0 = (g % (MI / 2)) * 2;
          constexpr int gn = g + 1;
          if constexpr (g % (NP / LB) == 0) b_ahead(std::integral_constant<int, g / (NP / LB)>());
          if constexpr (gn < NP)
            a_read(hc, std::integral_constant<int, gn / (MI / 2)>(),
                   std::integral_constant<int, (gn % (MI / 2)) * 2>(), afp[gn & 1]);
          else if constexpr (h + 1 < 18)
            a_read(std::integral_constant<int, h + 1>(), std::integral_constant<int, 0>(),
                   std::integral_constant<int, 0>(), afp[0]);
          if (!PC_DBG(32)) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
              for (int j = 0; j < NJ; ++j)
                acc[i0 + e][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(bf16x8_t, Bq[h % 3][j][k2]), afp[g & 1][e], acc[i0 + e][j], 0, 0, 0);
          }
        });
      } else {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const int kk = hh * 2 + k2;
          const int ad = rb + (tsw[s] ^ (kk << 5));
          bf16x8_t af[MI];
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            af[i] = *reinterpret_cast<const bf16x8_t*>(smem + ad + s * 128 + (i + r) * PC_ROWSTEP);
          }
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                  __builtin_bit_cast(bf16x8_t, Bq[h % 3][j][k2]), af[i], acc[i][j], 0, 0, 0);
        }
      }
    });
#ifdef CG_CONV_TIMING
    const unsigned long long tb1 = PC_NOW();
    t_blocks += tb1 - tb0;
    ++n_blocks;
#endif
    fresh = false;

    // every wave is done with window `buf`, and its pieces of the next window have landed (they are
    // older than the weights it waited for in half-slices 12..17): after the barrier so have
    // everybody's
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (the ReLU pass's ds_writes)
#ifdef CG_CONV_TIMING
    t_bar += PC_NOW() - tb1;
#endif

    if (cur.cb == a.cblocks - 1) {
      // ================= epilogue of the item, in the dead window buffer =================
      // the next block's first weights land before the stores below queue up behind them
      // (the ring registers are named so that none of them is handed to the epilogue while a load
      // into it is still in flight: see qconv_kernel)
#pragma unroll
      for (int q3 = 0; q3 < 3; ++q3)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(Bq[q3][j][0]), "+v"(Bq[q3][j][1])::"memory");
      fresh = true;
      const unsigned long long te0 = PC_NOW();
      pc_epilogue<BN, WM, WN, MI, NJ, FUSE, PC_TH, PC_HB>(a, acc, smem + buf * PC_HB, tid, lane, wave, wm, wn,
                                                          frow, half, cur.n, cur.ty, cur.tx, cur.nt, cur.st);
      zero_acc();
      // the staging rows are dead before any wave stages a piece of the window after next into them
      __syncthreads();
#ifdef CG_CONV_TIMING
      t_epi += PC_NOW() - te0;
#endif
    }

    cur = nxt;
    nxt = next_block(cur);
    buf ^= 1;
    if (bnp && cur.valid) {
      load_bn_table(cur);
      __syncthreads();
      bn_transform(cur, buf);
      __syncthreads();
    }
  }
  // nothing of this wave may still be in flight towards LDS when the workgroup's LDS is released
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PC_TDBG(2, t_blocks);
  PC_TDBG(3, t_epi);
  PC_TDBG(7, n_blocks);
  PC_TDBG(4, PC_NOW());
  PC_TDBG(6, __builtin_amdgcn_s_memrealtime());
#ifdef CG_CONV_TIMING
  if (a.tdbg && lane == 0) {
    unsigned long long* w = a.tdbg + 512 * 8 + ((size_t)blockIdx.x * 8 + wave) * 4;
    w[0] = t_blocks; w[1] = t_wait; w[2] = t_bar; w[3] = t_epi;
  }
#endif
}


// -------------------------------------------------------------------------------------------
// The same K loop in hconv_kernel's occupancy model ("qconv"): TWO independent 4-wave workgroups
// per CU (one wave per SIMD each, 256 VGPRs), each owning one 8 x 32 pixel tile x BN out-channels
// with ONE window buffer.  What the persistent kernel above taught (profiles/r04_pconv_*.txt,
// scripts/probe/mfma_probe.hip): its K loop alone reaches 1.4-1.7 PFLOP/s, but every persistent
// workgroup of the chip arrives at its epilogue in the same microsecond -- 33 MB of stores at the
// HBM write rate, 9 k cycles per item during which no MFMA issues -- and the first window of a
// launch is pure latency.  Two unsynchronised workgroups per CU hide exactly those phases behind
// each other's MFMA work (what carries hconv_kernel), while the loop keeps what made the
// persistent one lean: weights straight from the fragment-ordered image into registers (no
// per-slice barrier: ONE barrier pair per 64-channel block), 128 x 64 wave tiles, window fragments
// read one row pair ahead, ReLU applied once to the staged window.
// -------------------------------------------------------------------------------------------
constexpr int QC_TH = 8;
constexpr int QC_HROWS = (QC_TH + 2) * PC_PITCH;          // 340 window rows
constexpr int QC_PIECES = (QC_HROWS + 7) / 8;             // 43 pieces of 1 KiB
constexpr int QC_HB = QC_PIECES * 1024;
constexpr int QC_SLOTS = (QC_PIECES + 3) / 4;             // 11 pieces per wave (wave 3: 10)

template <int BN, bool RELU, int FUSE>
__global__ __launch_bounds__(256, 2) void qconv_kernel(PConvArgs a) {
  constexpr int WM = 2, WN = 2, MI = 4;
  constexpr int NJ = BN / (32 * WN);
  constexpr int LB = 2 * NJ;
  constexpr int TAB_OFF = QC_HB;
  constexpr int LDS_BYTES = QC_HB + (FUSE == 1 ? 1024 : 0);
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & 31, half = lane >> 5;
  const uint32_t lds0 = cg_lds_addr(smem);
  const int us = a.in_up;

  // ---- workgroup -> (image, tile row, tile column, out-channel tile), XCD-contiguous ----
  const int q = pc_xcd_remap(blockIdx.x, gridDim.x);
  const int st = (int)fdiv((uint32_t)q, a.dNt);
  const int nt = q - st * a.ntiles;
  const int t1 = (int)fdiv((uint32_t)st, a.dTx);
  const int tx = st - t1 * a.tiles_x;
  const int n = (int)fdiv((uint32_t)t1, a.dTy);
  const int ty = t1 - n * a.tiles_y;

  const cg_i32x4_t rs_b = cg_make_rsrc(a.btf, a.btf_bytes);
  cg_i32x4_t rs_w;
  {
    const int py = ((ty * QC_TH) >> us) - 1, px = ((tx * PC_TW) >> us) - 1;
    rs_w = cg_make_rsrc(a.in + (((int64_t)n * (a.H >> us) + py) * (a.W >> us) + px) * a.Ci, 0x7fffffffu);
  }
  // per piece one register (see pconv_kernel): offset | padding-condition bits
  uint32_t prel[QC_SLOTS];
#pragma unroll
  for (int j = 0; j < QC_SLOTS; ++j) {
    const int row = (wave + 4 * j) * 8 + (lane >> 3);
    const int hy = row / PC_PITCH, hx = row - hy * PC_PITCH;
    const int c8 = ((lane & 7) ^ ((hx >> 1) & 7)) * 8;
    const uint32_t rel =
        (uint32_t)(((((hy + us) >> us) * (a.W >> us) + ((hx + us) >> us)) * a.Ci + c8) * 2);
    prel[j] = row < QC_HROWS ? (rel | (hy == 0 ? 1u : 0u) | (hy == QC_TH + 1 ? 2u : 0u) |
                                (hx == 0 ? 4u : 0u) | (hx == PC_TW + 1 ? 8u : 0u) |
                                (c8 >= 32 ? 0x40000000u : 0u))
                             : PC_OOB;
  }
  const uint32_t edge = 0x80000000u | (ty == 0 ? 1u : 0u) | (ty == a.tiles_y - 1 ? 2u : 0u) |
                        (tx == 0 ? 4u : 0u) | (tx == a.tiles_x - 1 ? 8u : 0u);
  auto window_issue = [&](int cb) {
    const uint32_t pmask = edge | (a.Ci - cb * 64 < 64 ? 0x40000000u : 0u);
    pc_static_for<0, QC_SLOTS>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const int piece = wave + 4 * j;
      if (piece < QC_PIECES) {
        const uint32_t pr = prel[j];
        pc_dma16(rs_w, (pr & pmask) != 0u ? PC_OOB : (pr & 0x3ffffff0u), (uint32_t)(cb * 128),
                 lds0 + piece * 1024);
      }
    });
  };
  auto relu_own_pieces = [&]() {
    pc_static_for<0, QC_SLOTS>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const int piece = wave + 4 * j;
      if (piece < QC_PIECES) {
        bf16x8_t* p = reinterpret_cast<bf16x8_t*>(smem + piece * 1024 + lane * 16);
        *p = pc_relu(*p);
      }
    });
  };

  const uint32_t lane16 = (uint32_t)lane * 16u;
  // weight fragment `idx` (= j * 2 + k2) of half-slice (cb, tap, hh); past the last block: zeros
  auto b_issue_one = [&](pc_frag_t (&dst)[NJ][2], int cb, int tap, auto hhc, auto idxc) {
    constexpr int hh = decltype(hhc)::value, idx = decltype(idxc)::value;
    constexpr int j = idx >> 1, k2 = idx & 1;
    const int ct = nt * (BN / 32) + wn * NJ + j;
    const bool ok = cb < a.cblocks && ct < a.cotiles;
    const uint32_t soff = (uint32_t)(((ok ? ct : 0) * a.nslices + (ok ? cb : 0) * 9 + tap) * 4096);
    pc_bload<hh * 2048 + k2 * 1024>(dst[j][k2], ok ? lane16 : PC_OOB, rs_b, soff);
  };

  // fused batch-norm prologue (hconv_kernel's, on the 10 x 34 window)
  const bool bnp = FUSE == 1 && a.bn_mean != nullptr;
  auto load_bn_table = [&](int cb) {
    if (tid < 64) {
      float* tab = reinterpret_cast<float*>(smem + TAB_OFF);
      const int ch = min(cb * 64 + tid, a.Ci - 1);
      const int64_t pidx = a.bn_per_sample ? (int64_t)n * a.Ci + ch : ch;
      const int64_t sidx = a.bn_stat_group > 0 ? (int64_t)(n / a.bn_stat_group) * a.Ci + ch : ch;
      tab[tid] = a.bn_mean[sidx];
      tab[64 + tid] = rsqrtf(a.bn_var[sidx] + a.bn_eps);
      tab[128 + tid] = a.bn_gamma ? a.bn_gamma[pidx] : 1.f;
      tab[192 + tid] = a.bn_beta ? a.bn_beta[pidx] : 0.f;
    }
  };
  auto bn_transform = [&](int cb) {
    const float* tab = reinterpret_cast<const float*>(smem + TAB_OFF);
    const int crem = a.Ci - cb * 64;
    const int c8 = (tid & 7) * 8;
    if (c8 >= crem) return;
    float cm[8], cr[8], cg[8], cbt[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      cm[e] = tab[c8 + e];
      cr[e] = tab[64 + c8 + e];
      cg[e] = tab[128 + c8 + e];
      cbt[e] = tab[192 + c8 + e];
    }
    for (int row = tid >> 3; row < QC_HROWS; row += 32) {
      const int hy = row / PC_PITCH, hx = row - hy * PC_PITCH;
      const int iy = ty * QC_TH - 1 + hy, ix = tx * PC_TW - 1 + hx;
      if (!((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)) continue;
      uint4* p = reinterpret_cast<uint4*>(smem + (row * 8 + ((tid & 7) ^ ((hx >> 1) & 7))) * 16);
      float v[8];
      unpack8_bf16(*p, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = (v[e] - cm[e]) * cr[e];
        t = t * cg[e] + cbt[e];
        v[e] = fmaxf(t, 0.f);
      }
      *p = pack8_bf16(v);
    }
  };

  int tsw[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) tsw[s] = (half ^ (((frow + s) >> 1) & 7)) << 4;
  const int rb = ((wm * MI) * PC_PITCH + frow) * 128;

  f32x16_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
  pc_frag_t Bq[3][NJ][2];

  // the first two half-slices of weights leave before the window (their latency is the shorter one)
  pc_static_for<0, LB>([&](auto ic) { b_issue_one(Bq[0], 0, 0, std::integral_constant<int, 0>(), ic); });
  pc_static_for<0, LB>([&](auto ic) { b_issue_one(Bq[1], 0, 0, std::integral_constant<int, 1>(), ic); });

  for (int cb = 0; cb < a.cblocks; ++cb) {
    // ---- the block's window: staged by all four waves, ReLU'd by the wave that staged the piece ----
    window_issue(cb);
    if (bnp) load_bn_table(cb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // pieces + the first weights of the block
    if (RELU) relu_own_pieces();
    __syncthreads();
    if (bnp) {
      bn_transform(cb);
      __syncthreads();
    }
    auto a_read = [&](auto hc, auto k2c, auto i0c, bf16x8_t (&dst)[2]) {
      constexpr int h = decltype(hc)::value, k2 = decltype(k2c)::value, i0 = decltype(i0c)::value;
      constexpr int tap = h >> 1, hh = h & 1, r = tap / 3, s = tap % 3, kk = hh * 2 + k2;
      const int ad = rb + (tsw[s] ^ (kk << 5));
#pragma unroll
      for (int e = 0; e < 2; ++e)
        dst[e] = *reinterpret_cast<const bf16x8_t*>(smem + ad + s * 128 + (i0 + e + r) * PC_ROWSTEP);
    };
    bf16x8_t afp[2][2];
    a_read(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(),
           std::integral_constant<int, 0>(), afp[0]);
    pc_static_for<0, 18>([&](auto hc) {
      constexpr int h = decltype(hc)::value;
      constexpr int h2 = h + 2;
      auto b_ahead = [&](auto idxc) {
        if constexpr (h2 < 18)
          b_issue_one(Bq[h2 % 3], cb, h2 >> 1, std::integral_constant<int, h2 & 1>(), idxc);
        else
          b_issue_one(Bq[h2 % 3], cb + 1, (h2 - 18) >> 1, std::integral_constant<int, h2 & 1>(), idxc);
      };
      // the last load of this half-slice left in the last row pair of half-slice h - 2; behind it only
      // the LB loads of half-slice h - 1 (half-slices 0 and 1 of a block landed with its window)
      if constexpr (h >= 2) {
        if constexpr (NJ == 2)
          pc_wait<LB>(Bq[h % 3][0][0], Bq[h % 3][0][1], Bq[h % 3][1][0], Bq[h % 3][1][1]);
        else
          pc_wait<LB>(Bq[h % 3][0][0], Bq[h % 3][0][1]);
      }
      constexpr int NP = MI;   // 2 k-steps x MI / 2 row pairs
      pc_static_for<0, NP>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int k2 = g / (MI / 2), i0 = (g % (MI / 2)) * 2;
        constexpr int gn = g + 1;
        if constexpr (g % (NP / LB) == 0) b_ahead(std::integral_constant<int, g / (NP / LB)>());
        if constexpr (gn < NP)
          a_read(hc, std::integral_constant<int, gn / (MI / 2)>(),
                 std::integral_constant<int, (gn % (MI / 2)) * 2>(), afp[gn & 1]);
        else if constexpr (h + 1 < 18)
          a_read(std::integral_constant<int, h + 1>(), std::integral_constant<int, 0>(),
                 std::integral_constant<int, 0>(), afp[0]);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i0 + e][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8_t, Bq[h % 3][j][k2]), afp[g & 1][e], acc[i0 + e][j], 0, 0, 0);
      });
    });
    // every wave is done with the window before it is restaged (or becomes the epilogue's staging)
    __syncthreads();
  }
  // The zero reads issued past the last block are still in flight INTO the ring registers.  hipcc sees
  // those registers as dead here and hands them to the epilogue's address arithmetic, which it is
  // free to schedule above a bare `s_waitcnt` -- the late load then lands on top of it (one wave of a
  // tile wrong, a few tiles per launch, different ones each time).  Naming the registers in the
  // wait keeps them allocated until the loads are home.
#pragma unroll
  for (int q3 = 0; q3 < 3; ++q3)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(Bq[q3][j][0]), "+v"(Bq[q3][j][1])::"memory");
  pc_epilogue<BN, WM, WN, MI, NJ, FUSE, QC_TH, QC_HB>(a, acc, smem, tid, lane, wave, wm, wn, frow, half, n,
                                                      ty, tx, nt, st);
}

int pc_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// ---- fragment-ordered weight image ----
// src: row-major operand image [R][Kp] (k = tap * Cin + c; cg_weight_prep's bt_fwd with R = Co, Cin =
// Ci, or bt_bwd with R = Ci, Cin = Co).  dst[rt][cb][tap][kk][lane][8]: row rt*32 + (lane & 31),
// channel cb*64 + kk*16 + (lane >> 5)*8 + e; zero outside.
struct FragItem {
  const bf16_t* src;
  bf16_t* dst;
  int R, Cin, Kp, cblocks;
  int blk0;   // first block of this item
};
constexpr int FRAG_MAXT = 24;
struct FragChunk {
  FragItem it[FRAG_MAXT];
  int cnt;
};
__global__ __launch_bounds__(256) void frag_prep_kernel(FragChunk c) {
  int t = 0;
#pragma unroll 1
  for (int i = 1; i < c.cnt; ++i)
    if ((int)blockIdx.x >= c.it[i].blk0) t = i;
  const FragItem& f = c.it[t];
  // one block = one (rt, cb, tap) unit of 4 KiB = 256 chunks of 16 B
  const int u = blockIdx.x - f.blk0;
  const int tap = u % 9, rc = u / 9;
  const int cb = rc % f.cblocks, rt = rc / f.cblocks;
  const int kk = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = rt * 32 + (lane & 31), ch = cb * 64 + kk * 16 + (lane >> 5) * 8;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (row < f.R && ch < f.Cin)
    v = *reinterpret_cast<const uint4*>(f.src + (int64_t)row * f.Kp + tap * f.Cin + ch);
  *reinterpret_cast<uint4*>(f.dst + ((int64_t)u * 256 + threadIdx.x) * 8) = v;
}

}  // namespace

// ---- host side ----
#ifdef CG_CONV_TIMING
static unsigned long long* g_pconv_tdbg = nullptr;
extern "C" void cg_debug_set_pconv_timing_buffer(void* p) { g_pconv_tdbg = (unsigned long long*)p; }
#endif
size_t cg_weight_frag_elems(int T, int Cin, int R) {
  // OFF by default (round 4): per shape the kernel is 6-12 % faster than hconv_kernel on forward
  // launches and slower on gated data gradients, and inside the train steps the gain is eaten by the
  // second weight image (one more prep launch per network call): resnet128 D sub-step 5.12 -> 5.26 ms,
  // cifar step 7.05 -> 7.21 ms (profiles/r04_pconv_*.txt).  CGAMD_PCONV=1 switches it on.
  static const int enabled = pc_env("CGAMD_PCONV", 0) | pc_env("CGAMD_QCONV", 0);
  if (!enabled || T != 9 || (Cin % 32) != 0 || (R % 8) != 0 || R < 64) return 0;
  return (size_t)cdiv(R, 32) * cdiv(Cin, 64) * 9 * 2048;
}

void cg_weight_frag_launch(const cgFragJob* jobs, int n, hipStream_t st) {
  for (int i0 = 0; i0 < n; i0 += FRAG_MAXT) {
    const int cnt = (n - i0) < FRAG_MAXT ? (n - i0) : FRAG_MAXT;
    FragChunk c;
    c.cnt = cnt;
    int blocks = 0;
    for (int i = 0; i < cnt; ++i) {
      const cgFragJob& j = jobs[i0 + i];
      FragItem& f = c.it[i];
      f.src = (const bf16_t*)j.rowmajor;
      f.dst = (bf16_t*)j.frag;
      f.R = j.R; f.Cin = j.Cin; f.Kp = (9 * j.Cin + 7) & ~7;
      f.cblocks = cdiv(j.Cin, 64);
      f.blk0 = blocks;
      blocks += cdiv(j.R, 32) * f.cblocks * 9;
    }
    if (blocks > 0) frag_prep_kernel<<<blocks, 256, 0, st>>>(c);
  }
}

bool cg_pconv_geom_ok(const cgConvGeom* g) {
  if (g->S != 1 || g->U != 1 || g->kh != 3 || g->kw != 3 || g->pt != 1 || g->pl != 1) return false;
  if (g->Ho != g->Hin || g->Wo != g->Win) return false;
  if ((g->Hin % PC_TH) != 0 || (g->Win % PC_TW) != 0) return false;
  if ((g->Ci % 32) != 0 || (g->Co % 8) != 0 || g->Co < 64) return false;
  if (cg_weight_frag_elems(9, g->Ci, g->Co) == 0) return false;
  if ((int64_t)(PC_TH + 2) * g->Win * g->Ci * 2 >= (1ll << 30)) return false;
  return true;
}

static int pc_bn(const cgConvGeom* g) {
  static const int force64 = pc_env("CGAMD_QCONV_BN64", 0);   // experiment: 64-channel tiles, 3 workgroups / CU
  return (g->Co <= 64 || force64) ? 64 : 128;
}

bool cg_pconv_use(const cgConvGeom* g) {
  static const int enabled = pc_env("CGAMD_PCONV", 0);
  static const int min_items = pc_env("CGAMD_PCONV_MIN", 224);
  if (!enabled || !cg_pconv_geom_ok(g)) return false;
  const int64_t items = (int64_t)g->N * (g->Hin / PC_TH) * (g->Win / PC_TW) * cdiv(g->Co, pc_bn(g));
  return items >= min_items;
}

bool cg_qconv_use(const cgConvGeom* g) {
  static const int enabled = pc_env("CGAMD_QCONV", 0);
  static const int min_items = pc_env("CGAMD_QCONV_MIN", 100);
  if (!enabled) return false;
  if (g->S != 1 || g->U != 1 || g->kh != 3 || g->kw != 3 || g->pt != 1 || g->pl != 1) return false;
  if (g->Ho != g->Hin || g->Wo != g->Win) return false;
  if ((g->Hin % QC_TH) != 0 || (g->Win % PC_TW) != 0) return false;
  if ((g->Ci % 32) != 0 || (g->Co % 8) != 0 || g->Co < 64) return false;
  if (cg_weight_frag_elems(9, g->Ci, g->Co) == 0) return false;
  if ((int64_t)(QC_TH + 2) * g->Win * g->Ci * 2 >= (1ll << 30)) return false;
  const int64_t items = (int64_t)g->N * (g->Hin / QC_TH) * (g->Win / PC_TW) * cdiv(g->Co, pc_bn(g));
  return items >= min_items;
}

void cg_qconv_launch(const cgConvGeom* g, const void* in, const void* bt, void* out, int out_is_f32,
                     const float* bias, const void* gate_in, const void* gate_out, float slope_out,
                     const void* residual, const cgConvFusion* fu, hipStream_t st) {
  PConvArgs a;
  memset(&a, 0, sizeof(a));
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
  const int Kp = (9 * g->Ci + 7) & ~7;
  a.in = (const bf16_t*)in;
  a.btf = (const bf16_t*)bt + (size_t)g->Co * Kp;
  a.btf_bytes = (uint32_t)(cg_weight_frag_elems(9, g->Ci, g->Co) * 2);
  a.out = out;
  a.bias = bias;
  a.self_gate = (gate_out != nullptr && gate_out == out);
  a.gate_out = a.self_gate ? nullptr : (const bf16_t*)gate_out;
  a.residual = (const bf16_t*)residual;
  a.N = g->N; a.H = g->Hin; a.W = g->Win; a.Ci = g->Ci; a.Co = g->Co;
  a.cblocks = cdiv(g->Ci, 64);
  a.nslices = 9 * a.cblocks;
  a.cotiles = cdiv(g->Co, 32);
  a.tiles_x = g->Win / PC_TW;
  a.tiles_y = g->Hin / QC_TH;
  const int bn = pc_bn(g);
  a.ntiles = cdiv(g->Co, bn);
  a.nitems = g->N * a.tiles_y * a.tiles_x * a.ntiles;
  a.out_f32 = out_is_f32;
  a.slope_out = slope_out;
  a.dNt = make_fastdiv(a.ntiles);
  a.dTx = make_fastdiv(a.tiles_x);
  a.dTy = make_fastdiv(a.tiles_y);
  const bool relu = gate_in != nullptr && a.bn_mean == nullptr;
  CgProfScope prof(bn == 128 ? CG_PROF_PCONV_128 : CG_PROF_PCONV_64, g, st);
#define QC_LAUNCH2(BN_, FUSE_)                                                       \
  do {                                                                               \
    if (relu) qconv_kernel<BN_, true, FUSE_><<<a.nitems, 256, 0, st>>>(a);           \
    else qconv_kernel<BN_, false, FUSE_><<<a.nitems, 256, 0, st>>>(a);              \
  } while (0)
#define QC_LAUNCH(BN_)                                                               \
  do {                                                                               \
    if (a.pool) QC_LAUNCH2(BN_, 2);                                                  \
    else if (a.bn_mean || a.stats) QC_LAUNCH2(BN_, 1);                               \
    else QC_LAUNCH2(BN_, 0);                                                         \
  } while (0)
  if (bn == 128) QC_LAUNCH(128);
  else QC_LAUNCH(64);
#undef QC_LAUNCH
#undef QC_LAUNCH2
}

int cg_pconv_stats_rows(const cgConvGeom* g) { return g->N * (g->Hin / PC_TH) * (g->Win / PC_TW); }

void cg_pconv_launch(const cgConvGeom* g, const void* in, const void* bt, void* out, int out_is_f32,
                     const float* bias, const void* gate_in, const void* gate_out, float slope_out,
                     const void* residual, const cgConvFusion* fu, hipStream_t st) {
  PConvArgs a;
  memset(&a, 0, sizeof(a));
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
  const int Kp = (9 * g->Ci + 7) & ~7;
  a.in = (const bf16_t*)in;
  a.btf = (const bf16_t*)bt + (size_t)g->Co * Kp;   // the fragment image follows the row-major one
  a.btf_bytes = (uint32_t)(cg_weight_frag_elems(9, g->Ci, g->Co) * 2);
  a.out = out;
  a.bias = bias;
  a.self_gate = (gate_out != nullptr && gate_out == out);
  a.gate_out = a.self_gate ? nullptr : (const bf16_t*)gate_out;
  a.residual = (const bf16_t*)residual;
  a.N = g->N; a.H = g->Hin; a.W = g->Win; a.Ci = g->Ci; a.Co = g->Co;
  a.cblocks = cdiv(g->Ci, 64);
  a.nslices = 9 * a.cblocks;
  a.cotiles = cdiv(g->Co, 32);
  a.tiles_x = g->Win / PC_TW;
  a.tiles_y = g->Hin / PC_TH;
  const int bn = pc_bn(g);
  a.ntiles = cdiv(g->Co, bn);
  a.nitems = g->N * a.tiles_y * a.tiles_x * a.ntiles;
  a.out_f32 = out_is_f32;
  a.slope_out = slope_out;
  a.dNt = make_fastdiv(a.ntiles);
  a.dTx = make_fastdiv(a.tiles_x);
  a.dTy = make_fastdiv(a.tiles_y);
  static const int grid_max = pc_env("CGAMD_PCONV_GRID", 256);
  static const int wm2 = pc_env("CGAMD_PCONV_WM2", 0);
  static const int pipe = pc_env("CGAMD_PCONV_PIPE", 1);
#ifdef CG_CONV_TIMING
  a.tdbg = g_pconv_tdbg;
#endif
  static const int prio = pc_env("CGAMD_PCONV_PRIO", 1);
  a.prio = prio;
  static const int dbg = pc_env("CGAMD_PCONV_DBG", 0);
  a.dbg = dbg;
  const int grid = a.nitems < grid_max ? a.nitems : grid_max;
  const bool relu = gate_in != nullptr && a.bn_mean == nullptr;   // the BN prologue includes the ReLU
  CgProfScope prof(bn == 128 ? CG_PROF_PCONV_128 : CG_PROF_PCONV_64, g, st);
#define PC_LAUNCH3(BN_, WM_, FUSE_, PIPE_)                                          \
  do {                                                                              \
    if (relu) pconv_kernel<BN_, WM_, true, FUSE_, PIPE_><<<grid, 512, 0, st>>>(a);  \
    else pconv_kernel<BN_, WM_, false, FUSE_, PIPE_><<<grid, 512, 0, st>>>(a);      \
  } while (0)
#define PC_LAUNCH2(BN_, WM_, FUSE_)                                              \
  do {                                                                           \
    if (pipe) PC_LAUNCH3(BN_, WM_, FUSE_, true);                                 \
    else PC_LAUNCH3(BN_, WM_, FUSE_, false);                                     \
  } while (0)
#define PC_LAUNCH(BN_, WM_)                                                      \
  do {                                                                           \
    if (a.pool) PC_LAUNCH2(BN_, WM_, 2);                                         \
    else if (a.bn_mean || a.stats) PC_LAUNCH2(BN_, WM_, 1);                      \
    else PC_LAUNCH2(BN_, WM_, 0);                                                \
  } while (0)
  if (bn == 128) {
    if (wm2) PC_LAUNCH(128, 2);
    else PC_LAUNCH(128, 4);
  } else {
    PC_LAUNCH(64, 4);
  }
#undef PC_LAUNCH
#undef PC_LAUNCH2
#undef PC_LAUNCH3
}
