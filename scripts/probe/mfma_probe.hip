// Stand-alone ladder for the main loop of the 3x3 convolution kernels (cg_conv_pers.hip /
// cg_conv_halo.hip): what does each ingredient of the K loop cost on top of a pure MFMA loop?
// One persistent 8-wave workgroup per CU (2 waves per SIMD, 128 pixels x 64 channels per wave = 4 x 2
// accumulator tiles of 32x32), `iters` half-slices of 16 MFMAs per wave.  MODE bits:
//   1  window fragments from LDS (2 ds_read_b128 per 4 MFMAs, read one pair ahead)
//   2  weight fragments straight from global memory (4 buffer_load_dwordx4 per half-slice, ring of 3,
//      counted vmcnt), L2-resident
//   4  one 1-KiB LDS-DMA piece per half-slice and wave from a large (HBM) buffer
//   8  a workgroup barrier every 18 half-slices
//  32  (with 4) the DMA pieces walk 8 MiB per workgroup of a 2 GiB buffer (HBM, never re-read) in the
//      window pattern: 8 rows of 128 B at a 256-byte pitch per piece
//  64  (with 4) the DMA pieces stream 8 MiB per workgroup of the 2 GiB buffer sequentially (full lines);
//      template DMAX = pieces per wave and half-slice: how much does HBM traffic cost beside MFMA work?
//  16  operands change between MFMAs (xor with the loop counter): register-resident operands that
//      never toggle clock higher
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probe/mfma_probe.hip -o scripts/probe/mfma_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;

__device__ __forceinline__ i32x4_t make_rsrc(const void* base, uint32_t bytes) {
  const uint64_t p = (uint64_t)base;
  i32x4_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)p);
  r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((p >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
template <int IMM>
__device__ __forceinline__ void bload(i32x4_t& dst, uint32_t voff, i32x4_t rs, uint32_t soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=&v"(dst) : "v"(voff), "s"(rs), "s"(soff), "n"(IMM));
}
template <int N>
__device__ __forceinline__ void wait4(i32x4_t& a, i32x4_t& b, i32x4_t& c, i32x4_t& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
__device__ __forceinline__ void dma16(i32x4_t rs, uint32_t voff, uint32_t soff, uint32_t lds) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds), "s"(soff));
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>()); static_for<I + 1, N>(f); }
}

constexpr int WIN = 77 * 1024;

template <int MODE, int DMAX = 1>
__global__ __launch_bounds__(512, 2) void probe(const uint16_t* wts, uint32_t wbytes, const uint16_t* big,
                                                int iters, float* sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * WIN];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // LDS filled with pseudo-random bf16
  for (int i = tid; i < 2 * WIN / 4; i += 512) {
    uint32_t s = (uint32_t)(i * 2654435761u + blockIdx.x * 97u);
    s = s * 1664525u + 1013904223u;
    reinterpret_cast<uint32_t*>(smem)[i] = (s & 0x807f807fu) | 0x3f003f00u;
  }
  __syncthreads();
  const i32x4_t rs_w = make_rsrc(wts, wbytes);
  const uint32_t region = (MODE & (32 | 64)) ? (8u << 20) : (1u << 20);
  const i32x4_t rs_b = make_rsrc((const char*)big + (size_t)blockIdx.x * region, region + 4096u);
  const uint32_t dma_voff = (MODE & 32) ? (uint32_t)((lane >> 3) * 256 + (lane & 7) * 16) : (uint32_t)lane * 16u;
  const uint32_t dma_step = (MODE & 32) ? 2048u : 1024u;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
  f32x16_t acc[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
  i32x4_t Bq[3][2][2];
  for (int q = 0; q < 3; ++q) for (int j = 0; j < 2; ++j) for (int k = 0; k < 2; ++k)
    for (int e = 0; e < 4; ++e) Bq[q][j][k][e] = 0x3e803e80 ^ (lane * 0x01010101 + q + j + k + e);
  const int rowbase = ((wave >> 1) * 4 * 34 + (lane & 31)) * 128;
  int tsw[3];
  for (int s = 0; s < 3; ++s) tsw[s] = ((lane >> 5) ^ ((((lane & 31) + s) >> 1) & 7)) << 4;
  const uint32_t lane16 = lane * 16u;
  auto a_read = [&](int ad, auto i0c, bf16x8_t (&dst)[2]) {
    constexpr int i0 = decltype(i0c)::value;
    for (int e = 0; e < 2; ++e)
      dst[e] = *reinterpret_cast<const bf16x8_t*>(smem + ad + (i0 + e) * 34 * 128);
  };
  bf16x8_t afp[2][2];
  for (int q = 0; q < 2; ++q) for (int e = 0; e < 2; ++e) afp[q][e] = __builtin_bit_cast(bf16x8_t, Bq[q][e][0]);
  if (MODE & 2) {
    static_for<0, 4>([&](auto ic) { constexpr int i = decltype(ic)::value; bload<(i & 1) * 1024>(Bq[0][i >> 1][i & 1], lane16, rs_w, (wave & 1) * 8192 + (i >> 1) * 4096); });
    static_for<0, 4>([&](auto ic) { constexpr int i = decltype(ic)::value; bload<2048 + (i & 1) * 1024>(Bq[1][i >> 1][i & 1], lane16, rs_w, (wave & 1) * 8192 + (i >> 1) * 4096); });
  }
  int buf = 0;
  uint32_t soff_w = 0, dma_off = 0;
  for (int it = 0; it < iters; it += 18) {
    const int rb = rowbase + buf * WIN;
    if (MODE & 1) a_read(rb + tsw[0], std::integral_constant<int, 0>(), afp[0]);
    static_for<0, 18>([&](auto hc) {
      constexpr int h = decltype(hc)::value;
      constexpr int tap = h >> 1, hh = h & 1, r = tap / 3, s = tap % 3;
      constexpr int h2 = (h + 2) % 18;
      auto b_ahead = [&](auto idxc) {
        constexpr int idx = decltype(idxc)::value;
        if (MODE & 2)
          bload<(h2 & 1) * 2048 + (idx & 1) * 1024>(Bq[(h + 2) % 3][idx >> 1][idx & 1], lane16, rs_w,
                                                    soff_w + ((wave & 1) * 2 + (idx >> 1)) * 4096 * 18 + (h2 >> 1) * 4096);
      };
      constexpr int allow = 4 + ((MODE & 4) ? DMAX : 0);
      if (MODE & 2) wait4<allow>(Bq[h % 3][0][0], Bq[h % 3][0][1], Bq[h % 3][1][0], Bq[h % 3][1][1]);
      if (MODE & 4) {
#pragma unroll
        for (int dx = 0; dx < DMAX; ++dx) {
          dma16(rs_b, dma_voff, dma_off + wave * dma_step, lds0 + (buf ^ 1) * WIN + ((wave + 8 * ((h + dx) % 9)) * 1024));
          dma_off = (dma_off + 8 * dma_step) & (region - 1);
        }
      }
      static_for<0, 4>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int k2 = g >> 1, i0 = (g & 1) * 2, kk = hh * 2 + k2;
        b_ahead(gc);
        if (MODE & 1) {
          constexpr int gn = g + 1;
          if constexpr (gn < 4) {
            constexpr int kkn = hh * 2 + (gn >> 1);
            a_read(rb + (tsw[s] ^ (kkn << 5)) + s * 128 + r * 34 * 128, std::integral_constant<int, (gn & 1) * 2>(), afp[gn & 1]);
          } else if constexpr (h + 1 < 18) {
            constexpr int tn = (h + 1) >> 1, rn = tn / 3, sn = tn % 3, kkn = ((h + 1) & 1) * 2;
            a_read(rb + (tsw[sn] ^ (kkn << 5)) + sn * 128 + rn * 34 * 128, std::integral_constant<int, 0>(), afp[0]);
          }
        }
        for (int e = 0; e < 2; ++e)
          for (int j = 0; j < 2; ++j) {
            bf16x8_t bw = __builtin_bit_cast(bf16x8_t, Bq[h % 3][j][k2]);
            bf16x8_t aw = afp[g & 1][e];
            if ((MODE & 16) && !(MODE & 1)) {   // toggling register operands
              i32x4_t t = __builtin_bit_cast(i32x4_t, aw);
              t[0] ^= (it + h * 4 + g) * 0x00010001;
              aw = __builtin_bit_cast(bf16x8_t, t);
              afp[g & 1][e] = aw;
            }
            acc[i0 + e][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw, aw, acc[i0 + e][j], 0, 0, 0);
          }
      });
    });
    if (MODE & 8) asm volatile("s_barrier" ::: "memory");
    soff_w = 0;
    buf ^= (MODE & 4) ? 1 : 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float sres = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) sres += acc[i][j][lane & 15];
  if (sres == 123456.789f) sink[0] = sres;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE, int DMAX = 1>
void run(const uint16_t* w, uint32_t wb, const uint16_t* big, float* sink, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  probe<MODE, DMAX><<<256, 512>>>(w, wb, big, iters, sink);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    probe<MODE, DMAX><<<256, 512>>>(w, wb, big, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double fl = 256.0 * 8 * iters * 16.0 * 2 * 32 * 32 * 16;
  if (MODE & 4) printf("  [DMA x%d: %.2f GB moved, %.2f TB/s]  ", DMAX, 256.0 * 8 * iters * DMAX * 1024 / 1e9,
                       256.0 * 8 * iters * DMAX * 1024 / best / 1e9);
  printf("mode %2d (%s%s%s%s%s): %.3f ms  %.0f TFLOP/s\n", MODE, (MODE & 1) ? "lds " : "", (MODE & 2) ? "wload " : "",
         (MODE & 4) ? "dma " : "", (MODE & 8) ? "barrier " : "", (MODE & 32) ? "hbm-window " : "", best, fl / best / 1e9);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 18 * 400;
  uint16_t *w, *big; float* sink;
  const uint32_t wb = 4u << 20;
  CK(hipMalloc(&w, wb)); CK(hipMalloc(&big, (size_t)2052 << 20)); CK(hipMalloc(&sink, 64));
  std::vector<uint16_t> h(wb / 2);
  uint32_t s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(((s >> 16) & 0x807f) | 0x3e80); }
  CK(hipMemcpy(w, h.data(), wb, hipMemcpyHostToDevice));
  for (size_t o = 0; o < ((size_t)2052 << 20); o += wb) CK(hipMemcpy((char*)big + o, h.data(), wb, hipMemcpyHostToDevice));
  run<0>(w, wb, big, sink, iters);
  run<1>(w, wb, big, sink, iters);
  run<2>(w, wb, big, sink, iters);
  run<3>(w, wb, big, sink, iters);
  run<4>(w, wb, big, sink, iters);
  run<36>(w, wb, big, sink, iters);
  run<7>(w, wb, big, sink, iters);
  run<39>(w, wb, big, sink, iters);
  run<47>(w, wb, big, sink, iters);
  // is HBM traffic free beside MFMA work?  sequential full-line streaming, 0 / 1 / 2 / 4 KiB per wave and half-slice
  run<0>(w, wb, big, sink, iters);
  run<4 | 64, 1>(w, wb, big, sink, iters);
  run<4 | 64, 2>(w, wb, big, sink, iters);
  run<4 | 64, 4>(w, wb, big, sink, iters);
  run<3>(w, wb, big, sink, iters);
  run<7 | 64, 1>(w, wb, big, sink, iters);
  run<7 | 64, 2>(w, wb, big, sink, iters);
  run<7 | 64, 4>(w, wb, big, sink, iters);
  run<8>(w, wb, big, sink, iters);
  run<9>(w, wb, big, sink, iters);
  run<15>(w, wb, big, sink, iters);
  run<0>(w, wb, big, sink, iters);
  return 0;
}
