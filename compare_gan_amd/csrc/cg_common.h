// Shared device/host helpers for libcgamd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "cgamd.h"

typedef uint16_t bf16_t;  // raw bfloat16 storage

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ---- error bookkeeping -------------------------------------------------------------------
void cg_set_error(const char* fmt, ...);
#define CG_FAIL(code, ...)          \
  do {                              \
    cg_set_error(__VA_ARGS__);      \
    return (code);                  \
  } while (0)
#define CG_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess)                                                 \
      CG_FAIL(CG_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__));      \
  } while (0)

// ---- bf16 conversion (round-to-nearest-even, same as torch / TF) ---------------------------
__host__ __device__ __forceinline__ float bf2f(bf16_t v) {
  union { uint32_t u; float f; } c;
  c.u = ((uint32_t)v) << 16;
  return c.f;
}
__host__ __device__ __forceinline__ bf16_t f2bf(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  uint32_t u = c.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// 8 floats <-> 8 packed bf16 with the hardware conversions (v_cvt_pk_bf16_f32: round to nearest
// even, NaN stays NaN -- the same results as f2bf above)
typedef __attribute__((ext_vector_type(8))) float f32x8_t;
__device__ __forceinline__ uint4 pack8_bf16(const float* v) {
  const f32x8_t f = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
  return __builtin_bit_cast(uint4, __builtin_convertvector(f, bf16x8_t));
}
__device__ __forceinline__ void unpack8_bf16(uint4 q, float* v) {
  v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u);
  v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
  v[4] = __uint_as_float(q.z << 16); v[5] = __uint_as_float(q.z & 0xffff0000u);
  v[6] = __uint_as_float(q.w << 16); v[7] = __uint_as_float(q.w & 0xffff0000u);
}

typedef __attribute__((ext_vector_type(4))) float cg_f32x4_t;
typedef __attribute__((ext_vector_type(4))) __bf16 cg_bf16x4_t;
__device__ __forceinline__ uint2 pack4_bf16(const float* v) {
  const cg_f32x4_t f = {v[0], v[1], v[2], v[3]};
  return __builtin_bit_cast(uint2, __builtin_convertvector(f, cg_bf16x4_t));
}

struct __attribute__((aligned(16))) bf16x8_raw {
  bf16_t v[8];
};

// ---- division by a runtime constant --------------------------------------------------------
struct FastDiv {
  uint32_t d, mul, shr;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  uint32_t s = 0;
  while ((1ull << s) < d) ++s;
  f.shr = s;
  f.mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << s) - d)) / d + 1);
  return f;
}
// valid for 0 <= n < 2^31
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  return (__umulhi(n, f.mul) + n) >> f.shr;
}

// ---- reductions ------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread.
__device__ __forceinline__ float block_sum_256(float v, float* sm4) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm4[w] = v;
  __syncthreads();
  return sm4[0] + sm4[1] + sm4[2] + sm4[3];
}

// ---- optional HIP-event timing of the convolution kernels (cg_error.hip) -------------------------
// one family per kernel symbol, so that the numbers line up with `rocprofv3 --kernel-trace --stats`
enum {
  CG_PROF_HALO_CONV = 0,          // halo_conv_kernel<*>
  CG_PROF_HCONV_128,              // hconv_kernel<128, *> (cg_conv_halo.hip)
  CG_PROF_HCONV_64,               // hconv_kernel<64, *>
  CG_PROF_FAST_CONV_128x128,      // fast_conv_kernel<128, 128, *>
  CG_PROF_FAST_CONV_64x128,       // fast_conv_kernel<64, 128, *>
  CG_PROF_FAST_CONV_128x64,       // fast_conv_kernel<128, 64, *>
  CG_PROF_FAST_CONV_128x32,       // fast_conv_kernel<128, 32, *>
  CG_PROF_STEM_FWD,               // stem_fwd_kernel<*>
  CG_PROF_GCONV_GENERIC,          // gconv_kernel<...> (channel counts not a multiple of 64, leaky gates)
  CG_PROF_HWGRAD,                 // hwgrad_kernel<*> (cg_conv_halo.hip; + split reduce)
  CG_PROF_HALO_WGRAD,             // halo_wgrad_kernel<*> (+ split reduce)
  CG_PROF_FAST_WGRAD_128,         // fast_wgrad_kernel<128, *> (+ split reduce)
  CG_PROF_FAST_WGRAD_64,          // fast_wgrad_kernel<64, *> (+ split reduce)
  CG_PROF_STEM_WGRAD,             // stem_wgrad_kernel<*> incl. the narrow-output (adjoint) form
  CG_PROF_GWGRAD_GENERIC,         // gwgrad_kernel<...>
  CG_PROF_SCONV,                  // sconv_kernel<*> (cg_conv_small.hip)
  CG_PROF_SWGRAD,                 // swgrad_kernel<*> (cg_conv_small.hip)
  CG_PROF_FAST_CONV_128x192,      // fast_conv_kernel<128, 192, *>
  CG_PROF_COUNT
};
void cg_prof_begin(int family, double flops, double bytes, hipStream_t st);
void cg_prof_end(int family, hipStream_t st);
bool cg_prof_enabled();
// useful MACs x 2 (structural zeros of a zero-inserted input are not counted) and the minimum bf16
// HBM traffic 2 * (in + out + weights) bytes of one launch
void cg_conv_algorithmic_cost(const cgConvGeom* g, double* flops, double* bytes);
void cg_prof_tag_geom(const cgConvGeom* g);
struct CgProfScope {
  int fam;
  hipStream_t st;
  CgProfScope(int family, const cgConvGeom* g, hipStream_t s) : fam(family), st(s) {
    if (cg_prof_enabled()) {
      double f, b;
      cg_conv_algorithmic_cost(g, &f, &b);
      cg_prof_tag_geom(g);
      cg_prof_begin(fam, f, b, st);
    }
  }
  ~CgProfScope() { cg_prof_end(fam, st); }
};

// ---- LDS-DMA (buffer_load ... lds) issued from inline asm ----------------------------------------
// hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the first ds_read_b64_tr_b16 builtin that
// follows a __builtin_amdgcn_raw_ptr_buffer_load_lds in program order (it cannot tell the LDS region
// the DMA writes from the one the read touches), which drains the prefetch of the NEXT tile before
// the current one is multiplied -- the weight-gradient kernels lost their double buffering to it.
// Issued from inline asm the DMA is invisible to that pass: the kernel orders its LDS reads behind
// the DMA with its own counted `s_waitcnt vmcnt(N)` (+ a barrier where other waves staged the data),
// and every other wait (LDS reads -> MFMA) stays compiler-managed.
//   rs   : the buffer descriptor as four wave-uniform dwords (cg_make_rsrc)
//   voff : per-lane byte offset (>= 0x80000000: the bounds check returns zeros)
//   soff : wave-uniform byte offset;  lds : wave-uniform LDS byte address, lane l lands at lds + 16 l
// M0 (the DMA's LDS base) is saved and restored inside the statement; s_nop 4 covers a descriptor /
// offset SGPR written by v_readfirstlane just before (cdna_hip_programming.md section 5.7).
typedef __attribute__((ext_vector_type(4))) int cg_i32x4_t;
__device__ __forceinline__ cg_i32x4_t cg_make_rsrc(const void* base, uint32_t bytes) {
  const uint64_t p = (uint64_t)base;
  cg_i32x4_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)p);
  r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((p >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void cg_dma16_asm(cg_i32x4_t rs, uint32_t voff, uint32_t soff,
                                             uint32_t lds) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 4\n\t"
      "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rs), "s"(lds), "s"(soff)
      : "memory");
}
// The same without the M0 save / restore and with the single wait state an M0 write needs before
// an LDS-DMA: for kernels in which NOTHING else uses M0 (no LDS-DMA builtin, no s_movrel, no
// s_sendmsg -- check the .s) and whose descriptor / offset SGPRs are written by scalar instructions
// long before (cg_make_rsrc at kernel entry).
__device__ __forceinline__ void cg_dma16_asm_m0(cg_i32x4_t rs, uint32_t voff, uint32_t soff,
                                                uint32_t lds) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %1, %3 offen lds"
      :
      : "v"(voff), "s"(rs), "s"(lds), "s"(soff)
      : "memory");
}
// global_load_lds_dwordx4 form (flat global address per lane instead of a buffer descriptor)
__device__ __forceinline__ void cg_glds16_asm(const void* gptr, uint32_t lds) {
  uint32_t keep;
  const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds);
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gptr), "s"(l)
      : "memory");
}
__device__ __forceinline__ uint32_t cg_lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
