// Calibration microbenchmarks for bench.py (SURVEY.md section 8d asks for MEASURED roofline
// denominators next to the datasheet ones): boxes of one pool differ by +-15 % in sustained clocks,
// so a kernel's fraction of the dense bf16 MFMA peak / of the HBM bandwidth is only comparable
// between runs when it is normalised by what the same box reaches on a pure MFMA loop / a pure copy
// in the same process.  Not part of the hot path.
#include "cg_common.h"

namespace {

// every wave runs `iters` rounds of 8 independent v_mfma_f32_32x32x16_bf16 on register operands
// (full-range pseudo-random bf16 values: zero-filled operands clock ~20 % higher, see
// cdna_hip_programming.md section 5.4 rule 25); 2 x 32 x 32 x 16 flops per wave-instruction
// zero_operands: the same loop on all-zero operands (the chip holds a higher clock under it: what
// separates "the loop is not tight" from "the clock is power-capped" when the random-operand rate
// falls short of the datasheet's)
__global__ __launch_bounds__(256) void calib_mfma_kernel(int iters, int zero_operands, float* sink) {
  const int lane = threadIdx.x & 63;
  uint32_t seed = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  union { bf16x8_t v; uint32_t u[4]; } a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    seed = seed * 1664525u + 1013904223u;
    // two bf16 values in [-2, 2): sign random, exponent 0x3f / 0x3e, mantissa random
    a.u[i] = zero_operands ? 0u : (seed & 0x807f807fu) | 0x3f003f00u;
    seed = seed * 1664525u + 1013904223u;
    b.u[i] = zero_operands ? 0u : (seed & 0x807f807fu) | 0x3e803e80u;
  }
  f32x16_t acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[t], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) s += acc[t][lane & 15];
  if (s == 123456.789f) sink[0] = s;   // keeps the loop alive without a store in practice
}

__global__ __launch_bounds__(256) void calib_copy_kernel(const float4* __restrict__ src,
                                                         float4* __restrict__ dst, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    dst[i] = src[i];
}

}  // namespace

extern "C" int cg_calib_mfma_bf16(int blocks, int iters, float* sink, double* flops,
                                  cgStream stream) {
  if (blocks <= 0 || iters <= 0 || !sink) CG_FAIL(CG_ERR_BAD_ARG, "cg_calib_mfma_bf16: bad argument");
  calib_mfma_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(iters, 0, sink);
  CG_CHECK_LAUNCH("cg_calib_mfma_bf16");
  if (flops) *flops = (double)blocks * 4.0 * iters * 8.0 * 2.0 * 32 * 32 * 16;
  return CG_OK;
}

extern "C" int cg_calib_mfma_bf16_zero(int blocks, int iters, float* sink, double* flops,
                                       cgStream stream) {
  if (blocks <= 0 || iters <= 0 || !sink)
    CG_FAIL(CG_ERR_BAD_ARG, "cg_calib_mfma_bf16_zero: bad argument");
  calib_mfma_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(iters, 1, sink);
  CG_CHECK_LAUNCH("cg_calib_mfma_bf16_zero");
  if (flops) *flops = (double)blocks * 4.0 * iters * 8.0 * 2.0 * 32 * 32 * 16;
  return CG_OK;
}

extern "C" int cg_calib_copy(const void* src, void* dst, size_t bytes, cgStream stream) {
  if (!src || !dst || (bytes % 16) != 0) CG_FAIL(CG_ERR_BAD_ARG, "cg_calib_copy: bad argument");
  calib_copy_kernel<<<4096, 256, 0, (hipStream_t)stream>>>((const float4*)src, (float4*)dst,
                                                           (int64_t)(bytes / 16));
  CG_CHECK_LAUNCH("cg_calib_copy");
  return CG_OK;
}
