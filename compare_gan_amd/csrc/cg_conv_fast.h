// Internal interface of the fast convolution paths (cg_conv_fast.hip), used by the dispatchers in
// cg_gconv.hip.  Not part of the C-ABI.
#pragma once
#include "cg_common.h"

bool cg_fast_conv_supported(const cgConvGeom* g, const void* in, const void* gate_in,
                            float slope_in);
void cg_fast_conv_launch(const cgConvGeom* g, const void* in, const void* bt, void* out,
                         int out_is_f32, const float* bias, const void* gate_in,
                         const void* gate_out, float slope_out, const void* residual,
                         hipStream_t st);

// the same kernels on channel slices: consecutive pixels of the input / output are in_ld / out_ld
// elements apart (>= Ci / Co, multiples of 8; the gate / residual tensors, if any, share out's pitch)
bool cg_fast_conv_ld_supported(const cgConvGeom* g, int in_ld, int out_ld);
void cg_fast_conv_launch_ld(const cgConvGeom* g, const void* in, int in_ld, const void* bt, void* out,
                            int out_ld, int out_is_f32, const float* bias, const void* gate_in,
                            const void* gate_out, float slope_out, const void* residual,
                            hipStream_t st);

// ... conv + bias with ReLU on the output channels [0, relu_cols) only (relu_cols % 8 == 0)
void cg_fast_conv_launch_ld_cols(const cgConvGeom* g, const void* in, int in_ld, const void* bt,
                                 void* out, int out_ld, int out_is_f32, const float* bias,
                                 int relu_cols, hipStream_t st);

// halo-staged kernel for unit-stride <= 3x3 filters on >= 16x16 maps (cg_conv_halo.hip)
bool cg_hconv_supported(const cgConvGeom* g, const void* in, const void* gate_in, float slope_in);
bool cg_hconv_narrow(const cgConvGeom* g);   // Co < 8: scalar epilogue, no gate tensor / residual
bool cg_hconv_narrow_ok(const cgConvGeom* g);   // ... and the geometry hconv_kernel<64, *, *, 3> covers
void cg_hconv_launch(const cgConvGeom* g, const void* in, const void* bt, void* out,
                     int out_is_f32, const float* bias, const void* gate_in, const void* gate_out,
                     float slope_out, const void* residual, hipStream_t st);

// 64 -> 64 channel 3x3 form with register-resident weights (persistent 128-pixel tiles)
bool cg_hconv_rw_supported(const cgConvGeom* g, const void* in, const void* gate_in,
                           float slope_in);
void cg_hconv_rw_launch(const cgConvGeom* g, const void* in, const void* bt, void* out,
                        int out_is_f32, const float* bias, const void* gate_in,
                        const void* gate_out, float slope_out, const void* residual,
                        hipStream_t st);
// small-map 3x3 form: 64-pixel x 64-channel workgroups, K split over the waves (cg_conv_small.hip)
bool cg_sconv_geom_ok(const cgConvGeom* g);
bool cg_sconv_supported(const cgConvGeom* g, const void* in, const void* gate_in, float slope_in);
void cg_sconv_launch(const cgConvGeom* g, const void* in, const void* bt, void* out, int out_is_f32,
                     const float* bias, const void* gate_in, const void* gate_out, float slope_out,
                     const void* residual, hipStream_t st);
// small-map weight gradient: one tap x 64 x 64 per workgroup, all pixels, no partials; several
// layers per launch (cg_conv_small.hip)
bool cg_swgrad_geom_ok(const cgConvGeom* g);
bool cg_swgrad_supported(const cgConvGeom* g, const void* in, const void* gate_in, float slope_in,
                         const void* gate_dy, bool grouped);
void cg_swgrad_launch(const cgConvGeom* g, const void* in, const void* gate_in, const void* dy,
                      float* dw, int accumulate, float* dbias, hipStream_t st);
#define CG_SWGRAD_MAX_JOBS 24
void cg_swgrad_launch_multi(const cgConvGeom* const* geoms, const void* const* ins,
                            const int* relus, const void* const* dys, float* const* dws,
                            const int* accumulates, float* const* dbs, int n, hipStream_t st);
// the same launch with the fused batch-norm prologue / statistics epilogue (cgConvFusion, cgamd.h)
void cg_hconv_launch_fused(const cgConvGeom* g, const void* in, const void* bt, void* out,
                           int out_is_f32, const float* bias, const void* gate_in,
                           const void* gate_out, float slope_out, const void* residual,
                           const cgConvFusion* fu, hipStream_t st);
int cg_hconv_stats_rows(const cgConvGeom* g);
int cg_hconv_stats_phases(const cgConvGeom* g);   // phase blocks of the statistics rows (1 or U*U)
bool cg_hconv_geom_ok(const cgConvGeom* g);   // cg_hconv_supported without the grid-size policy

bool cg_fast_wgrad_supported(const cgConvGeom* g, const void* in, const void* gate_in,
                             float slope_in, const void* gate_dy);
size_t cg_fast_wgrad_workspace_bytes(const cgConvGeom* g);
void cg_fast_wgrad_launch(const cgConvGeom* g, const void* in, const void* gate_in, const void* dy,
                          float* dw, int accumulate, float* dbias, void* ws, hipStream_t st);

// out_a[i] (+)= sum_z part_a[z][i] over n4_a float4 items (and the same for the optional b pair:
// bias-gradient partials), 8 split lanes per output; splits >= 1
// Deferred reductions (cgDeferCtx, include/cgamd.h).  The context is CALLER-owned and arrives as an
// argument of cg_gwgrad_deferred / cg_gwgrad_pooled_deferred; ReduceDeferScope makes it visible to
// the launchers below the entry point for the duration of THAT call on THAT thread (a thread-local
// pointer restored on exit: nothing outlives the call, the library keeps no process-wide state).
class ReduceDeferScope {
 public:
  explicit ReduceDeferScope(cgDeferCtx* ctx);
  ~ReduceDeferScope();
 private:
  cgDeferCtx* old_;
};
// scope in which the split reductions run at once even inside a ReduceDeferScope (call sites that
// reuse the workspace the partials sit in before the caller could flush)
class ReduceDeferSuspend {
 public:
  explicit ReduceDeferSuspend(bool on);
  ~ReduceDeferSuspend();
 private:
  bool on_;
};
void cg_split_reduce4_pair(const float* part_a, int64_t n4_a, float* out_a, const float* part_b,
                           int64_t n4_b, float* out_b, int splits, int accumulate,
                           hipStream_t st);

// halo-staged 3x3 weight gradient (cg_conv_halo.hip)
bool cg_hwgrad_supported(const cgConvGeom* g, const void* in, const void* gate_in, float slope_in,
                         const void* gate_dy);
size_t cg_hwgrad_workspace_bytes(const cgConvGeom* g);
void cg_hwgrad_launch(const cgConvGeom* g, const void* in, const void* gate_in, const void* dy,
                      float* dw, int accumulate, float* dbias, void* ws, hipStream_t st);

// RGB-input 3x3 convolutions with the input window staged in LDS (cg_conv_halo.hip)
bool cg_wstem_conv_supported(const cgConvGeom* g, const void* in, const void* out,
                             const void* gate_in, float slope_in, const void* gate_out,
                             const void* residual);
void cg_wstem_conv_launch(const cgConvGeom* g, const void* in, const void* bt, void* out,
                          int out_is_f32, const float* bias, const void* gate_in,
                          const void* gate_out, float slope_out, hipStream_t st);
bool cg_wstem_wgrad_supported(const cgConvGeom* g, const void* in, const void* gate_in,
                              float slope_in, const void* gate_dy);
size_t cg_wstem_wgrad_workspace_bytes(const cgConvGeom* g);
void cg_wstem_wgrad_launch(const cgConvGeom* g, const void* in, const void* gate_in,
                           const void* dy, int want_bias, void* ws, int* splits_out,
                           hipStream_t st);

// forms with a 2x2 average pooling fused behind the convolution (pooled output / pooled-resolution dy)
void cg_hwgrad_launch_pooled(const cgConvGeom* g, const void* in, const void* gate_in,
                             const void* dy, int dy_pooled, float* dw, int accumulate,
                             float* dbias, void* ws, hipStream_t st);
void cg_wstem_conv_launch_pool(const cgConvGeom* g, const void* in, const void* bt, void* out,
                               int out_is_f32, const float* bias, const void* gate_in,
                               const void* gate_out, float slope_out, int pool, hipStream_t st);
void cg_wstem_wgrad_launch_pooled(const cgConvGeom* g, const void* in, const void* gate_in,
                                  const void* dy, int dy_pooled, int want_bias, void* ws,
                                  int* splits_out, hipStream_t st);
// strided partial reduce of the stem weight gradients (cg_conv_fast.hip)
void cg_stem_partial_reduce(const cgConvGeom* g, const void* ws, int splits, float* dw,
                            float* dbias, int accumulate, hipStream_t st);

// image-like inputs (Ci <= 4): im2col-in-LDS stem kernels
bool cg_stem_conv_supported(const cgConvGeom* g, const void* in, const void* out,
                            const void* gate_in, float slope_in, const void* gate_out,
                            const void* residual);
void cg_stem_conv_launch(const cgConvGeom* g, const void* in, const void* bt, void* out,
                         int out_is_f32, const float* bias, const void* gate_in,
                         const void* gate_out, float slope_out, hipStream_t st);
bool cg_stem_wgrad_supported(const cgConvGeom* g, const void* in, const void* gate_in,
                             float slope_in, const void* gate_dy);
size_t cg_stem_wgrad_workspace_bytes(const cgConvGeom* g);
void cg_stem_wgrad_launch(const cgConvGeom* g, const void* in, const void* gate_in, const void* dy,
                          float* dw, int accumulate, float* dbias, void* ws, hipStream_t st);

// narrow outputs (Co <= 4): weight gradient through the adjoint geometry (dbias not included)
bool cg_narrow_wgrad_supported(const cgConvGeom* g, const void* gate_in, const void* gate_dy);
size_t cg_narrow_wgrad_workspace_bytes(const cgConvGeom* g);
void cg_narrow_wgrad_launch(const cgConvGeom* g, const void* in, const void* dy, float* dw,
                            int accumulate, void* ws, hipStream_t st);
