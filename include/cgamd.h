/*
 * cgamd.h -- C-ABI of libcgamd.so: the MI355X (gfx950) kernels behind compare_gan's
 * G/D forward-backward hot path and its FID/IS eval path.
 *
 * Boundary rules:
 *   - extern "C", plain pointers + sizes, no torch / C++ types in any signature.
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - every call is stream-ordered on `stream` (a hipStream_t passed as void*), re-entrant,
 *     allocation-free and synchronisation-free (hipGraph-capturable); scratch memory is supplied
 *     by the caller (`ws`, sized by the matching *_workspace_bytes query).
 *   - state: none between calls.  What the library keeps is (a) the last error text per thread,
 *     (b) the opt-in cg_prof_* instrumentation of bench.py (process-wide, off by default),
 *     (c) dispatch switches read once from the environment (scripts/README.md).  Deferred
 *     reductions live in a CALLER-owned cgDeferCtx that travels with every call that records.
 *   - return value: 0 = ok; negative = error (never throws, never aborts):
 *       CG_ERR_BAD_ARG (-1) null pointer / non-positive dim / inconsistent geometry,
 *       CG_ERR_UNSUPPORTED (-2) shape class the kernel family does not cover,
 *       CG_ERR_LAUNCH (-3) hipGetLastError() after the launch,
 *       CG_ERR_WORKSPACE (-4) workspace too small.
 *   - activations are NHWC bf16 (raw uint16 storage); master weights are fp32 in the reference's
 *     own layouts (conv HWIO [kh,kw,Ci,Co], linear [in,out]); accumulation is fp32; FID stats fp64.
 *
 * Each entry point cites the reference code (paths relative to the compare_gan tree) it replaces.
 */
#ifndef CGAMD_H_
#define CGAMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CG_OK 0
#define CG_ERR_BAD_ARG (-1)
#define CG_ERR_UNSUPPORTED (-2)
#define CG_ERR_LAUNCH (-3)
#define CG_ERR_WORKSPACE (-4)

typedef void* cgStream; /* hipStream_t */

/* Library/ABI version (bumped when a signature changes; 5 = round 5). */
int cg_abi_version(void);
/* Human-readable description of the last error on this thread ("" if none). */
const char* cg_last_error(void);

/* Optional timing of the convolution kernels with HIP events recorded on the launch stream
 * (bench.py's roofline measurement).  One family per kernel symbol (cg_prof_family_name), so the
 * figures line up with `rocprofv3 --kernel-trace --stats`.  While enabled every launch is bracketed
 * by an event pair; cg_prof_collect synchronises on them and returns the accumulated kernel time,
 * launch count and ALGORITHMIC flops / bytes (useful MACs x 2; minimum bf16 traffic
 * 2*(in + out + weights) bytes).  Not capturable into a hipGraph. */
int cg_prof_family_count(void);
const char* cg_prof_family_name(int family);
int cg_prof_enable(int on);
int cg_prof_reset(void);
int cg_prof_collect(int family, double* total_ms, int64_t* launches, double* flops, double* bytes);

/* Calibration microbenchmarks for the roofline denominators of bench.py (the
 * boxes of a pool differ in sustained clocks, so fractions are also reported against what THIS box
 * reaches): a pure v_mfma_f32_32x32x16_bf16 loop (`blocks` workgroups of 4 waves, `iters` rounds of
 * 8 independent MFMAs per wave on random operands; *flops = what the launch executes) and a float4
 * copy of `bytes` (multiple of 16) bytes.  The caller times them with stream events. */
int cg_calib_mfma_bf16(int blocks, int iters, float* sink, double* flops, cgStream stream);
/* The same loop on all-zero operands (bench.py `calibration.mfma_zero_tflops`): the chip holds a higher
 * clock under it, which tells a power-capped clock from a loop that is not tight. */
int cg_calib_mfma_bf16_zero(int blocks, int iters, float* sink, double* flops, cgStream stream);
int cg_calib_copy(const void* src, void* dst, size_t bytes, cgStream stream);

/* ------------------------------------------------------------------------------------------
 * Generalised convolution geometry.
 *
 * One "gather convolution" primitive covers every dense contraction of the hot path:
 *   out[n,oh,ow,co] = sum_{r,s,ci} IN_v[n, oh*S - pt + r, ow*S - pl + s, ci] * B[(r,s,ci), co]
 * where IN_v is `in` with U-1 zeros inserted between pixels (virtual size Hin*U x Win*U) and
 * zero padding outside.  U=1,S=stride is tf.nn.conv2d(SAME) (architectures/arch_ops.py:559-573);
 * U=2,S=1 is resnet_ops.unpool + conv2d (architectures/resnet_ops.py:35-56,112-134);
 * U=stride,S=1 with flipped/transposed weights is tf.nn.conv2d_transpose
 * (arch_ops.py:579-592) and the data-gradient of a strided conv; kh=kw=1,H=W=1 is
 * arch_ops.linear (arch_ops.py:538-556).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t N, Hin, Win, Ci; /* input  NHWC */
  int32_t Ho, Wo, Co;      /* output NHWC */
  int32_t kh, kw;          /* filter taps */
  int32_t S;               /* output stride over the virtual input */
  int32_t U;               /* zero-insertion factor of the input (1 = none) */
  int32_t pt, pl;          /* top / left zero padding of the virtual input */
} cgConvGeom;

/* Weight preparation: fp32 HWIO master weight -> bf16 MFMA operand image(s).
 *   w        [kh,kw,Ci,Co] fp32
 *   scale    optional device scalar multiplied in (1/sigma of spectral norm,
 *            arch_ops.py:531 `w / norm_value`), may be NULL
 *   bt_fwd   optional [Co][kh*kw*Ci] bf16  (k = (r*kw+s)*Ci+ci contiguous)       -> cg_gconv fwd
 *   bt_bwd   optional [Ci][kh*kw*Co] bf16  with taps flipped (r'=kh-1-r, s'=kw-1-s),
 *            k' = (r'*kw+s')*Co+co contiguous                                    -> data-gradient /
 *            conv2d_transpose form.
 * Buffer sizes: cg_weight_prep_elems(kh, kw, Ci, Co, which) bf16 elements (which = 0: bt_fwd,
 * 1: bt_bwd): the row-major image with K padded to a multiple of 8.
 */
size_t cg_weight_prep_elems(int kh, int kw, int Ci, int Co, int which);
int cg_weight_prep(const float* w, int kh, int kw, int Ci, int Co, const float* scale,
                   void* bt_fwd, void* bt_bwd, cgStream stream);

/* out = d(gate_out) * ( gconv( d(gate_in) * in , bt ) + bias ) + residual
 *   d(g) = g > 0 ? 1 : slope   (elementwise ReLU / leaky-ReLU derivative gates,
 *   arch_ops.py:595-597 lrelu; resnet_ops.py:165,175 tf.nn.relu); gate_in==in gives act(in).
 *   in        [N,Hin,Win,Ci] bf16
 *   bt        [Co][kh*kw*Ci] bf16 (from cg_weight_prep: cg_weight_prep_elems elements)
 *   out       [N,Ho,Wo,Co] bf16, or fp32 when out_is_f32 != 0
 *   bias      [Co] fp32 or NULL;  gate_in like `in` (bf16) or NULL;  gate_out / residual like
 *             `out` (bf16) or NULL.  gate_out == out selects the value itself as the gate, i.e.
 *             out = lrelu_{slope_out}(gconv + bias) + residual (conv + bias + ReLU of the
 *             Inception graph, eval_utils.py:165-175).
 */
int cg_gconv(const cgConvGeom* geom, const void* in, const void* bt, void* out, int out_is_f32,
             const float* bias, const void* gate_in, float slope_in, const void* gate_out,
             float slope_out, const void* residual, cgStream stream);

/* cg_gconv on CHANNEL SLICES: consecutive pixels of `in` are in_ld elements apart and those of `out`
 * out_ld (in_ld >= Ci, out_ld >= Co, both multiples of 8, Co % 8 == 0; bf16 in, bf16 or fp32 out).
 * What it replaces: the tf.concat(axis=3) that ends every Inception block and the reads of a block's
 * sibling 1x1 convolutions (the frozen graph behind eval_utils.py:41-49,165-175) -- a branch writes its
 * channels straight into the block's output, and the 1x1 convolutions that share an input run as one
 * convolution whose output the next layers read slice by slice.  Same arithmetic as cg_gconv with
 * gate_in = residual = NULL and the ReLU of gate_out = out (slope 0) applied to the output channels
 * [0, relu_cols) only (relu_cols a multiple of 8: Co = everywhere, 0 = nowhere; in between for a merged
 * convolution some of whose columns are finished by another kernel -- the 1x1 convolution behind a
 * block's 3x3 average pooling commutes with it and runs in front, cg_pool2d_ld adds bias and ReLU).
 * cg_gconv_ld_supported: 1 when the geometry has a kernel with this addressing (the MFMA one-tap
 * kernel: Ci % 32 == 0), else 0 -- the caller then uses cg_gconv on dense tensors. */
int cg_gconv_ld_supported(const cgConvGeom* geom, int in_ld, int out_ld);
int cg_gconv_ld(const cgConvGeom* geom, const void* in, int in_ld, const void* bt, void* out,
                int out_ld, int out_is_f32, const float* bias, int relu_cols, cgStream stream);

/* Batch-norm fusion around cg_gconv for forward passes that keep no autograd graph (the generator
 * forward of every discriminator sub-step, modular_gan.py:465-467): the producer convolution emits
 * the per-channel sums of the values it stores, the consumer convolution normalises its input in
 * LDS -- the normalised activation (arch_ops.py:289-313 + resnet_ops.py:165,175 ReLU) never goes
 * through HBM.
 *   bn_mean / bn_var [Ci] (NULL = no prologue), bn_gamma / bn_beta [Ci] or [N,Ci] when
 *   bn_per_sample (conditional BN, arch_ops.py:423-445), NULL = 1 / 0:
 *     in' = relu(((in - mean) * rsqrt(var + eps)) * gamma + beta), zero padding applied to in'.
 *   stats_out (NULL = none): [rows][2*Co] fp32, rows = cg_gconv_fused_rows(geom); row r holds
 *     sum(out) in [0,Co) and sum(out^2) in [Co,2Co) over a disjoint part of the output pixels (of
 *     the values as stored, i.e. after the bf16 rounding); cg_bn_finalize reduces them.
 * cg_gconv_fused_rows returns 0 when the geometry is not covered by the fused kernel (unit-stride
 * <= 3x3 filters on >= 16x16 maps, Ci % 32 == 0): call cg_gconv and the cg_bn_* kernels then.
 * Fewer than 8 output channels (the RGB convolution that ends a generator, resnet_cifar.py:108-111):
 * the prologue only -- stats_out, pool_out, in_up, residual and a gate tensor are refused. */
typedef struct {
  const float* bn_mean;
  const float* bn_var;
  const float* bn_gamma;
  const float* bn_beta;
  float bn_eps;
  int32_t bn_per_sample;
  float* stats_out;
  /* 2x2 average pooling fused around the convolution (resnet_ops.py:131-133: conv -> tf.nn.pool AVG):
   *   pool_out != 0: out is [N, Ho/2, Wo/2, Co] = avgpool2(conv + bias) + residual (residual at the
   *                  pooled resolution; no output gate);
   *   in_up != 0   : `in` is [N, Hin/2, Win/2, Ci] and stands for its nearest-neighbour 2x
   *                  up-sampling -- with out_scale = 1/4 this is the gradient of the pooling feeding
   *                  the data-gradient convolution.  out_scale multiplies the convolution sum
   *                  (0 = 1). */
  int32_t pool_out;
  int32_t in_up;
  float out_scale;
  /* > 0: bn_mean / bn_var are [N / bn_stat_group][Ci] -- one set of statistics per bn_stat_group
   * consecutive samples (cg_bn_stats_groups: several network calls batched into one); 0: [Ci]. */
  int32_t bn_stat_group;
} cgConvFusion;
int cg_gconv_fused_rows(const cgConvGeom* geom);
/* 1 when cg_gconv_fused covers `geom` with the batch-norm prologue alone (bn_mean set; no stats_out,
 * pool_out, in_up, residual, gate tensor): every geometry with cg_gconv_fused_rows > 0, and 3x3
 * convolutions to fewer than 8 output channels on the same maps. */
int cg_gconv_fused_prologue_supported(const cgConvGeom* geom);
/* Layout of those rows for cg_bn_finalize: [phases][rows / phases]; U*U when the output phases of a
 * zero-inserted input run as separate workgroups, 1 when one workgroup covers all of them (0 when
 * the geometry is not covered). */
int cg_gconv_fused_phases(const cgConvGeom* geom);
int cg_gconv_fused(const cgConvGeom* geom, const void* in, const void* bt, void* out,
                   int out_is_f32, const float* bias, const void* gate_in, float slope_in,
                   const void* gate_out, float slope_out, const void* residual,
                   const cgConvFusion* fusion, cgStream stream);
/* 1 when cg_gconv_fused (pool_out) and cg_gwgrad_pooled cover `geom` (also RGB-input 3x3
 * convolutions), else 0. */
int cg_gconv_pool_supported(const cgConvGeom* geom);
/* cg_gwgrad for a convolution whose output was 2x2 average-pooled: dy_pooled is
 * [N, Ho/2, Wo/2, Co] (the gradient w.r.t. the pooled output); workspace as cg_gwgrad. */
int cg_gwgrad_pooled(const cgConvGeom* geom, const void* in, const void* gate_in, float slope_in,
                     const void* dy_pooled, float* dw, int accumulate, float* dbias, void* ws,
                     size_t ws_bytes, cgStream stream);
/* The same with the split reductions recorded in `defer` (see "Deferred reductions" below; NULL =
 * cg_gwgrad_pooled). */
struct cgDeferCtx;
int cg_gwgrad_pooled_deferred(const cgConvGeom* geom, const void* in, const void* gate_in,
                              float slope_in, const void* dy_pooled, float* dw, int accumulate,
                              float* dbias, void* ws, size_t ws_bytes, cgStream stream,
                              struct cgDeferCtx* defer);

/* Deferred reductions.  Weight-gradient kernels that split their pixels leave per-split partials in
 * the workspace and a small fixed-order reduction behind them (60 launches per ResNet-CIFAR train
 * step).  cg_gwgrad_deferred / cg_gwgrad_pooled_deferred (below) launch the partial-sum kernels but
 * only RECORD those reductions in a CALLER-OWNED context; cg_defer_flush(ctx, stream) runs every
 * recorded one in one launch per kernel form (same summation order per output as the separate
 * launches: results are bit-identical) and leaves the context empty and reusable.  Until the flush
 * the caller keeps every workspace and output of the recorded calls alive, unread, and on `stream`.
 * The library holds no process-wide state: one context per stream / replica / backward pass, used
 * by one thread at a time; contexts of different replicas never see each other's reductions.
 * cg_defer_abort() drops the recorded reductions (error paths); cg_defer_pending() counts them.
 * tf.gradients hands the optimiser all kernel gradients at once (modular_gan.py:480-483,494-497):
 * nothing in the data-gradient chain reads them earlier. */
typedef struct cgDeferCtx cgDeferCtx;
cgDeferCtx* cg_defer_create(void);              /* NULL when out of memory */
void cg_defer_destroy(cgDeferCtx* ctx);         /* recorded reductions are dropped; NULL is fine */
int cg_defer_pending(cgDeferCtx* ctx);
int cg_defer_flush(cgDeferCtx* ctx, cgStream stream);
int cg_defer_abort(cgDeferCtx* ctx);
/* mean / var (and the moving averages, as cg_bn_stats) from `rows` rows of partial sums
 * [rows][2*C] over `count` values per channel. */
int cg_bn_finalize(const float* partials, int rows, int C, int64_t count, float* mean, float* var,
                   float* moving_mean, float* moving_var, float decay, cgStream stream);
/* The same for `groups` statistics groups of consecutive samples (see cg_bn_stats_groups): the
 * rows are [phases][rows / phases] (phases = U*U of the producing convolution), the rows of one
 * phase split evenly and in order over the groups; `count` values per channel and group;
 * mean / var [groups][C]. */
int cg_bn_finalize_groups(const float* partials, int rows, int C, int64_t count, int groups,
                          int phases, float* mean, float* var, float* moving_mean,
                          float* moving_var, float decay, cgStream stream);

/* Weight gradient of the same primitive (tf.gradients of arch_ops.conv2d / deconv2d / linear
 * w.r.t. the kernel):
 *   dw[(r,s,ci),co] (+)= sum_{n,oh,ow} d(gate_in)*IN_v[n,oh*S-pt+r,ow*S-pl+s,ci] * d(gate_dy)*dy[n,oh,ow,co]
 *   dw fp32 [kh,kw,Ci,Co]; accumulate != 0 adds into dw instead of overwriting.
 *   dbias optional fp32 [Co] = column sums of the gated dy (NULL to skip).
 *   ws: workspace of at least cg_gwgrad_workspace_bytes(geom) bytes.
 */
size_t cg_gwgrad_workspace_bytes(const cgConvGeom* geom);
int cg_gwgrad(const cgConvGeom* geom, const void* in, const void* gate_in, float slope_in,
              const void* dy, const void* gate_dy, float slope_dy, float* dw, int accumulate,
              float* dbias, void* ws, size_t ws_bytes, cgStream stream);
/* The same with the split reductions recorded in `defer` (NULL = cg_gwgrad): dw / dbias are valid
 * after cg_defer_flush(defer, stream); ws must stay alive until then. */
int cg_gwgrad_deferred(const cgConvGeom* geom, const void* in, const void* gate_in, float slope_in,
                       const void* dy, const void* gate_dy, float slope_dy, float* dw,
                       int accumulate, float* dbias, void* ws, size_t ws_bytes, cgStream stream,
                       cgDeferCtx* defer);
/* Several weight gradients in one call: tf.gradients(loss, var_list) hands all kernel gradients
 * of a network to the optimiser at once (modular_gan.py:480-483,494-497), so they need not be
 * computed inside the data-gradient chain.  `items` is a HOST array (device pointers by value in
 * the kernel arguments: hipGraph-capturable).  The 3x3 layers on small maps (N*H*W <= 8192 pixels,
 * channel counts multiples of 64; resnet_cifar.py:119-167 blocks B3/B4, resnet5.py:99-145 blocks
 * B4/B5) share launches: none of them fills the chip alone without splitting its pixel sum into
 * fp32 partials.  Every other item runs as cg_gwgrad with the shared workspace
 * (ws_bytes >= max over the items of cg_gwgrad_workspace_bytes).  gate_in: NULL or == in (ReLU
 * self-gate, slope_in 0) for the grouped form, anything cg_gwgrad accepts otherwise. */
typedef struct {
  cgConvGeom geom;
  const void* in;
  const void* gate_in;
  float slope_in;
  int32_t accumulate;
  const void* dy;
  float* dw;
  float* dbias;
} cgWgradItem;
int cg_gwgrad_multi(const cgWgradItem* items_host, int n, void* ws, size_t ws_bytes,
                    cgStream stream);
/* 1 when cg_gwgrad_multi would run `geom` in a shared launch (worth deferring), else 0. */
int cg_gwgrad_groupable(const cgConvGeom* geom);

/* ------------------------------------------------------------------------------------------
 * Spectral normalisation (arch_ops.py:453-535): one power-iteration round, sigma, u update.
 *   w [K,Co] fp32 (HWIO flattened); mode 0 = "left" (u is [K]), 1 = "right" (u is [Co]).
 *   left : v = l2n(w^T u), u' = l2n(w v), sigma = u'^T w v
 *   right: v = l2n(w u^T)  ([K]), u' = l2n(v^T w) ([Co]), sigma = v^T w u'^T
 *   l2n(x) = x * rsqrt(max(sum x^2, eps)).
 * Outputs: u_out (may alias u_in), v_out (the non-persisted vector), sigma (1 float),
 *          inv_sigma (1 float).  ws >= cg_spectral_norm_workspace_bytes(K,Co).
 * ------------------------------------------------------------------------------------------ */
size_t cg_spectral_norm_workspace_bytes(int K, int Co);
int cg_spectral_norm(const float* w, int K, int Co, int mode, float eps, const float* u_in,
                     float* u_out, float* v_out, float* sigma, float* inv_sigma, void* ws,
                     size_t ws_bytes, cgStream stream);
/* Gradient through w_bar = w / sigma with u,v constants (arch_ops.py:519-531):
 *   dw = (dwbar - <dwbar, w>/sigma * a b^T) / sigma   with (a,b) = (u',v) left, (v,u') right,
 *   a is [K], b is [Co].  dw may alias dwbar.  ws >= cg_sn_backward_workspace_bytes(K,Co). */
size_t cg_sn_backward_workspace_bytes(int K, int Co);
int cg_sn_backward(const float* dwbar, const float* w, int K, int Co, const float* a_k,
                   const float* b_co, const float* sigma, float* dw, void* ws, size_t ws_bytes,
                   cgStream stream);

/* ------------------------------------------------------------------------------------------
 * Multi-tensor forms: every spectrally-normalised weight of a network call in five launches
 * (instead of five per weight), the backward of all of them in two, and the MFMA operand images of
 * all convolution / linear weights in two.  `items` is a HOST array; the per-tensor device pointers
 * travel by value in the kernel arguments (chunks of CG_MULTI_MAX), so nothing is uploaded and the
 * launches stay hipGraph-capturable although gradient tensors move between calls.
 * ------------------------------------------------------------------------------------------ */
#define CG_MULTI_MAX 32
typedef struct {
  const float* w;   /* [K, Co] fp32 */
  float* u;         /* persisted power-iteration vector, updated in place ([K] mode 0, [Co] mode 1) */
  float* u_out;     /* per-call copy of the updated u (kept for the backward) */
  float* v_out;     /* the other singular-vector estimate ([Co] mode 0, [K] mode 1) */
  float* sigma;     /* [2]: sigma, 1 / sigma */
  float* wbar;      /* [K, Co] w / sigma, or NULL */
  float* ws;        /* >= cg_spectral_norm_multi_workspace_floats(K, Co) floats */
  int32_t K, Co, mode;
} cgSNItem;
size_t cg_spectral_norm_multi_workspace_floats(int K, int Co);
int cg_spectral_norm_multi(const cgSNItem* items_host, int n, float eps, cgStream stream);
typedef struct {
  const float* dwbar; /* gradient w.r.t. w / sigma */
  const float* w;
  const float* a_k;   /* [K]  (u' for mode 0, v for mode 1) */
  const float* b_co;  /* [Co] (v for mode 0, u' for mode 1) */
  const float* sigma; /* [2] */
  float* dw;          /* may alias dwbar */
  float* ws;          /* >= cg_sn_backward_multi_workspace_floats(K, Co) floats */
  int32_t K, Co;
} cgSNBwdItem;
size_t cg_sn_backward_multi_workspace_floats(int K, int Co);
int cg_sn_backward_multi(const cgSNBwdItem* items_host, int n, cgStream stream);
typedef struct {
  const float* w;  /* [T, Ci, Co] fp32 (T = kh*kw) */
  void* bt_fwd;    /* [Co][Kp] bf16 or NULL  (layouts and sizes of cg_weight_prep) */
  void* bt_bwd;    /* [Ci][Kbp] bf16 or NULL */
  int32_t T, Ci, Co;
} cgPrepItem;
int cg_weight_prep_multi(const cgPrepItem* items_host, int n, cgStream stream);

/* Gradient bucket for the data-parallel all-reduce (CrossShardOptimizer, modular_gan.py:606-616):
 * copies n fp32 tensors (host arrays of device pointers / element counts) back to back into
 * `flat`.  Pointers by value in the kernel arguments: capturable although the gradient tensors
 * move from step to step. */
int cg_flatten_multi(const float* const* srcs_host, const int64_t* sizes_host, int n, float* flat,
                     cgStream stream);

/* out = x * (*scale_dev) * scale_host on fp32 (w_bar = w * (1/sigma), arch_ops.py:531; also loss
 * gradient scaling).  scale_dev may be NULL (= 1). out may alias x. */
int cg_scale_f32(const float* x, const float* scale_dev, float scale_host, float* out, int64_t n,
                 cgStream stream);

/* ------------------------------------------------------------------------------------------
 * Batch normalisation family (arch_ops.py:194-319 standardize_batch, :327-367 batch_norm,
 * :423-445 conditional_batch_norm).  x is [N, HW, C] bf16 (2-D inputs use HW=1).
 * ------------------------------------------------------------------------------------------ */
/* mean[c] = sum x / n ; var[c] = sum x^2 / n - mean^2 (fp32, arch_ops.py:294-297).
 * moving_mean / moving_var (both or neither, may be NULL): the moving-average update
 * m <- m - (1-decay)(m - batch) of arch_ops.py:105-114 folded into the same launch.
 * ws >= cg_bn_stats_workspace_bytes(N*HW, C). */
size_t cg_bn_stats_workspace_bytes(int64_t rows, int C);
int cg_bn_stats(const void* x, int64_t rows, int C, float* mean, float* var, float* moving_mean,
                float* moving_var, float decay, void* ws, size_t ws_bytes, cgStream stream);
/* Statistics of `groups` consecutive blocks of rows / groups rows each, mean / var [groups][C]:
 * the batch norm of several network calls that were batched into one call.  The reference runs
 * the generator once per sub-step on the same weights (modular_gan.py:464-467); one batched call
 * with per-call statistics is the same arithmetic.  The moving averages receive the groups'
 * updates in order (arch_ops.py:105-114 once per call).  ws >= cg_bn_stats_groups_workspace_bytes. */
size_t cg_bn_stats_groups_workspace_bytes(int64_t rows, int C, int groups);
int cg_bn_stats_groups(const void* x, int64_t rows, int C, int groups, float* mean, float* var,
                       float* moving_mean, float* moving_var, float decay, void* ws,
                       size_t ws_bytes, cgStream stream);
/* y = act( (x - mean) * rsqrt(var + eps) * gamma + beta ), act = relu if relu != 0.
 * gamma/beta: fp32 [C] (per_sample == 0) or [N,C] (per_sample != 0, conditional BN), NULL = 1 / 0.
 * y bf16 same shape as x. */
int cg_bn_apply(const void* x, int N, int HW, int C, const float* mean, const float* var,
                float eps, const float* gamma, const float* beta, int per_sample, int relu,
                void* y, cgStream stream);
/* cg_bn_apply with mean / var [N / stat_group][C] (stat_group > 0; N % stat_group == 0). */
int cg_bn_apply_groups(const void* x, int N, int HW, int C, const float* mean, const float* var,
                       float eps, const float* gamma, const float* beta, int per_sample,
                       int stat_group, int relu, void* y, cgStream stream);
/* Backward of cg_bn_apply, in two stream-ordered stages so that data-parallel sync-BN can
 * all-reduce the per-channel means in between (tpu_ops.py:94-125 applied to the backward sums):
 *   reduce: dz = dy * (y > 0) if relu;  dbeta = sum dz;  dgamma = sum dz * xhat (per [C] or [N,C]);
 *           m12[0:C] = mean_n(g*dz), m12[C:2C] = mean_n(g*dz*xhat)   (local batch means)
 *   apply : dx = rstd * (g*dz - m12[0] - xhat * m12[1])   when batch_stats != 0
 *           dx = rstd * g * dz                              when batch_stats == 0 (eval; m12 unused)
 * dgamma/dbeta may be NULL.  ws >= cg_bn_backward_workspace_bytes(N, HW, C). */
size_t cg_bn_backward_workspace_bytes(int N, int HW, int C);
int cg_bn_backward_reduce(const void* x, const void* y, const void* dy, int N, int HW, int C,
                          const float* mean, const float* var, float eps, const float* gamma,
                          int per_sample, int relu, float* dgamma, float* dbeta, float* m12,
                          void* ws, size_t ws_bytes, cgStream stream);
int cg_bn_backward_apply(const void* x, const void* y, const void* dy, int N, int HW, int C,
                         const float* mean, const float* var, float eps, const float* gamma,
                         int per_sample, int relu, int batch_stats, const float* m12, void* dx,
                         cgStream stream);
/* In-place helpers around the sync-BN all-reduce: to_variance == 0: second <- var + mean^2;
 * to_variance == 1: mean *= scale; second *= scale; second <- second - mean^2. */
int cg_bn_moments_convert(float* mean, float* second, int C, int to_variance, float scale,
                          cgStream stream);
/* Accumulator statistics for inference (arch_ops.py:122-191, filled by eval_gan_lib.py:65-92):
 *   cg_bn_accumulate          accu_mean += mean; accu_var += var; *accu_counter += 1
 *   cg_bn_accumulated_moments mean = accu_mean / *accu_counter; var = accu_var / *accu_counter
 * (accu_counter is a device scalar initialised to 1e-12, arch_ops.py:166.) */
int cg_bn_accumulate(float* accu_mean, float* accu_var, float* accu_counter, const float* mean,
                     const float* var, int C, cgStream stream);
int cg_bn_accumulated_moments(const float* accu_mean, const float* accu_var,
                              const float* accu_counter, float* mean, float* var, int C,
                              cgStream stream);
/* Moving averages m <- m - (1-decay) * (m - batch)  (arch_ops.py:105-114), both vectors [C]. */
int cg_bn_update_moving(float* moving_mean, float* moving_var, const float* mean,
                        const float* var, int C, float decay, cgStream stream);

/* ------------------------------------------------------------------------------------------
 * Element-wise / pooling glue of the G and D graphs.
 * ------------------------------------------------------------------------------------------ */
/* y = x > 0 ? x : slope*x  (arch_ops.py:595-597; slope 0 = tf.nn.relu), bf16, n elements. */
int cg_lrelu(const void* x, float slope, void* y, int64_t n, cgStream stream);
/* dx = dy * (x > 0 ? 1 : slope) */
int cg_lrelu_bwd(const void* x, const void* dy, float slope, void* dx, int64_t n, cgStream stream);
/* out = a*alpha + b*beta (bf16), b may be NULL. */
int cg_axpby(const void* a, float alpha, const void* b, float beta, void* out, int64_t n,
             cgStream stream);
/* out = alpha*a + beta*b on fp32 (logit + projection term, resnet_biggan.py:423;
 * d_loss += lambda * penalty, modular_gan.py:670).  b may be NULL. */
/* out = a + b + c (+ d, may be NULL): bf16 tensors of n elements, summed in fp32 with one rounding.
 * Replaces autograd's accumulation of the gradient of a tensor with three or four consumers -- the
 * input of the self-attention block, compare_gan/architectures/arch_ops.py:709-758 (theta, phi, g
 * projections and the residual path). */
int cg_sum4(const void* a, const void* b, const void* c, const void* d, void* out, int64_t n,
            cgStream stream);
int cg_axpby_f32(const float* a, float alpha, const float* b, float beta, float* out, int64_t n,
                 cgStream stream);
/* HOST utility (no device work): CRC32C (Castagnoli) of n bytes, continuing from `seed` (0 for a
 * fresh checksum) -- TFRecord payloads of the TFDS shards (datasets.py:430-532) and the blocks /
 * tensors of TF-1 tensor bundles (modular_gan.py:266-285 restores them). */
uint32_t cg_host_crc32c(const void* data, size_t n, uint32_t seed);
/* out = x * scale on fp32 and *nan_count += (number of NaNs in x): the sampled images of an
 * evaluation set are scaled to [0, 255] and tested with np.isnan (eval_utils.py:144-162) -- one
 * pass per generator batch, written straight into the batch's slot of the set.  nan_count: int32 on
 * the device, zeroed by the caller; out may alias x. */
int cg_scale_count_nan_f32(const float* x, float scale, float* out, int64_t n, int32_t* nan_count,
                           cgStream stream);
/* out = x + (*sigma) * o on bf16 with a device-resident trainable scalar
 * (arch_ops.py:755-758 `x + sigma * attn_g`); x may be NULL (out = sigma * o). */
int cg_axpy_dev(const void* x, const void* o, const float* sigma, void* out, int64_t n,
                cgStream stream);
/* <a, b> of two bf16 arrays into one fp32 (gradient of that scalar). */
size_t cg_dot_bf16_workspace_bytes(int64_t n);
int cg_dot_bf16(const void* a, const void* b, int64_t n, float* out, void* ws, size_t ws_bytes,
                cgStream stream);
/* 2x2 average pooling, stride 2 (resnet_ops.py:131-133), x [N,H,W,C] -> y [N,H/2,W/2,C]. */
int cg_avgpool2(const void* x, int N, int H, int W, int C, void* y, cgStream stream);
int cg_avgpool2_bwd(const void* dy, int N, int H, int W, int C, void* dx, cgStream stream);
/* Zero-insertion 2x upsampling without a convolution (resnet_ops.py:35-56 `unpool`, used bare by
 * the BigGAN-deep generator's shortcut, resnet_biggan_deep.py:102-103): x [N,H,W,C] ->
 * y [N,2H,2W,C], y[n,2i,2j,:] = x[n,i,j,:], zero elsewhere; `residual` [N,2H,2W,C] (may be NULL) is
 * added (the block's `outputs += shortcut`).  Gradient w.r.t. x: dx[n,i,j,:] = dy[n,2i,2j,:]
 * (H, W are the sizes of x in both calls). */
int cg_unpool2(const void* x, const void* residual, int N, int H, int W, int C, void* y,
               cgStream stream);
int cg_unpool2_bwd(const void* dy, int N, int H, int W, int C, void* dx, cgStream stream);
/* 2x2 max pooling stride 2 (arch_ops.py:741,750) and its gradient (routes to first max). */
int cg_maxpool2(const void* x, int N, int H, int W, int C, void* y, cgStream stream);
int cg_maxpool2_bwd(const void* x, const void* dy, int N, int H, int W, int C, void* dx,
                    cgStream stream);
/* Spatial reduction over HW per (n,c): out = scale * sum_hw x * (gate ? gate>0 : 1)
 * (gate == x: resnet_cifar.py:154-156 relu+reduce_mean, resnet_biggan.py:404-405 relu+reduce_sum).
 * x, gate [N,HW,C] bf16 -> out [N,C] bf16. */
int cg_spatial_reduce(const void* x, const void* gate, int N, int HW, int C, float scale,
                      void* out, cgStream stream);
/* dx[n,hw,c] = scale * dout[n,c] * (gate ? gate>0 : 1) */
/* The discriminator's output head, one launch per direction (resnet_cifar.py:154-157,
 * resnet5.py:141-145, resnet_biggan.py:404-407: relu -> reduce_mean / reduce_sum over [1, 2] ->
 * linear(C -> 1), arch_ops.py:538-556):
 *   pooled[n,c] = bf16(scale * sum_hw relu(x[n,hw,c]))     x [N,HW,C] bf16, pooled [N,C] bf16
 *   logit[n]    = sum_c pooled[n,c] * bf16(w[c]) + bias[0]  w [C] fp32 (the [C,1] kernel, after
 *                                                           spectral norm), bias [1] fp32 or NULL
 * Backward: dlogit [N] fp32 (NULL = no gradient through the logits), dpooled_ext [N,C] bf16 or NULL
 * (a gradient arriving through `pooled` itself: projection discriminators read it,
 * resnet_biggan.py:408-415), both added; dx [N,HW,C] bf16; dw [C] / dbias [1] fp32 or NULL.  The
 * bf16 rounding points are those of the separate launches (pooled, d pooled, dx).
 * C % 8 == 0 (cg_pooled_head_supported).  ws >= cg_pooled_head_bwd_workspace_bytes(N, C) when dw. */
/* out[0] = sum_i a[i] * b[i] on fp32, one workgroup, fixed summation order: d(loss)/d(sigma) of a
 * scalar folded into a weight tensor (w_eff = sigma * w: d sigma = <d w_eff, w>; the self-attention
 * block's `x + sigma * conv(attn_g, w)`, arch_ops.py:755-758, runs as conv(attn_g, sigma * w) with
 * x as the convolution's residual).  Meant for weight-sized n. */
int cg_dot_f32(const float* a, const float* b, int64_t n, float* out, cgStream stream);
int cg_pooled_head_supported(int HW, int C);
int cg_pooled_head_fwd(const void* x, int N, int HW, int C, float scale, const float* w,
                       const float* bias, void* pooled, float* logit, cgStream stream);
size_t cg_pooled_head_bwd_workspace_bytes(int N, int C);
int cg_pooled_head_bwd(const void* x, int N, int HW, int C, float scale, const float* w,
                       const float* dlogit, const void* dpooled_ext, const void* pooled, void* dx,
                       float* dw, float* dbias, void* ws, size_t ws_bytes, cgStream stream);
int cg_spatial_reduce_bwd(const void* gate, const void* dout, int N, int HW, int C, float scale,
                          void* dx, cgStream stream);
/* Output heads (resnet_cifar.py:112 sigmoid; resnet_biggan.py:301 / sndcgan.py:74-78
 * (tanh+1)/2; dcgan.py:81 0.5*tanh+0.5).  kind: 0 sigmoid, 1 (tanh+1)/2.
 * x fp32 [n] -> y fp32 [n];  bwd: dx(bf16) = dy(fp32 or bf16) * head'(x). */
int cg_head(const float* x, int kind, float* y, int64_t n, cgStream stream);
int cg_head_bwd(const float* y, int kind, const void* dy, int dy_is_f32, void* dx_bf16, int64_t n,
                cgStream stream);
/* Casts. */
int cg_cast_f32_to_bf16(const float* x, void* y, int64_t n, cgStream stream);
int cg_cast_bf16_to_f32(const void* x, float* y, int64_t n, cgStream stream);
/* y = x * a + b elementwise on fp32 -> bf16 (sndcgan.py:108 `x * 2.0 - 1.0`, plus cast). */
int cg_affine_f32_to_bf16(const float* x, float a, float b, void* y, int64_t n, cgStream stream);
/* Layer normalisation (arch_ops.py:448-450: tf.contrib.layers.layer_norm defaults -- statistics per
 * sample over (H, W, C), gamma / beta per channel, variance_epsilon 1e-12; used when D.layer_norm =
 * True, resnet_ops.py:162-173):
 *   y[n,p,c] = ((x[n,p,c] - mean_n) * rstd_n) * gamma[c] + beta[c],  rstd_n = rsqrt(var_n + eps)
 * x, y, dy, dx bf16 [N, M, C] (M * C % 8 == 0; C = 3 covers the RGB input of a first block); gamma,
 * beta, dgamma, dbeta fp32 [C]; mean, rstd fp32 [N]
 * (written by the forward, read by the backward).  dgamma / dbeta may be NULL.
 * ws >= cg_layer_norm_bwd_workspace_bytes(N, C). */
int cg_layer_norm_fwd(const void* x, int N, int64_t M, int C, const float* gamma, const float* beta,
                      float eps, void* y, float* mean, float* rstd, cgStream stream);
size_t cg_layer_norm_bwd_workspace_bytes(int N, int C);
int cg_layer_norm_bwd(const void* x, const void* dy, const float* mean, const float* rstd,
                      const float* gamma, int N, int64_t M, int C, void* dx, float* dgamma,
                      float* dbeta, void* ws, size_t ws_bytes, cgStream stream);
/* Second order of the same backward (a gradient penalty through D.layer_norm = True,
 * resnet_ops.py:162-173 under penalty_lib.py:59-82): for dx = cg_layer_norm_bwd(x, dy) and an
 * upstream gradient u = dL/d(dx) [N, M, C] bf16 -> d_dy, d_x [N, M, C] bf16 and d_gamma [C] fp32
 * (NULL to skip); ws >= cg_layer_norm_bwd_bwd_workspace_bytes(N, C). */
size_t cg_layer_norm_bwd_bwd_workspace_bytes(int N, int C);
int cg_layer_norm_bwd_bwd(const void* x, const void* dy, const void* u, const float* mean,
                          const float* rstd, const float* gamma, int N, int64_t M, int C,
                          void* d_dy, void* d_x, float* d_gamma, void* ws, size_t ws_bytes,
                          cgStream stream);
/* Column sums of a [rows, C] bf16 matrix into fp32 [C] (bias gradients).
 * ws >= cg_colsum_workspace_bytes(rows, C). */
size_t cg_colsum_workspace_bytes(int64_t rows, int C);
int cg_colsum(const void* x, int64_t rows, int C, float* out, void* ws, size_t ws_bytes,
              cgStream stream);
/* out[b] = sum_c a[b,c]*b[b,c]  (projection discriminator, resnet_biggan.py:423), bf16 in,
 * fp32 out [B]; bwd: da = dout[b]*b, db = dout[b]*a. */
int cg_rowdot(const void* a, const void* b, int B, int C, float* out, cgStream stream);
int cg_rowdot_bwd(const void* a, const void* b, const float* dout, int B, int C, void* da,
                  void* db, cgStream stream);
/* one_hot(labels, K) as bf16 [B,K]  (modular_gan.py:359-363). */
int cg_one_hot(const int32_t* labels, int B, int K, void* out, cgStream stream);

/* ------------------------------------------------------------------------------------------
 * Self-attention core of arch_ops.non_local_block (arch_ops.py:744-753):
 *   attn = softmax(theta phi^T) ; out = attn g,  per image.
 *   theta [B,Lq,Dk], phi [B,Lk,Dk], g [B,Lk,Dv] bf16 -> out [B,Lq,Dv] bf16, lse [B,Lq] fp32.
 * Backward recomputes the probabilities from lse (no [B,Lq,Lk] tensor is materialised).
 * ------------------------------------------------------------------------------------------ */
int cg_attention_fwd(const void* theta, const void* phi, const void* g, int B, int Lq, int Lk,
                     int Dk, int Dv, void* out, float* lse, cgStream stream);
size_t cg_attention_bwd_workspace_bytes(int B, int Lq, int Lk, int Dk, int Dv);
int cg_attention_bwd(const void* theta, const void* phi, const void* g, const void* out,
                     const float* lse, const void* dout, int B, int Lq, int Lk, int Dk, int Dv,
                     void* dtheta, void* dphi, void* dg, void* ws, size_t ws_bytes,
                     cgStream stream);

/* ------------------------------------------------------------------------------------------
 * Losses (gans/loss_lib.py:53-148) and WGAN-GP (gans/penalty_lib.py:59-82).
 * kind: 0 non_saturating, 1 wasserstein, 2 least_squares, 3 hinge.
 * logits fp32 [2B] = real (first B) then fake (last B) (modular_gan.py:657-661).
 * losses[4] = d_loss, d_loss_real, d_loss_fake, g_loss.
 * dlogits_d [2B] = d d_loss / d logits;  dlogits_g [2B] = d g_loss / d logits (real half = 0).
 * ------------------------------------------------------------------------------------------ */
int cg_gan_loss(int kind, const float* logits, int B, float* losses, float* dlogits_d,
                float* dlogits_g, cgStream stream);
/* Rotation loss of the self-supervised GAN (gans/ssgan.py:186-203):
 *   *loss = -mean_i log(softmax(logits[i, :])[labels[i]] + eps)  (the reference adds eps = 1e-10
 *   INSIDE the log), dlogits [n,k] = d loss / d logits.  logits fp32 [n,k], labels int32 [n]. */
int cg_softmax_xent_eps(const float* logits, const int32_t* labels, int n, int k, float eps,
                        float* loss, float* dlogits, cgStream stream);
/* S3GAN label handling (gans/s3gan.py:118-160): is_label_available[i] = sum_k y[i,k] > 0.5;
 * y_out[i,:] = y[i,:] where a label is available, otherwise the predictor's label --
 * softmax(aux_logits[i,:]) (soft != 0) or one_hot(argmax) -- y / y_out bf16 [n,k], aux_logits fp32
 * [n,k] or NULL (no predictor: y_out = y). */
int cg_s3gan_labels(const float* aux_logits, const void* y, int n, int k, int soft, void* y_out,
                    float* is_label_available, cgStream stream);
/* tf.losses.softmax_cross_entropy(labels, logits, weights) with SUM_BY_NONZERO_WEIGHTS
 * (gans/s3gan.py:311-313): *loss = sum_i w_i CE_i / #{w_i != 0}; dlogits [n,k].  labels bf16 [n,k]
 * (one-hot or soft), logits fp32 [n,k], weights fp32 [n]. */
int cg_softmax_xent_weighted(const float* logits, const void* labels, const float* weights, int n,
                             int k, float* loss, float* dlogits, cgStream stream);
/* interpolates = x + alpha[b] * (x_fake - x)   (penalty_lib.py:72-73), fp32 in, bf16 out. */
int cg_interpolate(const float* x, const float* x_fake, const float* alpha, int B, int64_t per,
                   void* out_bf16, cgStream stream);
/* slopes[b] = sqrt(1e-4 + sum g^2); penalty = mean((slopes-1)^2)  (penalty_lib.py:77-81).
 * g fp32 [B, per]. */
int cg_gradient_penalty(const float* g, int B, int64_t per, float* slopes, float* penalty,
                        cgStream stream);
/* dg = upstream * 2/B * (slopes[b]-1)/slopes[b] * g   -> bf16 */
int cg_gradient_penalty_bwd(const float* g, const float* slopes, const float* upstream, int B,
                            int64_t per, void* dg_bf16, cgStream stream);

/* sums[0] = sum x, sums[1] = sum x^2 over n fp32 values (two-stage, deterministic): the global
 * variance of DRAGAN (penalty_lib.py:46 tf.nn.moments over all axes) and tf.nn.l2_loss of a kernel
 * (penalty_lib.py:99-102).  ws >= cg_moments_workspace_bytes(). */
size_t cg_moments_workspace_bytes(void);
int cg_moments_f32(const float* x, int64_t n, float* sums, void* ws, size_t ws_bytes,
                   cgStream stream);
/* x_noisy = clip(x + std * (u - 0.5), 0, 1) * a + b as bf16 (penalty_lib.py:47-49; a, b: the
 * discriminator's input affine), std = sqrt(sums[1]/n - (sums[0]/n)^2) from cg_moments_f32 of x. */
int cg_dragan_perturb(const float* x, const float* u, const float* sums, int64_t n, float a,
                      float b, void* out_bf16, cgStream stream);

/* ------------------------------------------------------------------------------------------
 * Optimiser: tf.train.AdamOptimizer (TF1 epsilon placement) + tf.train.ExponentialMovingAverage
 * (modular_gan.py:480-483,494-508) over a list of tensors in ONE launch.
 *   lr_t = lr * sqrt(1-beta2^t) / (1-beta1^t);  m,v update;  p -= lr_t * m / (sqrt(v) + eps)
 *   ema (optional): s <- s - (1-d)(s - p), d = ema_decay * [t_gen >= ema_start]
 * table: device array of cgAdamEntry (one per tensor); step: device int64 counter holding the
 * number of updates already applied (the kernel uses t = *step + 1); it is NOT modified here.
 * grad_scale multiplies every gradient (1/world_size for summed data-parallel gradients).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  float* param;
  const float* grad;
  float* m;
  float* v;
  float* ema; /* NULL if this tensor has no shadow */
  int64_t n;
  int64_t chunk_begin; /* prefix sum of ceil(n / CG_ADAM_CHUNK) over previous entries */
} cgAdamEntry;
#define CG_ADAM_CHUNK 16384
int cg_adam_multi(const cgAdamEntry* table, int n_entries, int64_t total_chunks, float lr,
                  float beta1, float beta2, float eps, float grad_scale, const int64_t* step,
                  float ema_decay, int64_t ema_start_step, cgStream stream);
/* counter += inc (device int64 step counters: global_step, global_step_disc). */
int cg_counter_add(int64_t* counter, int64_t inc, cgStream stream);
/* Gather a list of fp32 tensors into / scatter out of one flat buffer (gradient buckets for the
 * RCCL all-reduce).  table is the same cgAdamEntry array (uses .grad/.n/.chunk_begin), offsets in
 * the flat buffer are chunk_begin * CG_ADAM_CHUNK-independent: flat_offsets[i] device int64. */
int cg_multi_gather(const cgAdamEntry* table, const int64_t* flat_offsets, int n_entries,
                    int64_t total_chunks, float* flat, cgStream stream);
int cg_multi_scatter(const cgAdamEntry* table, const int64_t* flat_offsets, int n_entries,
                     int64_t total_chunks, const float* flat, cgStream stream);

/* ------------------------------------------------------------------------------------------
 * Stateless counter-based RNG (tpu/tpu_random.py:54-154 semantics: reproducible, distinct per
 * op / step / replica).  Philox4x32-10 keyed by (seed, op_id), counter = (*step_ptr, stream_id,
 * element index).  step_ptr may be NULL (uses 0).
 *   kind 0: uniform [lo,hi)   kind 1: normal(mean=lo, stddev=hi)   -> fp32 out[n]
 * ------------------------------------------------------------------------------------------ */
int cg_random(int kind, float lo, float hi, uint64_t seed, uint32_t op_id, uint32_t stream_id,
              const int64_t* step_ptr, float* out, int64_t n, cgStream stream);
/* Uniform int32 labels in [0,K) (modular_gan.py:386-391). */
int cg_random_labels(int K, uint64_t seed, uint32_t op_id, uint32_t stream_id,
                     const int64_t* step_ptr, int32_t* out, int64_t n, cgStream stream);

/* ------------------------------------------------------------------------------------------
 * FID / Inception-score statistics (metrics/fid_score.py:44-75, metrics/inception_score.py:39-48;
 * the arithmetic the reference delegates to tensorflow_gan, restated in SURVEY section 8c).
 * ------------------------------------------------------------------------------------------ */
/* mean[d] (fp64) and unbiased covariance cov[d,d] (fp64, divide by n-1) of x [n,d] fp32.
 * ws >= cg_mean_cov_workspace_bytes(n, d). */
size_t cg_mean_cov_workspace_bytes(int64_t n, int d);
int cg_mean_cov_f64(const float* x, int64_t n, int d, double* mean, double* cov, void* ws,
                    size_t ws_bytes, cgStream stream);
/* C = op(A) * op(B), fp64 row-major, A [m,k] (or [k,m] if ta), B [k,n] (or [n,k] if tb). */
int cg_gemm_f64(const double* a, const double* b, double* c, int m, int n, int k, int ta, int tb,
                cgStream stream);
/* C = alpha * op(A) * op(B) + beta_eye * I, and the pieces of the inverse-free Newton-Schulz
 * iteration for the symmetric square root of a well-conditioned covariance (metrics/fid_score.py
 * replaces tfgan's SVD-based _symmetric_matrix_square_root, fid_score.py:49-51 of the reference, by
 * GEMMs where every eigenvalue is provably above tfgan's 1e-10 threshold):
 *   cg_axpby_eye_f64: out = alpha * a + beta_eye * I (n x n);
 *   cg_mat_stats_f64: out3[0] = trace(a), out3[1] = sum of squares of a (Frobenius norm squared),
 *                     out3[2] = smallest diagonal entry (an upper bound of the smallest eigenvalue
 *                     of a symmetric matrix); ws >= cg_mat_stats_workspace_bytes(). */
int cg_gemm_f64_ex(const double* a, const double* b, double* c, int m, int n, int k, int ta, int tb,
                   double alpha, double beta_eye, cgStream stream);
int cg_axpby_eye_f64(const double* a, double alpha, double beta_eye, double* out, int n,
                     cgStream stream);
size_t cg_mat_stats_workspace_bytes(void);
int cg_mat_stats_f64(const double* a, int n, double* out3, void* ws, size_t ws_bytes,
                     cgStream stream);
/* out[r, :] = a[r, :] * scale[r] (fp64): diag(f(w)) V when rebuilding a matrix function from its
 * eigen-decomposition (the symmetric square roots of the FID, metrics/fid_score.py:49-51). */
int cg_rowscale_f64(const double* a, const double* scale, double* out, int rows, int cols,
                    cgStream stream);
/* Symmetric eigen-decomposition by parallel cyclic one-sided Jacobi: a [d,d] fp64 symmetric
 * (destroyed), eigenvalues -> w [d] (unsorted), eigenvectors -> ROWS of v [d,d]
 * (a = v^T diag(w) v).  A pair of rows is rotated while |<g_p, g_q>| > tol * |g_p| |g_q|
 * (tol e.g. 1e-12) and neither row is numerically zero (squared norm below 1e-26 of the largest
 * row of `a`: the null space of a rank-deficient matrix); a sweep without rotations ends the work
 * on the device (the remaining launches return at once), max_sweeps bounds it.  From d = 256
 * (d % 64 == 0) the sweeps run in block form: 32 rows per workgroup, their Gram matrix, one cyclic
 * sweep on it, one pass applying the accumulated rotations.  ws >= cg_syevj_workspace_bytes(d).
 * v = NULL (block form only: d >= 256, d % 64 == 0): no eigenvectors are accumulated (half the
 * traffic of every apply pass) and w receives the MAGNITUDES |lambda_i| = |g_i| -- the singular
 * values, which is what the second matrix square root of the Frechet distance consumes
 * (tfgan's _symmetric_matrix_square_root sums f(s_i) over the SVD of a matrix that is positive
 * semi-definite up to rounding; metrics/fid_score.py:58-75). */
size_t cg_syevj_workspace_bytes(int d);
int cg_syevj_f64(double* a, int d, double* w, double* v, int max_sweeps, double tol, void* ws,
                 size_t ws_bytes, cgStream stream);
/* Eigenvalues only, by Householder tridiagonalisation + Sturm bisection (csrc/cg_tridiag.hip): what
 * the SECOND matrix square root of the Frechet distance needs -- trace(sqrtm(sqrt(sigma) sigma_v
 * sqrt(sigma))) is a sum over a spectrum (metrics/fid_score.py:58-75 via tfgan trace_sqrt_product).
 *   a    [n, n] symmetric fp64, DESTROYED (n <= 4096)
 *   w    [n] eigenvalues, ascending
 *   fro_out  optional [1]: |a|_F (of the tridiagonal form, the same number)
 * Backward stable: w are the exact eigenvalues of a + E with |E|_F a modest multiple of u |a|_F -- an
 * absolute statement; tiny eigenvalues of a graded matrix carry no relative accuracy (cg_syevj_f64's do).
 * cg_spectral_sqrt_bound_f64: out2[0] = sum_i f(|w_i|) with tfgan's rule f(s) = s < eps ? s : sqrt(s),
 * out2[1] = a bound on the error of that sum for |E|_F <= delta_f_rel * *fro and |E|_2 <= d2 =
 * delta_2_rel * *fro (Hoffman-Wielandt + Cauchy-Schwarz: |E|_F * sqrt(sum_i sup|f'|^2), plus
 * sqrt(eps + 2 d2) for every value within d2 of the cut-off, where f jumps) -- the certificate under
 * which metrics/fid_score.py uses this path instead of the Jacobi solve. */
size_t cg_sytrd_eigvals_workspace_bytes(int n);
int cg_sytrd_eigvals_f64(double* a, int n, double* w, double* fro_out, void* ws, size_t ws_bytes,
                         cgStream stream);
int cg_spectral_sqrt_bound_f64(const double* w, int n, double eps, double delta_f_rel,
                               double delta_2_rel, const double* fro, double* out2, cgStream stream);
/* The scalar part of the Frechet distance on the device (fid_score.py:58-75 through tfgan's
 * frechet_classifier_distance_from_activations), so that no host round trip sits between the two
 * eigen-decompositions:
 *   cg_spectral_sqrt_f64: f[i] = sign(w[i]) * (|w[i]| < eps ? |w[i]| : sqrt|w[i]|)  (tfgan's
 *     _symmetric_matrix_square_root rule on the spectrum; f may be NULL), *sum_out = sum_i f[i]
 *     (may be NULL);
 *   cg_fid_combine_f64: *out = tr(sigma) + tr(sigma_v) - 2 * *sqrt_trace + |mean - mean_v|^2. */
int cg_spectral_sqrt_f64(const double* w, int n, double eps, double* f, double* sum_out,
                         cgStream stream);
/* d[i] = f(|w_i|) / w_i^2 (0 for w_i = 0), f as in cg_spectral_sqrt_f64: cg_syevj_f64 with v = NULL
 * leaves the rows g_i = lambda_i v_i in its matrix argument, so the symmetric square root of a
 * covariance, sum_i f(lambda_i) v_i v_i^T, is G^T diag(d) G -- one GEMM, no eigenvector accumulation
 * in the sweeps (metrics/fid_score.py:58-75 via tfgan _symmetric_matrix_square_root). */
int cg_spectral_root_scale_f64(const double* w, int n, double eps, double* d, cgStream stream);
int cg_fid_combine_f64(const double* sigma, const double* sigma_v, const double* mean,
                       const double* mean_v, int d, const double* sqrt_trace, double* out,
                       cgStream stream);
/* KID (metrics/kid_score.py:129-136): out2[0] = sum, out2[1] = trace of the cubic polynomial kernel
 * (gram / dim + 1)^3 of an [m, n] fp64 Gram block (cg_gemm_f64 of two activation blocks).
 * ws >= cg_poly3_kernel_workspace_bytes(). */
size_t cg_poly3_kernel_workspace_bytes(void);
int cg_poly3_kernel_sums_f64(const double* gram, int m, int n, double inv_dim, double* out2,
                             void* ws, size_t ws_bytes, cgStream stream);
/* Inception score pieces: logits [n,k] fp32 -> exp(mean_i KL(p_i || mean_j p_j)) in fp64. */
size_t cg_inception_score_workspace_bytes(int64_t n, int k);
int cg_inception_score_f64(const float* logits, int64_t n, int k, double* score, void* ws,
                           size_t ws_bytes, cgStream stream);
/* Bilinear resize, TF1 legacy `resize_bilinear` (align_corners=False, no half-pixel centres;
 * eval_utils.py:165-175 via tfgan.eval.preprocess_image), then (x-128)/128.
 * x [N,H,W,C] fp32 in [0,255] -> y [N,Ho,Wo,C] bf16. */
int cg_inception_preprocess(const float* x, int N, int H, int W, int C, int Ho, int Wo, void* y,
                            cgStream stream);
/* General pooling for the Inception graph: kind 0 max, 1 avg (count includes only valid taps when
 * pad > 0, TF 'SAME' avg-pool semantics), window k, stride s, symmetric padding p. bf16 NHWC. */
int cg_pool2d(const void* x, int N, int H, int W, int C, int k, int s, int p, int kind, int Ho,
              int Wo, void* y, cgStream stream);
/* The same pooling on channel slices (x_ld / y_ld: elements between consecutive pixels, >= C, multiples
 * of 8, C % 8 == 0), finished by y = relu?(pool + bias): a pooling branch of an Inception block writes
 * straight into the block output, and the `avg_pool 3x3 -> 1x1 conv -> ReLU` branch (the frozen graph
 * behind eval_utils.py:165-175) runs as `1x1 conv` (a column group of the block's merged head
 * convolution, cg_gconv_ld with relu_cols) `-> avg_pool 3x3 + bias + ReLU` -- the two linear maps commute,
 * the pooling then moves a quarter of the channels. */
int cg_pool2d_ld(const void* x, int x_ld, int N, int H, int W, int C, int k, int s, int p, int kind,
                 int Ho, int Wo, void* y, int y_ld, const float* bias, int relu, cgStream stream);

#ifdef __cplusplus
}
#endif
#endif /* CGAMD_H_ */
