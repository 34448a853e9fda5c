"""Loads libcgamd.so and declares the C-ABI (include/cgamd.h) for ctypes.

There is NO fallback: if the shared library is missing or a symbol is absent the import fails
loudly -- the product path only ever runs the hand-written HIP kernels.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libcgamd.so")

c_int, c_i64, c_f32, c_f64 = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double
c_u64, c_u32, c_sz, vp = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_void_p


class ConvGeom(ctypes.Structure):
    """Mirror of cgConvGeom."""
    _fields_ = [(n, ctypes.c_int32) for n in
                ("N", "Hin", "Win", "Ci", "Ho", "Wo", "Co", "kh", "kw", "S", "U", "pt", "pl")]

    def key(self):
        return tuple(getattr(self, f) for f, _ in self._fields_)


class AdamEntry(ctypes.Structure):
    """Mirror of cgAdamEntry."""
    _fields_ = [("param", vp), ("grad", vp), ("m", vp), ("v", vp), ("ema", vp),
                ("n", c_i64), ("chunk_begin", c_i64)]


class SNItem(ctypes.Structure):
    """Mirror of cgSNItem."""
    _fields_ = [("w", vp), ("u", vp), ("u_out", vp), ("v_out", vp), ("sigma", vp), ("wbar", vp),
                ("ws", vp), ("K", ctypes.c_int32), ("Co", ctypes.c_int32), ("mode", ctypes.c_int32)]


class SNBwdItem(ctypes.Structure):
    """Mirror of cgSNBwdItem."""
    _fields_ = [("dwbar", vp), ("w", vp), ("a_k", vp), ("b_co", vp), ("sigma", vp), ("dw", vp),
                ("ws", vp), ("K", ctypes.c_int32), ("Co", ctypes.c_int32)]


class PrepItem(ctypes.Structure):
    """Mirror of cgPrepItem."""
    _fields_ = [("w", vp), ("bt_fwd", vp), ("bt_bwd", vp), ("T", ctypes.c_int32),
                ("Ci", ctypes.c_int32), ("Co", ctypes.c_int32)]


class WgradItem(ctypes.Structure):
    """Mirror of cgWgradItem."""
    _fields_ = [("geom", ConvGeom), ("in_", vp), ("gate_in", vp), ("slope_in", c_f32),
                ("accumulate", ctypes.c_int32), ("dy", vp), ("dw", vp), ("dbias", vp)]


class ConvFusion(ctypes.Structure):
    """Mirror of cgConvFusion."""
    _fields_ = [("bn_mean", vp), ("bn_var", vp), ("bn_gamma", vp), ("bn_beta", vp),
                ("bn_eps", c_f32), ("bn_per_sample", ctypes.c_int32), ("stats_out", vp),
                ("pool_out", ctypes.c_int32), ("in_up", ctypes.c_int32), ("out_scale", c_f32),
                ("bn_stat_group", ctypes.c_int32)]


ADAM_CHUNK = 16384
GP = ctypes.POINTER(ConvGeom)

# name -> (restype, argtypes)
SIGNATURES = {
    "cg_abi_version": (c_int, []),
    "cg_last_error": (ctypes.c_char_p, []),
    "cg_layer_norm_fwd": (c_int, [vp, c_int, c_i64, c_int, vp, vp, c_f32, vp, vp, vp, vp]),
    "cg_layer_norm_bwd_workspace_bytes": (c_sz, [c_int, c_int]),
    "cg_layer_norm_bwd": (c_int, [vp, vp, vp, vp, vp, c_int, c_i64, c_int, vp, vp, vp, vp, c_sz, vp]),
    "cg_layer_norm_bwd_bwd_workspace_bytes": (c_sz, [c_int, c_int]),
    "cg_layer_norm_bwd_bwd": (c_int, [vp, vp, vp, vp, vp, vp, c_int, c_i64, c_int, vp, vp, vp, vp, c_sz, vp]),
    "cg_calib_mfma_bf16": (c_int, [c_int, c_int, vp, ctypes.POINTER(c_f64), vp]),
    "cg_calib_mfma_bf16_zero": (c_int, [c_int, c_int, vp, ctypes.POINTER(c_f64), vp]),
    "cg_calib_copy": (c_int, [vp, vp, c_sz, vp]),
    "cg_prof_family_count": (c_int, []),
    "cg_prof_family_name": (ctypes.c_char_p, [c_int]),
    "cg_prof_enable": (c_int, [c_int]),
    "cg_prof_reset": (c_int, []),
    "cg_prof_collect": (c_int, [c_int, vp, vp, vp, vp]),
    "cg_weight_prep": (c_int, [vp, c_int, c_int, c_int, c_int, vp, vp, vp, vp]),
    "cg_weight_prep_elems": (ctypes.c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "cg_gconv": (c_int, [GP, vp, vp, vp, c_int, vp, vp, c_f32, vp, c_f32, vp, vp]),
    "cg_gconv_ld_supported": (c_int, [GP, c_int, c_int]),
    "cg_gconv_ld": (c_int, [GP, vp, c_int, vp, vp, c_int, c_int, vp, c_int, vp]),
    "cg_defer_create": (vp, []),
    "cg_defer_destroy": (None, [vp]),
    "cg_defer_pending": (c_int, [vp]),
    "cg_defer_flush": (c_int, [vp, vp]),
    "cg_defer_abort": (c_int, [vp]),
    "cg_gconv_fused_rows": (c_int, [GP]),
    "cg_gconv_fused_prologue_supported": (c_int, [GP]),
    "cg_gconv_fused_phases": (c_int, [GP]),
    "cg_gconv_fused": (c_int, [GP, vp, vp, vp, c_int, vp, vp, c_f32, vp, c_f32, vp, vp, vp]),
    "cg_gconv_pool_supported": (c_int, [GP]),
    "cg_gwgrad_pooled": (c_int, [GP, vp, vp, c_f32, vp, vp, c_int, vp, vp, c_sz, vp]),
    "cg_gwgrad_pooled_deferred": (c_int, [GP, vp, vp, c_f32, vp, vp, c_int, vp, vp, c_sz, vp, vp]),
    "cg_bn_finalize": (c_int, [vp, c_int, c_int, c_i64, vp, vp, vp, vp, c_f32, vp]),
    "cg_gwgrad_workspace_bytes": (c_sz, [GP]),
    "cg_gwgrad": (c_int, [GP, vp, vp, c_f32, vp, vp, c_f32, vp, c_int, vp, vp, c_sz, vp]),
    "cg_gwgrad_deferred": (c_int, [GP, vp, vp, c_f32, vp, vp, c_f32, vp, c_int, vp, vp, c_sz, vp, vp]),
    "cg_gwgrad_multi": (c_int, [ctypes.POINTER(WgradItem), c_int, vp, c_sz, vp]),
    "cg_gwgrad_groupable": (c_int, [GP]),
    "cg_spectral_norm_workspace_bytes": (c_sz, [c_int, c_int]),
    "cg_spectral_norm": (c_int, [vp, c_int, c_int, c_int, c_f32, vp, vp, vp, vp, vp, vp, c_sz, vp]),
    "cg_sn_backward_workspace_bytes": (c_sz, [c_int, c_int]),
    "cg_sn_backward": (c_int, [vp, vp, c_int, c_int, vp, vp, vp, vp, vp, c_sz, vp]),
    "cg_spectral_norm_multi_workspace_floats": (c_sz, [c_int, c_int]),
    "cg_spectral_norm_multi": (c_int, [vp, c_int, c_f32, vp]),
    "cg_sn_backward_multi_workspace_floats": (c_sz, [c_int, c_int]),
    "cg_sn_backward_multi": (c_int, [vp, c_int, vp]),
    "cg_weight_prep_multi": (c_int, [vp, c_int, vp]),
    "cg_flatten_multi": (c_int, [vp, vp, c_int, vp, vp]),
    "cg_scale_f32": (c_int, [vp, vp, c_f32, vp, c_i64, vp]),
    "cg_scale_count_nan_f32": (c_int, [vp, c_f32, vp, c_i64, vp, vp]),
    "cg_host_crc32c": (ctypes.c_uint32, [vp, c_sz, ctypes.c_uint32]),
    "cg_bn_finalize_groups": (c_int, [vp, c_int, c_int, c_i64, c_int, c_int, vp, vp, vp, vp, c_f32,
                                     vp]),
    "cg_bn_stats_groups_workspace_bytes": (c_sz, [c_i64, c_int, c_int]),
    "cg_bn_stats_groups": (c_int, [vp, c_i64, c_int, c_int, vp, vp, vp, vp, c_f32, vp, c_sz, vp]),
    "cg_bn_apply_groups": (c_int, [vp, c_int, c_int, c_int, vp, vp, c_f32, vp, vp, c_int, c_int,
                                  c_int, vp, vp]),
    "cg_bn_stats_workspace_bytes": (c_sz, [c_i64, c_int]),
    "cg_bn_stats": (c_int, [vp, c_i64, c_int, vp, vp, vp, vp, c_f32, vp, c_sz, vp]),
    "cg_bn_apply": (c_int, [vp, c_int, c_int, c_int, vp, vp, c_f32, vp, vp, c_int, c_int, vp, vp]),
    "cg_bn_backward_workspace_bytes": (c_sz, [c_int, c_int, c_int]),
    "cg_bn_backward_reduce": (c_int, [vp, vp, vp, c_int, c_int, c_int, vp, vp, c_f32, vp, c_int,
                                      c_int, vp, vp, vp, vp, c_sz, vp]),
    "cg_bn_backward_apply": (c_int, [vp, vp, vp, c_int, c_int, c_int, vp, vp, c_f32, vp, c_int,
                                     c_int, c_int, vp, vp, vp]),
    "cg_bn_moments_convert": (c_int, [vp, vp, c_int, c_int, c_f32, vp]),
    "cg_bn_accumulate": (c_int, [vp, vp, vp, vp, vp, c_int, vp]),
    "cg_bn_accumulated_moments": (c_int, [vp, vp, vp, vp, vp, c_int, vp]),
    "cg_bn_update_moving": (c_int, [vp, vp, vp, vp, c_int, c_f32, vp]),
    "cg_lrelu": (c_int, [vp, c_f32, vp, c_i64, vp]),
    "cg_lrelu_bwd": (c_int, [vp, vp, c_f32, vp, c_i64, vp]),
    "cg_axpby": (c_int, [vp, c_f32, vp, c_f32, vp, c_i64, vp]),
    "cg_axpby_f32": (c_int, [vp, c_f32, vp, c_f32, vp, c_i64, vp]),
    "cg_sum4": (c_int, [vp, vp, vp, vp, vp, c_i64, vp]),
    "cg_axpy_dev": (c_int, [vp, vp, vp, vp, c_i64, vp]),
    "cg_dot_bf16_workspace_bytes": (c_sz, [c_i64]),
    "cg_dot_bf16": (c_int, [vp, vp, c_i64, vp, vp, c_sz, vp]),
    "cg_avgpool2": (c_int, [vp, c_int, c_int, c_int, c_int, vp, vp]),
    "cg_avgpool2_bwd": (c_int, [vp, c_int, c_int, c_int, c_int, vp, vp]),
    "cg_unpool2": (c_int, [vp, vp, c_int, c_int, c_int, c_int, vp, vp]),
    "cg_unpool2_bwd": (c_int, [vp, c_int, c_int, c_int, c_int, vp, vp]),
    "cg_maxpool2": (c_int, [vp, c_int, c_int, c_int, c_int, vp, vp]),
    "cg_maxpool2_bwd": (c_int, [vp, vp, c_int, c_int, c_int, c_int, vp, vp]),
    "cg_spatial_reduce": (c_int, [vp, vp, c_int, c_int, c_int, c_f32, vp, vp]),
    "cg_spatial_reduce_bwd": (c_int, [vp, vp, c_int, c_int, c_int, c_f32, vp, vp]),
    "cg_dot_f32": (c_int, [vp, vp, c_i64, vp, vp]),
    "cg_pooled_head_supported": (c_int, [c_int, c_int]),
    "cg_pooled_head_fwd": (c_int, [vp, c_int, c_int, c_int, c_f32, vp, vp, vp, vp, vp]),
    "cg_pooled_head_bwd_workspace_bytes": (c_sz, [c_int, c_int]),
    "cg_pooled_head_bwd": (c_int, [vp, c_int, c_int, c_int, c_f32, vp, vp, vp, vp, vp, vp, vp, vp,
                                   c_sz, vp]),
    "cg_head": (c_int, [vp, c_int, vp, c_i64, vp]),
    "cg_head_bwd": (c_int, [vp, c_int, vp, c_int, vp, c_i64, vp]),
    "cg_cast_f32_to_bf16": (c_int, [vp, vp, c_i64, vp]),
    "cg_cast_bf16_to_f32": (c_int, [vp, vp, c_i64, vp]),
    "cg_affine_f32_to_bf16": (c_int, [vp, c_f32, c_f32, vp, c_i64, vp]),
    "cg_colsum_workspace_bytes": (c_sz, [c_i64, c_int]),
    "cg_colsum": (c_int, [vp, c_i64, c_int, vp, vp, c_sz, vp]),
    "cg_rowdot": (c_int, [vp, vp, c_int, c_int, vp, vp]),
    "cg_rowdot_bwd": (c_int, [vp, vp, vp, c_int, c_int, vp, vp, vp]),
    "cg_one_hot": (c_int, [vp, c_int, c_int, vp, vp]),
    "cg_attention_fwd": (c_int, [vp, vp, vp, c_int, c_int, c_int, c_int, c_int, vp, vp, vp]),
    "cg_attention_bwd_workspace_bytes": (c_sz, [c_int, c_int, c_int, c_int, c_int]),
    "cg_attention_bwd": (c_int, [vp, vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, vp,
                                 vp, vp, vp, c_sz, vp]),
    "cg_gan_loss": (c_int, [c_int, vp, c_int, vp, vp, vp, vp]),
    "cg_softmax_xent_eps": (c_int, [vp, vp, c_int, c_int, c_f32, vp, vp, vp]),
    "cg_s3gan_labels": (c_int, [vp, vp, c_int, c_int, c_int, vp, vp, vp]),
    "cg_softmax_xent_weighted": (c_int, [vp, vp, vp, c_int, c_int, vp, vp, vp]),
    "cg_interpolate": (c_int, [vp, vp, vp, c_int, c_i64, vp, vp]),
    "cg_gradient_penalty": (c_int, [vp, c_int, c_i64, vp, vp, vp]),
    "cg_gradient_penalty_bwd": (c_int, [vp, vp, vp, c_int, c_i64, vp, vp]),
    "cg_moments_workspace_bytes": (c_sz, []),
    "cg_moments_f32": (c_int, [vp, c_i64, vp, vp, c_sz, vp]),
    "cg_dragan_perturb": (c_int, [vp, vp, vp, c_i64, c_f32, c_f32, vp, vp]),
    "cg_adam_multi": (c_int, [vp, c_int, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, vp, c_f32,
                              c_i64, vp]),
    "cg_counter_add": (c_int, [vp, c_i64, vp]),
    "cg_multi_gather": (c_int, [vp, vp, c_int, c_i64, vp, vp]),
    "cg_multi_scatter": (c_int, [vp, vp, c_int, c_i64, vp, vp]),
    "cg_random": (c_int, [c_int, c_f32, c_f32, c_u64, c_u32, c_u32, vp, vp, c_i64, vp]),
    "cg_random_labels": (c_int, [c_int, c_u64, c_u32, c_u32, vp, vp, c_i64, vp]),
    "cg_mean_cov_workspace_bytes": (c_sz, [c_i64, c_int]),
    "cg_mean_cov_f64": (c_int, [vp, c_i64, c_int, vp, vp, vp, c_sz, vp]),
    "cg_gemm_f64": (c_int, [vp, vp, vp, c_int, c_int, c_int, c_int, c_int, vp]),
    "cg_gemm_f64_ex": (c_int, [vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_f64, c_f64, vp]),
    "cg_axpby_eye_f64": (c_int, [vp, c_f64, c_f64, vp, c_int, vp]),
    "cg_mat_stats_workspace_bytes": (c_sz, []),
    "cg_mat_stats_f64": (c_int, [vp, c_int, vp, vp, c_sz, vp]),
    "cg_poly3_kernel_workspace_bytes": (c_sz, []),
    "cg_poly3_kernel_sums_f64": (c_int, [vp, c_int, c_int, c_f64, vp, vp, c_sz, vp]),
    "cg_rowscale_f64": (c_int, [vp, vp, vp, c_int, c_int, vp]),
    "cg_spectral_sqrt_f64": (c_int, [vp, c_int, c_f64, vp, vp, vp]),
    "cg_spectral_root_scale_f64": (c_int, [vp, c_int, c_f64, vp, vp]),
    "cg_fid_combine_f64": (c_int, [vp, vp, vp, vp, c_int, vp, vp, vp]),
    "cg_sytrd_eigvals_workspace_bytes": (c_sz, [c_int]),
    "cg_sytrd_eigvals_f64": (c_int, [vp, c_int, vp, vp, vp, c_sz, vp]),
    "cg_spectral_sqrt_bound_f64": (c_int, [vp, c_int, c_f64, c_f64, c_f64, vp, vp, vp]),
    "cg_syevj_workspace_bytes": (c_sz, [c_int]),
    "cg_syevj_f64": (c_int, [vp, c_int, vp, vp, c_int, c_f64, vp, c_sz, vp]),
    "cg_inception_score_workspace_bytes": (c_sz, [c_i64, c_int]),
    "cg_inception_score_f64": (c_int, [vp, c_i64, c_int, vp, vp, c_sz, vp]),
    "cg_inception_preprocess": (c_int, [vp, c_int, c_int, c_int, c_int, c_int, c_int, vp, vp]),
    "cg_pool2d": (c_int, [vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                          c_int, vp, vp]),
    "cg_pool2d_ld": (c_int, [vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                             c_int, c_int, vp, c_int, vp, c_int, vp]),
}

_lib = None


class CgamdError(RuntimeError):
    pass


def load():
    """Returns the loaded library (loads + type-declares on first use)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("CGAMD_LIB_PATH") or LIB_PATH   # override: instrumented debug builds only
    if not os.path.exists(path):
        raise CgamdError(
            "libcgamd.so not found at %s: run `python -m compare_gan_amd.csrc.build` "
            "(there is no CPU / eager fallback for the HIP kernels)" % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud failure
        fn.restype = res
        fn.argtypes = args
    if os.environ.get("CGAMD_TRACE"):
        lib = _TracingLib(lib)
    _lib = lib
    return lib


class _TracingLib(object):
    """Debug aid (CGAMD_TRACE=1): prints every C-ABI call and synchronises after it, so that a
    faulting kernel is the last name printed."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("cg_") or name in ("cg_last_error", "cg_abi_version",
                                                  "cg_prof_family_name", "cg_prof_family_count"):
            return fn

        def traced(*args):
            import sys
            import torch
            desc = []
            for a in args:
                if hasattr(a, "_obj") and isinstance(a._obj, ConvGeom):
                    desc.append(str(a._obj.key()))
                elif isinstance(a, (int, float)):
                    desc.append(repr(a))
            sys.stderr.write("[cgamd] %s %s\n" % (name, " ".join(desc)))
            sys.stderr.flush()
            rc = fn(*args)
            if not name.endswith("_bytes"):
                torch.cuda.synchronize()
            return rc
        return traced


def check(rc, what):
    if rc != 0:
        msg = load().cg_last_error()
        raise CgamdError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))
