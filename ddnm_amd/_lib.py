"""ctypes binding of libddnm_hip.so (the C ABI declared in include/ddnm_hip.h).

The product path has NO CPU / PyTorch fallback: if the shared library is missing
or a kernel returns an error, this module raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libddnm_hip.so")

c_f32p = c_void_p   # device pointers travel as raw addresses


class ConvDesc(Structure):
    _fields_ = [
        ("src0", c_void_p), ("src1", c_void_p), ("weight", c_void_p), ("bias", c_void_p),
        ("badd", c_void_p), ("res", c_void_p), ("gn_scale", c_void_p), ("gn_shift", c_void_p),
        ("out", c_void_p),
        ("B", c_int32), ("Hin", c_int32), ("Win", c_int32),
        ("C0", c_int32), ("C1", c_int32), ("Cout", c_int32),
        ("ksize", c_int32), ("stride", c_int32), ("pad", c_int32),
        ("Ho", c_int32), ("Wo", c_int32),
        ("ups", c_int32), ("gn_silu", c_int32), ("out_nchw", c_int32),
        ("badd_stride", c_int32), ("tile", c_int32),
        ("workspace", c_void_p), ("workspace_floats", c_int64),
        ("res_ups", c_int32), ("src_f16", c_int32),
        ("stats_out", c_void_p),
        ("skip0", c_void_p), ("skip1", c_void_p), ("skip_weight", c_void_p),
        ("SC0", c_int32), ("SC1", c_int32),
        ("acc_scale", c_float), ("flags", c_int32),
        ("amax_in", c_void_p),
    ]


class GemmDesc(Structure):
    _fields_ = [
        ("A", c_void_p), ("Bm", c_void_p), ("D", c_void_p), ("C", c_void_p),
        ("M", c_int32), ("N", c_int32), ("K", c_int32),
        ("lda", c_int32), ("ldb", c_int32), ("ldc", c_int32), ("ldd", c_int32),
        ("transb", c_int32), ("batch", c_int32), ("inner", c_int32),
        ("sAo", c_int64), ("sAi", c_int64), ("sBo", c_int64), ("sBi", c_int64),
        ("sCo", c_int64), ("sCi", c_int64), ("sDo", c_int64), ("sDi", c_int64),
        ("alpha", c_float), ("beta", c_float),
        ("transa", c_int32), ("reserved", c_int32),
    ]


class Conv16Desc(Structure):
    _fields_ = [
        ("src", c_void_p), ("weight", c_void_p), ("bias", c_void_p), ("res", c_void_p),
        ("skip0", c_void_p), ("skip1", c_void_p), ("skip_weight", c_void_p),
        ("out", c_void_p), ("stats_out", c_void_p),
        ("workspace", c_void_p), ("workspace_floats", c_int64),
        ("B", c_int32), ("H", c_int32), ("W", c_int32),
        ("Cin", c_int32), ("Cout", c_int32), ("ksize", c_int32),
        ("ups", c_int32), ("res_ups", c_int32),
        ("SC0", c_int32), ("SC1", c_int32),
        ("src1", c_void_p), ("gn_scale", c_void_p), ("gn_shift", c_void_p),
        ("C0", c_int32), ("gn_silu", c_int32),
        ("out_nchw_f32", c_int32), ("reserved", c_int32),
        ("fin_gamma", c_void_p), ("fin_beta", c_void_p), ("fin_film", c_void_p),
        ("fin_scale", c_void_p), ("fin_shift", c_void_p),
        ("fin_eps", c_float), ("fin_film_stride", c_int32), ("fin_groups", c_int32), ("reserved2", c_int32),
    ]


class StepScalars(Structure):
    _fields_ = [("sqrt_1m_at", c_float), ("sqrt_at", c_float), ("sqrt_at_next", c_float),
                ("c1", c_float), ("c2", c_float), ("lam", c_float),
                ("rng_on", c_uint32), ("rng_seed_lo", c_uint32), ("rng_seed_hi", c_uint32), ("rng_iter", c_uint32),
                ("rng_image_base", c_uint32), ("reserved_rng", c_uint32)]


# name -> (restype, argtypes); must list every symbol include/ddnm_hip.h declares
PROTOTYPES = {
    "ddnm_version": (c_int32, []),
    "ddnm_error_string": (c_char_p, [c_int32]),
    "ddnm_build_digest": (c_char_p, []),
    "ddnm_sizeof": (c_int32, [c_int32]),
    "ddnm_conv2d_f32": (c_int32, [POINTER(ConvDesc), c_void_p]),
    "ddnm_conv3x3_small_cout_f32": (c_int32, [POINTER(ConvDesc), c_void_p]),
    "ddnm_conv3x3_small_cout_f32_supported": (c_int32, [POINTER(ConvDesc)]),
    "ddnm_conv2d_f32_tile_n": (c_int32, [POINTER(ConvDesc)]),
    "ddnm_conv2d_f32_workspace_floats": (c_int64, [POINTER(ConvDesc)]),
    "ddnm_conv2d_f32_stats_tiles": (c_int32, [POINTER(ConvDesc)]),
    "ddnm_conv2d_f32_fuses_skip": (c_int32, [POINTER(ConvDesc)]),
    "ddnm_spectral_mix_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_float, c_float, c_float,
                                        c_float, c_int32, c_void_p]),
    "ddnm_hq_x0_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int64, c_float, c_float, c_int32,
                                 c_void_p]),
    "ddnm_hq_project_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "ddnm_copy_rect_f32": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int32,
                                     c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_hq_sample_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float,
                                     c_float, c_void_p]),
    "ddnm_patchify_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_im2col3x3_f16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                     c_int32, c_int32, c_void_p]),
    "ddnm_gn_apply_f16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                    c_int32, c_void_p]),
    "ddnm_gn_finalize_tiles_f32": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                             c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_int32,
                                             c_void_p, c_void_p]),
    "ddnm_gn_finalize_tiles_amax_f32": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                                  c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_int32,
                                                  c_void_p, c_void_p, c_void_p]),
    "ddnm_amax_bound_f32": (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_int32, c_void_p]),
    "ddnm_conv3x3_f16_f32": (c_int32, [POINTER(ConvDesc), c_void_p]),
    "ddnm_conv3x3_f16_supported": (c_int32, [POINTER(ConvDesc)]),
    "ddnm_conv3x3_f16_workspace_floats": (c_int64, [POINTER(ConvDesc)]),
    "ddnm_conv3x3_f16_stats_tiles": (c_int32, [POINTER(ConvDesc)]),
    "ddnm_conv3x3_s16_f32": (c_int32, [POINTER(ConvDesc), c_void_p]),
    "ddnm_conv3x3_s16_supported": (c_int32, [POINTER(ConvDesc)]),
    "ddnm_conv3x3_s16_persistent": (c_int32, [POINTER(ConvDesc)]),
    "ddnm_conv3x3_s16_workspace_floats": (c_int64, [POINTER(ConvDesc)]),
    "ddnm_conv3x3_s16_stats_tiles": (c_int32, [POINTER(ConvDesc)]),
    "ddnm_conv3x3_s16_act_scale": (c_float, []),
    "ddnm_conv_gather_s16_f32": (c_int32, [POINTER(ConvDesc), c_void_p]),
    "ddnm_conv_gather_s16_supported": (c_int32, [POINTER(ConvDesc)]),
    "ddnm_conv_gather_s16_workspace_floats": (c_int64, [POINTER(ConvDesc)]),
    "ddnm_conv_gather_s16_stats_tiles": (c_int32, [POINTER(ConvDesc)]),
    "ddnm_conv16": (c_int32, [POINTER(Conv16Desc), c_void_p]),
    "ddnm_conv16_supported": (c_int32, [POINTER(Conv16Desc)]),
    "ddnm_conv16_fuses_fin": (c_int32, [POINTER(Conv16Desc)]),
    "ddnm_conv16_workspace_floats": (c_int64, [POINTER(Conv16Desc)]),
    "ddnm_conv16_stats_tiles": (c_int32, [POINTER(Conv16Desc)]),
    "ddnm_gn_apply_h16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                    c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_im2col3x3_h16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                     c_int32, c_int32, c_void_p]),
    "ddnm_nchw_to_nhwc_h16": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_gn_stats_h16": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_attn16_d64": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_conv1x1_f16_f32": (c_int32, [POINTER(ConvDesc), c_void_p]),
    "ddnm_conv1x1_f16_supported": (c_int32, [POINTER(ConvDesc)]),
    "ddnm_conv1x1_f16_workspace_floats": (c_int64, [POINTER(ConvDesc)]),
    "ddnm_conv1x1_f16_stats_tiles": (c_int32, [POINTER(ConvDesc)]),
    "ddnm_gn_stats_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                    c_int32, c_void_p]),
    "ddnm_gn_nchunk": (c_int32, [c_int32, c_int32]),
    "ddnm_gn_finalize_f32": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                       c_float, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "ddnm_bgemm_f32": (c_int32, [POINTER(GemmDesc), c_void_p]),
    "ddnm_gn_bwd_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32,
                                  c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p,
                                  c_void_p]),
    "ddnm_gn_bwd_nchunk": (c_int32, [c_int32, c_int32]),
    "ddnm_gn_bwd_h16": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32,
                                  c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p,
                                  c_void_p]),
    "ddnm_pool_tokens_h16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                       c_void_p]),
    "ddnm_pool_tokens_bwd_h16": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_attn16_d64_lse": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_attn16_d64_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                      c_int32, c_void_p]),
    "ddnm_softmax_bwd_rows_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_float, c_void_p]),
    "ddnm_pool_tokens_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                       c_void_p]),
    "ddnm_pool_attn_fwd_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_pool_attn_bwd_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                         c_void_p]),
    "ddnm_pool_tokens_bwd_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_logsoftmax_grad_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "ddnm_softmax_rows_f32": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_float, c_void_p]),
    "ddnm_linear_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                  c_void_p]),
    "ddnm_timestep_embedding_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_avgpool2_nhwc_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32,
                                         c_int32, c_void_p]),
    "ddnm_embedding_add_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_nchw_im2col3x3_pad_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_nchw_to_nhwc_pad_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_step_x0_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int64, POINTER(StepScalars),
                                   c_void_p]),
    "ddnm_step_combine_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                        c_int32, c_int64, POINTER(StepScalars), c_void_p]),
    "ddnm_step_sr_avgpool_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int32, c_int32, c_int32, c_int32, POINTER(StepScalars), c_void_p]),
    "ddnm_step_color_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int32, c_int32, POINTER(c_float), POINTER(StepScalars), c_void_p]),
    "ddnm_step_inpaint_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int32,
                                        c_void_p, c_void_p, c_int32, c_int32, POINTER(StepScalars), c_void_p]),
    "ddnm_step_denoise_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int32, c_int64, POINTER(StepScalars), c_void_p]),
    "ddnm_axpby_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p]),
    "ddnm_axpby_strided_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int64, c_float, c_float,
                                         c_void_p]),
    "ddnm_mask_mix_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_int64, c_float, c_float,
                                    c_float, c_float, c_void_p]),
    "ddnm_site_spectral_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                         c_int32, c_int32, c_void_p, c_int32, c_float, c_float, c_float, c_float,
                                         c_void_p]),
    "ddnm_gather_scale_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int64, c_void_p]),
    "ddnm_site_matmul_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int64, c_int64, c_int64,
                                       c_int32, c_void_p]),
    "ddnm_mul_planes_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_int64, c_void_p]),
    "ddnm_fill_f32": (c_int32, [c_void_p, c_int64, c_float, c_void_p]),
    "ddnm_attn_fused_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_float, c_float, c_void_p]),
    "ddnm_attn_fused_supported": (c_int32, [c_int32, c_int32]),
    "ddnm_randn_philox_f32": (c_int32, [c_void_p, c_int32, c_int64, c_uint32, c_uint32, c_uint32, c_uint32, c_void_p]),
    "ddnm_renoise_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p]),
    "ddnm_op_avgpool_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_op_upsample_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_op_color_A_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, POINTER(c_float), c_void_p]),
    "ddnm_op_color_pinv_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, POINTER(c_float), c_void_p]),
    "ddnm_op_inpaint_A_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p]),
    "ddnm_op_inpaint_pinv_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p]),
    "ddnm_fwht2d_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "ddnm_fwht2d_masked_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p,
                                         c_void_p]),
    "ddnm_wh_gather_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_wh_scatter_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "ddnm_finalize_psnr_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p]),
}

_lib = None


class DDNMHipError(RuntimeError):
    pass


ABI_VERSION = 7


def lib():
    """The loaded shared library; raises if it has not been built (no fallback).

    The binary is git-ignored and shipped prebuilt, so it identifies itself: `ddnm_build_digest()` must equal the
    sha256 of the sources lying next to it (ddnm_amd/build.py::_digest).  A missing or stale binary is rebuilt when
    hipcc is available (DDNM_NO_AUTOBUILD=1 disables that) and otherwise refused; the descriptor structs' sizes are
    checked against the ctypes mirrors as well."""
    global _lib
    if _lib is None:
        from . import build as _build
        want = _build._digest()
        stale = not os.path.exists(LIB_PATH) or not os.path.exists(_build.STAMP) or open(_build.STAMP).read() != want
        if stale and os.environ.get("DDNM_NO_AUTOBUILD") != "1":
            try:
                _build.build(force=False)
            except Exception as e:      # noqa: BLE001
                raise DDNMHipError(f"{LIB_PATH} is missing or older than its sources and could not be rebuilt: {e}")
        if not os.path.exists(LIB_PATH):
            raise DDNMHipError(
                f"{LIB_PATH} is missing: build it with `python -m ddnm_amd.build` "
                "(the DDNM hot path has no CPU/PyTorch fallback)")
        # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Load torch FIRST so the
        # kernels run in the HIP runtime that owns torch's device memory and streams; loading ours first
        # would bring up a second runtime and every launch would fail with hipErrorNoDevice.
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        got = l.ddnm_build_digest().decode()
        if got != want:
            raise DDNMHipError(f"{LIB_PATH} was built from different sources (binary {got[:12]}, sources {want[:12]}): "
                               "rebuild it with `python -m ddnm_amd.build --force`")
        if l.ddnm_version() != ABI_VERSION:
            raise DDNMHipError(f"ABI version {l.ddnm_version()} != {ABI_VERSION}")
        for idx, st in enumerate((ConvDesc, GemmDesc, Conv16Desc, StepScalars)):
            if l.ddnm_sizeof(idx) != ctypes.sizeof(st):
                raise DDNMHipError(f"{st.__name__}: ctypes mirror is {ctypes.sizeof(st)} bytes, the binary's struct "
                                   f"{l.ddnm_sizeof(idx)}")
        _lib = l
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().ddnm_error_string(code)
        raise DDNMHipError(f"{what} failed with code {code}: {msg.decode() if msg else '?'}")
