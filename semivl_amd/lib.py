"""ctypes binding of libsemivl_hip.so (include/semivl_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised
(the same way ATen failures surface to `semivl.py` in the reference).
"""
import ctypes as C
import os
import subprocess

import torch  # noqa: F401  (must be imported BEFORE the .so so that both share torch's HIP runtime, not /opt/rocm's copy)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsemivl_hip.so")
CSRC = os.path.join(_HERE, "csrc")

c_f32p = C.c_void_p
c_i64p = C.c_void_p


class Operand(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ld", C.c_int64), ("bs_outer", C.c_int64), ("bs_inner", C.c_int64)]


class ConvGeom(C.Structure):
    _fields_ = [("H", C.c_int), ("W", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int), ("stride", C.c_int),
                ("C1", C.c_int), ("C2", C.c_int), ("rep", C.c_int),
                ("KH", C.c_int), ("KW", C.c_int), ("dil", C.c_int), ("pad", C.c_int), ("sign", C.c_int),
                ("src2", C.c_void_p), ("ld2", C.c_int64), ("patch", C.c_int)]


class GemmDesc(C.Structure):
    _fields_ = [("a_mode", C.c_int), ("b_mode", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("batch", C.c_int), ("batch_inner", C.c_int), ("ksplit", C.c_int),
                ("A", Operand), ("B", Operand), ("conv", ConvGeom),
                ("C", C.c_void_p), ("out_mode", C.c_int),
                ("ldc_m", C.c_int64), ("ldc_n", C.c_int64), ("c_bs_outer", C.c_int64), ("c_bs_inner", C.c_int64),
                ("ct_H", C.c_int), ("ct_W", C.c_int), ("ct_Cout", C.c_int),
                ("alpha", C.c_float), ("bias", C.c_void_p), ("bias_mod", C.c_int), ("act", C.c_int),
                ("preact", C.c_void_p), ("resid", C.c_void_p),
                ("ldr_m", C.c_int64), ("ldr_n", C.c_int64), ("r_bs_outer", C.c_int64), ("r_bs_inner", C.c_int64),
                ("accumulate", C.c_int), ("conv_w_planes", C.c_void_p), ("emu_ws", C.c_void_p)]


class PGemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("a_rows", C.c_int64), ("b_rows", C.c_int64),
                ("m_off", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("C", C.c_void_p), ("ldc", C.c_int64), ("planes_out", C.c_void_p), ("p_rows", C.c_int64),
                ("bias", C.c_void_p), ("act", C.c_int), ("preact", C.c_void_p), ("resid", C.c_void_p),
                ("ldr", C.c_int64), ("accumulate", C.c_int),
                ("fmt", C.c_int), ("p_fmt", C.c_int), ("a_sexp", C.c_void_p), ("b_sexp", C.c_void_p),
                ("a_rnorm", C.c_void_p), ("b_bound", C.c_void_p), ("p_sexp", C.c_void_p)]


class CeDesc(C.Structure):
    _fields_ = [("logits", C.c_void_p), ("B", C.c_int), ("N", C.c_int), ("HW", C.c_int64),
                ("target", C.c_void_p), ("use_ignore_t", C.c_int), ("conf", C.c_void_p), ("ign", C.c_void_p),
                ("conf_thresh", C.c_float), ("all_pixels", C.c_int), ("mc_target", C.c_void_p), ("partials", C.c_void_p),
                ("dlogits", C.c_void_p), ("gscale", C.c_void_p), ("img_weight", C.c_void_p)]


class CeUpDesc(C.Structure):
    _fields_ = [("logits", C.c_void_p), ("B", C.c_int), ("N", C.c_int), ("h", C.c_int), ("w", C.c_int), ("H", C.c_int),
                ("W", C.c_int), ("align_corners", C.c_int), ("target", C.c_void_p), ("use_ignore_t", C.c_int),
                ("conf", C.c_void_p), ("ign", C.c_void_p), ("conf_thresh", C.c_float), ("all_pixels", C.c_int),
                ("mc_target", C.c_void_p), ("partials", C.c_void_p), ("dlogits", C.c_void_p), ("gscale", C.c_void_p),
                ("img_weight", C.c_void_p)]


class SeqAttnDesc(C.Structure):
    _fields_ = [("groups", C.c_int), ("inner", C.c_int), ("seq", C.c_int), ("heads", C.c_int),
                ("outer_stride", C.c_int64), ("inner_stride", C.c_int64), ("seq_stride", C.c_int64),
                ("qkv", C.c_void_p), ("out", C.c_void_p), ("probs", C.c_void_p),
                ("dout", C.c_void_p), ("dqkv", C.c_void_p), ("dscores", C.c_void_p)]


# name -> (restype, argtypes); this table is also what tests use to check that every symbol of the header exists.
_I, _L, _F, _D, _P = C.c_int, C.c_int64, C.c_float, C.c_double, C.c_void_p
SIGNATURES = {
    "svl_version": (_I, []),
    "svl_last_error": (_I, [C.c_char_p, C.c_size_t]),
    "svl_gemm_f32": (_I, [C.POINTER(GemmDesc), _P]),
    "svl_set_gemm_emulation": (_I, [_I]),
    "svl_get_gemm_emulation": (_I, []),
    "svl_set_conv_tiled": (_I, [_I]),
    "svl_planes_rows": (_L, [_L]),
    "svl_planes_bytes": (_L, [_L, _I]),
    "svl_split_planes_bf16x3": (_I, [_P, _L, _L, _L, _I, _P, _L, _L, _P]),
    "svl_planes_bytes_fmt": (_L, [_L, _I, _I]),
    "svl_split_planes_f16x2": (_I, [_P, _L, _L, _L, _I, _P, _L, _L, _P, _P, _P]),
    "svl_gemm_planes_f32": (_I, [C.POINTER(PGemmDesc), _P]),
    "svl_conv3x3_wgrad_tiled_groups": (_I, [_I, _I, _I, _I, _I]),
    "svl_conv3x3_wgrad_tiled": (_I, [_P, _L, _I, _P, _L, _I, _P, _L, _I, _I, _I, _I, _I, _P, _I, _P, _P]),
    "svl_reduce_slabs_f32": (_I, [_P, _P, _I, _L, _I, _P]),
    "svl_softmax_max_f32": (_I, [_P, _I, _I, _L, _P, _P, _P]),
    "svl_cutmix_f32": (_I, [_P, _P, _P, _P, _I, _I, _L, _P]),
    "svl_cutmix_i64": (_I, [_P, _P, _P, _P, _I, _L, _P]),
    "svl_ce_num_blocks": (_L, [_I, _I, _L]),
    "svl_ce_fused_f32": (_I, [C.POINTER(CeDesc), _P]),
    "svl_ce_finalize": (_I, [_P, _L, _P, _P]),
    "svl_ce_up_num_blocks": (_L, [_I, _I, _I, _I, _I, _I, _I]),
    "svl_ce_up_fused_f32": (_I, [C.POINTER(CeUpDesc), _P]),
    "svl_softmax_max_up_f32": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "svl_semivl_gscale": (_I, [_P, _D, _F, _P, _P, _P, _P]),
    "svl_semivl_loss": (_I, [_P, _D, _F, _P, _P, _P, _P]),
    "svl_conf_ratio_f32": (_I, [_P, _P, _I, _L, _F, _P, _P, _P]),
    "svl_conf_avg_ws_doubles": (_L, [_I]),
    "svl_conf_avg_factor": (_I, [_P, _P, _I, _L, _P, _P, _P]),
    "svl_stream_release": (_I, [_P]),
    "svl_stream_prepare": (_I, [_P]),
    "svl_stream_helper": (_I, [_P, C.POINTER(C.c_void_p)]),
    "svl_last_gemm_path": (_I, []),
    "svl_shutdown": (_I, []),
    "svl_num_stream_contexts": (_I, []),
    "svl_clock_probe": (_I, [_P, _I, C.c_uint64, _P]),
    "svl_bernoulli_f32": (_I, [_P, _L, _F, C.c_uint64, C.c_uint64, _P]),
    "svl_softmax_planes_f32": (_I, [_P, _I, _I, _L, _P, _P]),
    "svl_count_valid_i64": (_I, [_P, _L, _P, _P]),
    "svl_maskclip_labels": (_I, [_P, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P]),
    "svl_concept_max_f32": (_I, [_P, _I, _I, _L, _P, _I, _P, _P]),
    "svl_iou_hist_i64": (_I, [_P, _P, _L, _I, _I, _P, _P]),
    "svl_layernorm_fwd": (_I, [_P, _P, _P, _F, _L, _I, _P, _P, _P]),
    "svl_layernorm_fwd_planes": (_I, [_P, _P, _P, _F, _L, _I, _P, _P, _P, _L, _P]),
    "svl_layernorm_fwd_planes_f16x2": (_I, [_P, _P, _P, _F, _L, _I, _P, _P, _P, _L, _P, _P, _P]),
    "svl_layernorm_bwd_parts": (_I, [_L]),
    "svl_layernorm_bwd": (_I, [_P, _P, _P, _P, _L, _I, _P, _P, _P, _P, _P]),
    "svl_softmax_rows_fwd": (_I, [_P, _L, _I, _L, _F, _P]),
    "svl_softmax_rows_bwd": (_I, [_P, _P, _L, _I, _L, _F, _P]),
    "svl_l2norm_fwd": (_I, [_P, _L, _I, _F, _P, _P, _P]),
    "svl_l2norm_bwd": (_I, [_P, _P, _P, _L, _I, _P, _P]),
    "svl_colsum_ws_floats": (_L, [_L, _I]),
    "svl_colsum_f32": (_I, [_P, _L, _I, _L, _P, _I, _P, _P]),
    "svl_eltwise_f32": (_I, [_I, _P, _P, _P, _L, _P]),
    "svl_chanmask_f32": (_I, [_P, _P, _F, _L, _I, _I, _P, _P]),
    "svl_fill_f32": (_I, [_P, _F, _L, _P]),
    "svl_affine_planes_f32": (_I, [_P, _L, _I, _L, _P, _P, _P]),
    "svl_permute_rows_f32": (_I, [_P, _L, _I, _I, _I, _P, _P]),
    "svl_permute4_f32": (_I, [_P, _P, _L, _L, _L, _L, _L, _L, _L, _L, _P]),
    "svl_bound2_f32": (_I, [_P, _L, _P, _L, _P, _P]),
    "svl_copy2d_f32": (_I, [_P, _L, _L, _L, _P, _L, _L, _L, _L, _I, _I, _P]),
    "svl_groupnorm_fwd": (_I, [_P, _L, _P, _P, _F, _I, _L, _I, _I, _I, _P, _L, _P, _P]),
    "svl_conv3x3_gn_ws_doubles": (_L, [_I, _I, _I, _I]),
    "svl_conv3x3_gn_f32": (_I, [_P, _L, _I, _P, _L, _I, _I, _P, _I, _I, _I, _I, _P, _L, _F, _P, _P, _P, _P, _P]),
    "svl_conv3x3_weight_planes_bytes": (_L, [_I, _I]),
    "svl_conv3x3_weight_planes": (_I, [_P, _I, _I, _P, _P]),
    "svl_groupnorm_scale_shift": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "svl_groupnorm_apply": (_I, [_P, _L, _P, _P, _I, _L, _I, _I, _I, _P, _P, _L, _P]),
    "svl_groupnorm_bwd": (_I, [_P, _L, _P, _L, _P, _L, _P, _P, _P, _I, _L, _I, _I, _I, _P, _L, _P, _P]),
    "svl_conv3x3_gnb_ws_doubles": (_L, [_I, _I, _I, _I]),
    "svl_conv3x3_dgrad_gnb_f32": (_I, [_P, _L, _I, _P, _I, _I, _I, _I, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P]),
    "svl_groupnorm_bwd_apply": (_I, [_P, _L, _P, _L, _P, _P, _P, _I, _L, _I, _I, _I, _P, _P, _L, _P]),
    "svl_bn_ws_doubles": (_L, [_L, _I]),
    "svl_bn_stats": (_I, [_P, _L, _L, _I, _P, _P, _P]),
    "svl_bn_finalize": (_I, [_P, _D, _F, _F, _P, _P, _I, _P, _P, _P]),
    "svl_bn_eval_invstd": (_I, [_P, _F, _I, _P, _P]),
    "svl_bn_apply": (_I, [_P, _L, _L, _I, _P, _P, _P, _P, _P, _L, _I, _P, _L, _P]),
    "svl_bn_bwd_reduce": (_I, [_P, _L, _P, _L, _P, _L, _P, _P, _P, _P, _L, _I, _P, _P, _P]),
    "svl_bn_bwd_apply": (_I, [_P, _L, _P, _L, _P, _L, _P, _P, _P, _P, _P, _D, _L, _I, _P, _L, _P, _L, _P]),
    "svl_maxpool3x3s2_fwd": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "svl_maxpool3x3s2_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "svl_aug_resample_u8": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "svl_aug_to_float": (_I, [_P, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P]),
    "svl_aug_mask_i64": (_I, [_P, _I, _I, _I, _P, _P]),
    "svl_aug_photometric_u8": (_I, [_P, _I, _I, _F, _P, _P]),
    "svl_aug_gaussian_blur_u8": (_I, [_P, _I, _I, _F, _P, _P, _P]),
    "svl_attention_fwd": (_I, [_P, _I, _I, _I, _P, _P, _P, _L, _P]),
    "svl_attention_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _L, _P]),
    "svl_attention_h2_ws_bytes": (_L, [_I, _I, _I, _I]),
    "svl_attention_fwd_h2": (_I, [_P, _I, _I, _I, _P, _P, _P, _L, _P, _L, _P]),
    "svl_attention_bwd_h2": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _L, _P, _L, _P]),
    "svl_conv_cout1_fwd": (_I, [_P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "svl_conv_cout1_wgrad_blocks": (_I, [_I, _I, _I]),
    "svl_conv_cout1_wgrad": (_I, [_P, _P, _L, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "svl_tap_gather": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "svl_seqattn_fwd": (_I, [C.POINTER(SeqAttnDesc), _P]),
    "svl_seqattn_bwd": (_I, [C.POINTER(SeqAttnDesc), _P]),
    "svl_bilinear_nhwc_fwd": (_I, [_P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P, _L, _I, _P]),
    "svl_sum_rep_f32": (_I, [_P, _L, _L, _I, _L, _I, _P, _P]),
    "svl_bilinear_nhwc_bwd": (_I, [_P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P, _L, _I, _P]),
    "svl_bilinear_planes_fwd": (_I, [_P, _L, _I, _I, _I, _I, _I, _P, _P]),
    "svl_bilinear_planes_bwd": (_I, [_P, _L, _I, _I, _I, _I, _I, _P, _P]),
    "svl_avgpool_cat_fwd": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P]),
    "svl_avgpool_cat_bwd": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "svl_avgpool_cat_bwd_text": (_I, [_P, _I, _L, _I, _I, _I, _P, _P, _P]),
    "svl_adamw_step": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _L, _F, _F, _F, _I, _F, _P, _F, _P]),
}

_lib = None


def build_library(force=False):
    """Compile libsemivl_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if os.path.exists(LIB_PATH) and not force:
        srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
        srcs.append(os.path.join(_HERE, "..", "include", "semivl_hip.h"))
        if all(os.path.getmtime(s) <= os.path.getmtime(LIB_PATH) for s in srcs):
            return LIB_PATH
    subprocess.run(["make", "-C", CSRC, "-j8"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load():
    """Load the library; raises RuntimeError if it is absent (no CPU/eager fallback exists by design)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"semivl_amd: {LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C semivl_amd/csrc`. There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    buf = C.create_string_buffer(512)
    load().svl_last_error(buf, 512)
    return buf.value.decode()


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"semivl_hip {what} failed (status {rc}): {last_error()}")
