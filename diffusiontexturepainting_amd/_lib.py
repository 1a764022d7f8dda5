"""ctypes binding of libdtp.so (include/dtp.h).  There is no fallback: if the HIP library is
missing or an entry point fails, the caller gets an exception."""
import ctypes as C
import os

import torch  # noqa: F401  -- must come first: libdtp.so has to bind to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DTP_LIB") or os.path.join(_HERE, "libdtp.so")  # DTP_LIB: A/B another build of the same ABI
_lib = None
MAX_SLOTS = 16  # DTP_MAX_SLOTS


class DtpError(RuntimeError):
    pass


class Settings(C.Structure):
    _fields_ = [("steps", C.c_int), ("context_pad", C.c_int), ("tg_steps", C.c_int), ("cfg_weight", C.c_float),
                ("tg_weight", C.c_float), ("composite", C.c_int), ("output_u8", C.c_int)]


class ProfRow(C.Structure):
    _fields_ = [("kind", C.c_int), ("launches", C.c_int), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


_SHAPES = [(128, 128), (128, 64), (64, 64), (64, 128)]
PROF_KINDS = [f"gemm_kernel<{_SHAPES[i & 3][0]}, {_SHAPES[i & 3][1]}, {2 + i // 4}>" for i in range(12)] + \
    ["attention_kernel", "groupnorm (gn_stats+gn_apply | gn_fused)", "layernorm_kernel", "concat_kernel / small elementwise",
     "softmax_rows_kernel", "conv_halo_kernel<8, 16, 64>", "conv_halo_kernel<8, 16, 128>", "conv_halo_kernel<8, 8, 64>",
     "conv_halo_kernel<8, 8, 128>", "gemm_kernel<256, 128, 2>", "gemm_kernel<256, 128, 3>", "gemm_kernel<128, 256, 2>",
     "gemm_kernel<128, 256, 3>", "gemm_wide_kernel<256, 256>", "gemm_wide_kernel<256, 320>", "gemm_fp8_kernel"] + \
    [f"gemm_kernel<{_SHAPES[i & 3][0]}, {_SHAPES[i & 3][1]}, {2 + i // 4}, 2>" for i in range(8)] + \
    [f"gemm_kernel<{_SHAPES[i & 3][0]}, {_SHAPES[i & 3][1]}, 3, 1, {(4, 8)[i // 4]}>" for i in range(8)] + \
    ["xattn_kernel (cross-attention GEMM pair)", "conv_halo_kernel<8, 8, 64, 3 images>", "conv_halo_kernel<8, 8, 128, 3 images>",
     "lnlin_kernel (activation-stationary LayerNorm-folded Linear / GEGLU)", "convws_kernel<8, 8, 3 images>", "convws_kernel<16, 16>", "convws_kernel<8, 16, 2 n-tiles>", "convws_kernel<8, 16, 2 n-tiles, 2 workgroups per CU>",
     "gemmws_kernel (weight-streaming dense GEMM)"]


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("R", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("lda", C.c_int), ("ldw", C.c_int), ("ldc", C.c_int), ("ldr", C.c_int),
                ("conv", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int), ("Cin", C.c_int),
                ("stride", C.c_int), ("pad", C.c_int), ("upsample2x", C.c_int),
                ("flags", C.c_int), ("tile", C.c_int), ("splits", C.c_int), ("lns", C.c_void_p), ("ln_eps", C.c_float),
                ("A2", C.c_void_p), ("lda2", C.c_int), ("Cin2", C.c_int), ("Wcb", C.c_void_p),
                ("batch", C.c_int), ("a_bs", C.c_int64), ("w_bs", C.c_int64), ("c_bs", C.c_int64), ("r_bs", C.c_int64),
                ("bias_bs", C.c_int), ("lns_bs", C.c_int), ("sm_valid", C.c_int),
                ("st_out", C.c_void_p), ("st_in", C.c_void_p), ("st_parts", C.c_int), ("st_parts_out", C.c_int),
                ("W8", C.c_void_p), ("ldw8", C.c_int), ("a_scale", C.c_float), ("w_scale", C.c_float), ("gn_cpg", C.c_int), ("Wfr", C.c_void_p)]


GF_BIAS, GF_BIAS_M, GF_RESID, GF_GEGLU, GF_GELU, GF_QUICKGELU, GF_OUT_F32, GF_SILU, GF_LNFOLD = 1, 2, 4, 8, 64, 128, 256, 512, 1024
GF_SOFTMAX16 = 4096
GF_ROWSTATS = 2048
GF_GNSTATS = 1 << 24

# every symbol include/dtp.h declares: name -> (restype, argtypes)
_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
SYMBOLS = {
    "dtp_abi_version": (_i, []),
    "dtp_last_error": (C.c_char_p, []),
    "dtp_create": (_i, [_i, _i, _i, C.POINTER(_vp)]),
    "dtp_destroy": (None, [_vp]),
    "dtp_load_tensor": (_i, [_vp, C.c_char_p, _vp, _i, C.POINTER(_i64), _i]),
    "dtp_finalize_weights": (_i, [_vp]),
    "dtp_vae_encode": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "dtp_unet": (_i, [_vp, _vp, _f, _vp, _vp, _i, _vp]),
    "dtp_vae_decode": (_i, [_vp, _vp, _vp, _i, _vp]),
    "dtp_set_brush": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "dtp_set_conditioning": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "dtp_get_conditioning": (_i, [_vp, _vp, _vp, _vp]),
    "dtp_stamp": (_i, [_vp, _vp, C.POINTER(Settings), _vp, _vp, _vp, _i, _vp]),
    "dtp_stamp_slots": (_i, [_vp, _vp, C.POINTER(Settings), _vp, _vp, _vp, _i, C.POINTER(_i), _vp]),
    "dtp_set_brush_slot": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp]),
    "dtp_set_conditioning_slot": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "dtp_get_conditioning_slot": (_i, [_vp, _i, _vp, _vp, _vp]),
    "dtp_ddim_tables": (_i, [_i, C.POINTER(_i64), C.POINTER(_f), C.POINTER(_f)]),
    "dtp_last_stamp_times": (_i, [_vp, C.POINTER(_f * 3)]),
    "dtp_last_stamp_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "dtp_profile": (_i, [_vp, _i]),
    "dtp_profile_rows": (_i, [_vp, C.POINTER(ProfRow), _i, C.POINTER(_i)]),
    "dtp_profile_dump": (_i, [_vp, C.c_char_p]),
    "dtp_set_option": (_i, [_vp, C.c_char_p, _i]),
    "dtp_last_stamp_finite": (_i, [_vp, C.POINTER(_i)]),
    "dtp_op_gemm": (_i, [C.POINTER(GemmDesc), _vp]),
    "dtp_op_pack_linear": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "dtp_op_pack_conv": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "dtp_op_rowsum": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "dtp_op_quantize_w8": (_i, [_vp, _i, _i, _i, _vp, _i, C.POINTER(_f), _vp]),
    "dtp_op_pack_conv_cb": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "dtp_op_groupnorm_apply": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "dtp_op_pack_conv_ws": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "dtp_op_pack_conv_ws_elems": (C.c_longlong, [_i, _i, _i]),
    "dtp_op_pack_linear_ws": (_i, [_vp, _i, _vp, _i, _i, _vp]),
    "dtp_op_pack_linear_ws_elems": (C.c_longlong, [_i, _i]),
    "dtp_op_groupnorm": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "dtp_op_measure_peaks": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dtp_op_reduce_groupnorm": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "dtp_op_xattn": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "dtp_op_reduce_groupnorm_cx": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "dtp_op_xattn_ct": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "dtp_op_gn_fold_weights": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "dtp_op_ffchain": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "dtp_op_xchain": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "dtp_op_gn_linear": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp]),
    "dtp_op_layernorm": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _f, _vp]),
    "dtp_op_attention": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _f, _vp]),
    "dtp_op_attention_dma": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _f, _i, _vp]),
    "dtp_op_softmax_rows": (_i, [_vp, _i, _vp, _i, _i, _i, _f, _vp]),
    "dtp_op_attention_fp8": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _f, _f, _f, _vp]),
    "dtp_op_dilate": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
}


def load():
    """Load libdtp.so and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DtpError(f"{LIB_PATH} not found -- run `python -m diffusiontexturepainting_amd.build` "
                       "(there is no CPU or PyTorch fallback for the stamp path)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        if os.environ.get("DTP_LIB") and not hasattr(lib, name):
            continue  # A/B against an older build of the ABI: entry points it lacks simply stay unbound
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().dtp_last_error()
        raise DtpError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
