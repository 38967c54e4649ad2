"""ctypes binding of libsgclip_hip.so (the C ABI declared in include/sgclip.h).

There is no fallback: if the shared library is missing or a call fails, the
caller gets an exception.  Build it with `python __graft_entry__.py` (or
`make -C laser_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SGCLIP_LIB names another build of the same library (kernel experiments); the default is the in-tree one
LIB_PATH = os.environ.get("SGCLIP_LIB") or os.path.join(_HERE, "libsgclip_hip.so")

OK, EINVAL, ENOSPC, ELAUNCH, EBBOX = 0, -1, -2, -3, -4

CROP_OBJECT, CROP_PAIR, CROP_PLAIN = 0, 1, 2
LAYOUT_CHW, LAYOUT_PATCHES = 0, 1
DTYPE_F32, DTYPE_BF16 = 0, 1
SIM_LOGIT, SIM_SOFTMAX, SIM_SIGMOID = 0, 1, 2
EPI_BIAS_BF16, EPI_BIAS_GELU_BF16, EPI_BIAS_RESID_F32, EPI_PATCH_F32, EPI_F32, EPI_F32_ATOMIC = 0, 1, 2, 3, 4, 5
EPI_BIAS_GELU_BOTH_BF16, EPI_BIAS_MULGELUGRAD_BF16 = 6, 7
F32_PLAIN, F32_GELU, F32_RESID, F32_PATCH = 0, 1, 2, 3
CROP_LUT_BYTES = 768 + 768 * 4

(T_PATCH_W, T_CLS, T_POS, T_PRE_LN_W, T_PRE_LN_B, T_LN1_W, T_LN1_B, T_QKV_W, T_QKV_B, T_O_W, T_O_B, T_LN2_W,
 T_LN2_B, T_FC1_W, T_FC1_B, T_FC2_W, T_FC2_B, T_POST_LN_W, T_POST_LN_B, T_PROJ_W, T_TOK_EMB) = range(21)


class CropEntry(C.Structure):
    _fields_ = [("frame_off", C.c_int64), ("mask_a_off", C.c_int64), ("mask_b_off", C.c_int64),
                ("frame_w", C.c_int32), ("frame_h", C.c_int32),
                ("x1", C.c_int32), ("y1", C.c_int32), ("x2", C.c_int32), ("y2", C.c_int32),
                ("mode", C.c_int32), ("order", C.c_int32)]


class MaskEntry(C.Structure):
    _fields_ = [("packed_off", C.c_int64), ("out_off", C.c_int64), ("H", C.c_int32), ("W", C.c_int32)]


class TowerDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("hidden", C.c_int32), ("ffn", C.c_int32), ("layers", C.c_int32),
                ("heads", C.c_int32), ("tokens", C.c_int32), ("proj", C.c_int32), ("patch", C.c_int32),
                ("image", C.c_int32), ("vocab", C.c_int32), ("eos_id", C.c_int32), ("ln_eps", C.c_float)]


# name -> (restype, argtypes); every symbol include/sgclip.h declares
SIGNATURES = {
    "sgclip_abi_version": (C.c_int, []),
    "sgclip_last_error": (C.c_char_p, []),
    "sgclip_timing_enable": (C.c_int, [C.c_int, C.c_int]),
    "sgclip_timing_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "sgclip_crop_preprocess": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_void_p, C.c_void_p]),
    "sgclip_tower_layout": (C.c_int, [C.POINTER(TowerDesc), C.c_int, C.c_int, C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "sgclip_patch_kpad": (C.c_int, [C.c_int]),
    "sgclip_vit_workspace_bytes": (C.c_size_t, [C.POINTER(TowerDesc), C.c_int]),
    "sgclip_vit_encode": (C.c_int, [C.POINTER(TowerDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_size_t, C.c_void_p, C.c_void_p]),
    "sgclip_text_workspace_bytes": (C.c_size_t, [C.POINTER(TowerDesc), C.c_int, C.c_int]),
    "sgclip_text_encode": (C.c_int, [C.POINTER(TowerDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "sgclip_similarity_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "sgclip_similarity": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_size_t, C.c_void_p]),
    "sgclip_similarity_static_max": (C.c_int, []),
    "sgclip_similarity_static": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p,
                                           C.c_void_p]),
    "sgclip_segment_mean": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sgclip_topk_columns": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "sgclip_mask_unpack": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_void_p]),
    "sgclip_gemm_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "sgclip_linear_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sgclip_linear_exact_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "sgclip_attention_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sgclip_scale_softmax_columns": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "sgclip_layernorm": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "sgclip_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sgclip_gemm_tn_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]),
    "sgclip_qkv_attention_supported": (C.c_int, [C.c_int, C.c_int]),
    "sgclip_qkv_attention": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                       C.c_int, C.c_void_p]),
    "sgclip_attention_pooled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p]),
    "sgclip_transpose_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "sgclip_colsum": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sgclip_quick_gelu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "sgclip_quick_gelu_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "sgclip_cast_f32_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "sgclip_cast_colsum_f32_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sgclip_gemm_mulgelugrad_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sgclip_layernorm_bwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sgclip_attention_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p]),
    "sgclip_attention_bwd_bias": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p]),
    "sgclip_matmul_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                    C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "sgclip_dot_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "sgclip_prob_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sgclip_normalize_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sgclip_normalize_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sgclip_segment_mean_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p]),
    "sgclip_embed_text": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sgclip_embed_cls": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sgclip_rows_scatter_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sgclip_adamw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float,
                               C.c_float, C.c_float, C.c_int, C.c_void_p]),
}


class SgclipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"sgclip error {code}: {message}")
        self.code = code
        self.message = message


_lib = None


def load():
    """The loaded library (cached).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build the HIP extension first "
                               f"(python -c 'import __graft_entry__ as g; g.build()' or make -C laser_amd/csrc)")
        # torch ships its own HIP runtime (torch/lib/libamdhip64.so).  The process must hold ONE runtime: the one that
        # owns torch's device memory and streams has to be loaded before this library resolves its libamdhip64
        # dependency, or the kernels register with a second runtime that sees no device.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int):
    if rc != OK:
        msg = load().sgclip_last_error().decode("utf-8", "replace")
        if rc == EBBOX:
            raise AssertionError(msg)
        raise SgclipError(rc, msg)
