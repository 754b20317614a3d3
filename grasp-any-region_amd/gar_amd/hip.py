"""ctypes binding of libgar_hip.so (include/gar_hip.h). There is NO fallback: a missing library, a missing symbol,
an ABI mismatch or a non-gfx950 device raises. torch is used only to hold device memory and the stream."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GAR_HIP_LIB: load another build of the same sources (diagnostic builds of tools/: timeline stamps, de-phased GEMM)
LIB_PATH = os.environ.get("GAR_HIP_LIB") or os.path.join(_HERE, "libgar_hip.so")
# The twin build of the same sources whose 16-bit element type is IEEE binary16 (csrc/common.h, -DGAR_HALF_F16=1): the
# reference's `--data_type fp16` (demo/gar_with_mask.py:41-45). Same entry points; dtype code 1 means "the 16-bit type" there.
LIB_F16_PATH = os.environ.get("GAR_HIP_LIB_F16") or os.path.join(_HERE, "libgar_hip_f16.so")

GAR_F32, GAR_BF16 = 0, 1
(EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_SCALE_RES, EPI_RES, EPI_SWIGLU, EPI_PATCH_POS, EPI_QKV_ROPE,
 EPI_QKV_ROPE_LLM) = range(9)
ERR_UNSUPPORTED = -4
ABI_VERSION = 15


class GarError(RuntimeError):
    pass


class GemmParams(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("W", C.c_void_p), ("ldw", C.c_int64),
                ("C", C.c_void_p), ("ldc", C.c_int64), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("epilogue", C.c_int32), ("bias", C.c_void_p), ("residual", C.c_void_p), ("ldr", C.c_int64),
                ("gamma", C.c_void_p), ("pos", C.c_void_p), ("tokens_in", C.c_int32), ("tokens_out", C.c_int32),
                ("token_offset", C.c_int32), ("norm_eps", C.c_float), ("norm_w", C.c_void_p),
                ("qkv_q", C.c_void_p), ("qkv_k", C.c_void_p), ("qkv_sin", C.c_void_p), ("qkv_cos", C.c_void_p),
                ("qkv_heads", C.c_int32), ("qkv_head_dim", C.c_int32), ("qkv_tokens", C.c_int32),
                ("qkv_tokens_pad", C.c_int32), ("qkv_prefix", C.c_int32), ("qkv_q_scale", C.c_float),
                ("split_k", C.c_int32), ("partial", C.c_void_p), ("qkv_v", C.c_void_p),
                ("row_scale", C.c_void_p), ("row_stats", C.c_void_p),
                ("qkv_kv_heads", C.c_int32), ("qkv_kv_stride", C.c_int32), ("qkv_pos0", C.c_int32),
                ("qkv_pos_dev", C.c_void_p), ("qkv_left_pad", C.c_void_p), ("norm_folded", C.c_int32)]


_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
SIGNATURES = {
    "gar_abi_version": ([], _i),
    "gar_last_error": ([], C.c_char_p),
    "gar_check_device": ([_i], _i),
    "gar_gemm": ([_i, C.POINTER(GemmParams), _vp], _i),
    "gar_gemm_tile_takes": ([_i, C.POINTER(GemmParams)], _i),
    "gar_tokens_add": ([_i, _vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "gar_patch_im2col": ([_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "gar_mask_decode": ([_i, _vp, _vp, _i64, _i, _vp], _i),
    "gar_patch_embed_k": ([_i, _i], _i),
    "gar_patch_embed": ([_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp], _i),
    "gar_cls_pos_fill": ([_i, _vp, _vp, _vp, _i, _i, _i, _vp], _i),
    "gar_layernorm": ([_i, _vp, _vp, _vp, _vp, _i, _i, _i64, _i64, _f, _vp], _i),
    "gar_rmsnorm": ([_i, _vp, _vp, _vp, _i, _i, _i64, _i64, _f, _vp], _i),
    "gar_row_rstd": ([_i, _vp, _i, _i, _i64, _f, _i, _vp, _vp], _i),
    "gar_row_stats_finalize": ([_vp, _i, _i, _i, _f, _i, _vp, _vp], _i),
    "gar_splitk_residual_rmsnorm": ([_i, _vp, _i, _vp, _vp, _vp, _i, _i, _f, _vp], _i),
    "gar_vit_qkv_post": ([_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp], _i),
    "gar_vit_v_transpose": ([_i, _vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "gar_llm_qkv_post": ([_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp], _i),
    "gar_attention": ([_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp], _i),
    "gar_attention_vrow": ([_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp], _i),
    "gar_attention_decode_workspace": ([_i, _i, _i, _i], _i64),
    "gar_attention_decode": ([_i, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp], _i),
    "gar_attention_decode_qkv": ([_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _i, _i, _vp, _vp], _i),
    "gar_pool2x2":([_i, _vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "gar_placeholder_scan": ([_vp, _i, _i, _i64, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp], _i),
    "gar_pool_assemble": ([_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i64, _vp], _i),
    "gar_roi_replay_inplace": ([_i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "gar_embed_assemble": ([_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _vp], _i),
    "gar_roi_replay": ([_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _i, _i, _vp], _i),
    "gar_roi_replay_batched": ([_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "gar_resize_bicubic_h": ([_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp], _i),
    "gar_resize_bicubic_v_tiles": ([_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _f, _f, _vp], _i),
    "gar_resize_nearest_tiles": ([_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _f, _f, _vp], _i),
    "gar_rle_decode": ([C.c_char_p, _i64, _i, _i, _vp], _i64),
    "gar_embed_lookup": ([_i, _vp, _vp, _vp, _i, _i, _i64, _vp], _i),
    "gar_argmax": ([_i, _vp, _i64, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp], _i),
    "gar_argmax_workspace": ([_i, _i], _i64),
    "gar_sample": ([_i, _vp, _i64, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp], _i),
    "gar_counter_add": ([_vp, _i, _i, _vp], _i),
    "gar_input_check": ([_vp, _i, _i, _i64, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp], _i),
}

_lib = None
_lib_f16 = None
_last = None        # the library of the most recent call (whose gar_last_error() a failing rc refers to)


def load_library(path: str = None, f16: bool = False):
    """Loads the shared library and binds every symbol of include/gar_hip.h; raises if anything is missing."""
    global _lib, _lib_f16
    cached = _lib_f16 if f16 else _lib
    if cached is not None and path is None:
        return cached
    path = path or (LIB_F16_PATH if f16 else LIB_PATH)
    if not os.path.exists(path):
        raise GarError(f"{path} not found — build it with `python __graft_entry__.py` or "
                       f"`make -C grasp-any-region_amd/csrc` (there is no CPU/PyTorch fallback)")
    lib = C.CDLL(path)
    for name, (argtypes, restype) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GarError(f"{path} does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = restype
    v = lib.gar_abi_version()
    if v != ABI_VERSION:
        raise GarError(f"ABI mismatch: library {v}, binding {ABI_VERSION}")
    if f16:
        _lib_f16 = lib
    else:
        _lib = lib
    return lib


def lib(dt: torch.dtype = None):
    """The library that serves element type `dt`: libgar_hip_f16.so for torch.float16, libgar_hip.so otherwise."""
    global _last
    _last = load_library(f16=(dt == torch.float16))
    return _last


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = (_last or lib()).gar_last_error()
        raise GarError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def require_device(device_index: int = 0):
    if not torch.cuda.is_available():
        raise GarError("no GPU visible: gar_amd has no CPU path (the oracle under oracle/ is test infrastructure only)")
    check(lib().gar_check_device(device_index), "gar_check_device")


def num_cus(device_index: int = 0) -> int:
    """compute units of the device (the persistent tile GEMM launches one workgroup per CU: gar_amd/planner.py)."""
    return int(torch.cuda.get_device_properties(device_index).multi_processor_count)


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return GAR_F32
    if dt == torch.bfloat16 or dt == torch.float16:      # the 16-bit type of the library lib(dt) returns
        return GAR_BF16
    raise GarError(f"unsupported dtype {dt} (float32 parity mode, bfloat16 or float16)")


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream
