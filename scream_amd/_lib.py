"""ctypes binding of libscream_hip.so (include/scream_hip.h).

The product path has no CPU fallback: if the library cannot be built/loaded the first use
raises, loudly.  ``SCREAM_NO_BUILD=1`` forbids the lazy hipcc build (the GPU box normally
receives the prebuilt .so with the snapshot).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCREAM_LIB=<path>: load that build instead (A/B runs of two builds on the same GPU box; never built automatically)
LIB_PATH = os.environ.get("SCREAM_LIB") or os.path.join(_HERE, "libscream_hip.so")
ABI_VERSION = 18

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)
c_u64p = C.POINTER(C.c_uint64)


SPLIT_H1, SPLIT_H2, SPLIT_BF3 = 1, 2, 3


class TailExpsT(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("e_att", "e_wm", "e_m1", "e_w1", "e_h", "e_w2", "e_y", "e_wq", "e_x", "e_q")]


class LayerT(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("wqkv", "wq", "wkv", "wm", "w1", "w2", "g1", "b1", "g2", "b2", "tail")] +
                [(n, C.c_int32) for n in ("e_xq", "e_xkv", "e_wqkv", "e_wq", "e_wkv", "e_wm_g", "e_w1_g", "e_w2_g", "e_k", "e_v")] +
                [("tail_exps", TailExpsT), ("tail_next_q", C.c_int32), ("proj", C.c_void_p), ("tail_q_first", C.c_int32), ("proj_kv", C.c_void_p)])


class ModelT(C.Structure):
    _fields_ = [("n_self", C.c_int32), ("n_cross", C.c_int32), ("dim_t", C.c_void_p), ("emb_w", C.c_void_p),
                ("emb_b", C.c_void_p), ("pre_g", C.c_void_p), ("pre_b", C.c_void_p),
                ("layers_host", C.POINTER(LayerT)), ("c0_w", C.c_void_p), ("c0_b", C.c_void_p),
                ("c2_w", C.c_void_p), ("c2_b", C.c_void_p), ("c4_w", C.c_void_p), ("c4_b", C.c_void_p),
                ("stem_tgt_layers_host", C.POINTER(LayerT)), ("gemm_split", C.c_int32),
                ("e_c0x", C.c_int32), ("e_c0w", C.c_int32), ("e_c2x", C.c_int32), ("e_c2w", C.c_int32),
                ("wkv_cross", C.c_void_p), ("e_wkv_cross", C.c_int32), ("e_k_cross", C.c_int32), ("e_v_cross", C.c_int32),
                ("proj_cross", C.c_void_p)]


class BatchT(C.Structure):
    _fields_ = [("n_pairs", C.c_int32), ("rows_src", C.c_int64), ("rows_total", C.c_int64),
                ("max_chunks", C.c_int32), ("xyz", C.c_void_p), ("center", C.c_void_p),
                ("tile_cloud", C.c_void_p), ("cloud_row0", C.c_void_p), ("cloud_len", C.c_void_p)]


# name -> (restype, argtypes); every symbol include/scream_hip.h declares
V, I32, I64, F32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SIGNATURES = {
    "scream_version": (C.c_char_p, []),
    "scream_abi_version": (C.c_int, []),
    "scream_gemm_f32": (C.c_int, [V, I64, V, V, I64, I64, I32, I32, I32, I32, V, V, I64, V, V, V]),
    "scream_gemm_qkv_f32": (C.c_int, [V, I64, V, V, I64, I64, I32, I32, I32, V, V, V, I64, V, V]),
    "scream_kv_finalize": (C.c_int, [V, V, V, I64, I32, I32, V, V]),
    "scream_pack_w_split": (C.c_int, [V, I32, I32, I32, I32, V, V]),
    "scream_gemm_split_f32": (C.c_int, [V, I64, V, V, I64, I64, I32, I32, I32, I32, V, V, I64, V, V, I32, I32, I32, I32, V]),
    "scream_gemm_qkv_split_f32": (C.c_int, [V, I64, V, V, I64, I64, I32, I32, I32, V, V, V, I64, V, I32, I32, I32, I32, I32, I32, V]),
    "scream_proj_image_bytes": (C.c_int64, [I32, I32]),
    "scream_pack_proj": (C.c_int, [V, I32, I32, I32, I32, V, V]),
    "scream_proj_qkv_f32": (C.c_int, [V, V, V, I64, I32, I32, V, V, V, I64, V, I32, I32, I32, I32, I32, V]),
    "scream_tail_image_bytes": (C.c_int64, [I32, I32]),
    "scream_kv_image_bytes": (C.c_int64, []),
    "scream_pack_tail": (C.c_int, [V, V, V, V, I32, I32, C.POINTER(TailExpsT), V, V]),
    "scream_kv_finalize_image": (C.c_int, [V, V, V, I64, I32, I32, V, I32, I64, I64, I32, V]),
    "scream_layer_tail_f32": (C.c_int, [V, V, V, I32, V, V, V, V, V, V, V, V, V, I64, I32, C.POINTER(TailExpsT), V]),
    "scream_act_layout": (C.c_int, [V, V, I64, I32, V]),
    "scream_pe_embed_ln": (C.c_int, [V, V, V, V, V, V, V, V, V, I64, V]),
    "scream_pe_embed_ln_frag": (C.c_int, [V, V, V, V, V, V, V, V, V, I64, V]),
    "scream_kv_reduce": (C.c_int, [V, V, I64, I64, V, V, I32, I32, I32, V, V, V]),
    "scream_attn_apply": (C.c_int, [V, I64, V, V, I32, V, V, I64, I64, V]),
    "scream_coor_head": (C.c_int, [V, V, V, V, I64, V]),
    "scream_forward_workspace_bytes": (C.c_int64, [I64, I64, I32, I32, I32, I32]),
    "scream_forward": (C.c_int, [C.POINTER(ModelT), C.POINTER(BatchT), V, I64, V, V, V, V]),
    "scream_trace_create": (C.c_void_p, [I32]),
    "scream_trace_destroy": (None, [V]),
    "scream_trace_reset": (C.c_int, [V]),
    "scream_trace_read": (C.c_int, [V, I32, V, V, V, V, V]),
    "scream_trace_read_starts": (C.c_int, [V, I32, V]),
    "scream_nn_search": (C.c_int, [V, V, V, V, V, V, V, I32, I32, I32, I64, I64, F32, V, V, V, V, V, V]),
    "scream_square_distance": (C.c_int, [V, V, V, I32, I32, I32, V]),
    "scream_kabsch_corr": (C.c_int, [V, V, V, V, V, V, V, V, V, I32, V, V, V]),
    "scream_rigid_transform_3d": (C.c_int, [V, V, V, F32, I32, I32, V, V]),
    "scream_transformation_error": (C.c_int, [V, V, I32, V, V, V]),
    "scream_point_loss": (C.c_int, [V, V, V, V, V, V, I32, V, V]),
    "scream_icp_workspace_bytes": (C.c_int64, [I64, I64, I32]),
    "scream_icp_p2p": (C.c_int, [V, V, V, V, V, V, V, V, I32, I32, I32, I64, I64, F32, I32, F32, F32, V, V, V, V, I64, V]),
    "scream_icp_p2p_range": (C.c_int, [V, V, V, V, V, V, V, V, I32, I32, I32, I64, I64, F32, I32, F32, F32, V, V, V, I32, I32, V, V, I64, V]),
}

_lib: Optional[C.CDLL] = None


class ScreamHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load (building first if the in-tree .so is missing/stale and hipcc is present)."""
    global _lib
    if _lib is not None:
        return _lib
    if os.environ.get("SCREAM_NO_BUILD", "0") != "1" and not os.environ.get("SCREAM_LIB"):
        try:
            from . import build as _build
            _build.build()
        except Exception as e:  # no hipcc on this host: fall through to the prebuilt library
            if not os.path.exists(LIB_PATH):
                raise ScreamHipError("libscream_hip.so is missing and could not be built: %s" % e) from e
    if not os.path.exists(LIB_PATH):
        raise ScreamHipError("libscream_hip.so not found at %s (run `python -m scream_amd.build`)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ScreamHipError("libscream_hip.so does not export %s" % name) from e
        fn.restype = res
        fn.argtypes = args
    if lib.scream_abi_version() != ABI_VERSION:
        raise ScreamHipError("libscream_hip.so ABI %d != binding ABI %d" % (lib.scream_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc == -1:
        raise ScreamHipError("%s: invalid argument (SCREAM_EINVAL)" % what)
    if rc == -2:
        raise ScreamHipError("%s: unsupported shape (SCREAM_EUNSUPPORTED)" % what)
    raise ScreamHipError("%s: HIP error %d" % (what, rc))
