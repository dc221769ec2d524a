"""ctypes binding of libanyloc_hip.so (the C ABI declared in include/anyloc_hip.h).

There is NO CPU fallback: if the shared library has not been built, or no ROCm
GPU is visible, the product raises.  Build with ``python -m anyloc_amd.build``
(or ``__graft_entry__.build()``).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libanyloc_hip.so")

c_f32p = C.c_void_p      # device pointers travel as integers (tensor.data_ptr())
c_i64p = C.c_void_p
c_i64 = C.c_int64
c_sz = C.c_size_t


class AnylocHipError(RuntimeError):
    pass


class VitConfig(C.Structure):
    _fields_ = [("dim", C.c_int32), ("depth", C.c_int32), ("heads", C.c_int32),
                ("ffn_kind", C.c_int32), ("ffn_hidden", C.c_int32), ("patch", C.c_int32),
                ("patch_k_pad", C.c_int32)]


BLOCK_FIELDS = ("norm1_w", "norm1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ls1",
                "norm2_w", "norm2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ls2")


class VitBlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in BLOCK_FIELDS]


X3_FIELDS = ("qkv_w3", "proj_w3", "fc1_w3", "fc2_w3")


class VitBlockX3(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in X3_FIELDS]


H2_FIELDS = ("qkv_w2", "qkv_inv", "proj_w2", "proj_inv", "fc1_w2", "fc1_inv", "fc2_w2", "fc2_inv")


class VitBlockH2(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in H2_FIELDS] + [("fc1_bound", C.c_float * 4), ("fc1_b2", C.c_void_p),
                                                       ("fc1_layout", C.c_int32), ("reserved", C.c_int32)]


# name -> (restype, argtypes); also the list the symbol-export test checks
SIGNATURES = {
    "anyloc_version": (C.c_int, []),
    "anyloc_last_error": (C.c_char_p, []),
    "anyloc_set_option": (C.c_int, [C.c_char_p, c_i64]),
    "anyloc_get_option": (C.c_int, [C.c_char_p, C.POINTER(c_i64)]),
    "anyloc_reset_options": (C.c_int, []),
    "anyloc_l2norm_rows": (C.c_int, [c_f32p, c_f32p, c_i64, c_i64, C.c_float, C.c_void_p]),
    "anyloc_preprocess_u8": (C.c_int, [C.c_void_p, c_i64, c_i64, c_i64, c_i64, c_i64, C.POINTER(C.c_float),
                                       C.POINTER(C.c_float), c_f32p, C.c_void_p]),
    "anyloc_resize_bicubic": (C.c_int, [c_f32p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32p,
                                        C.c_void_p]),
    "anyloc_pool_tokens": (C.c_int, [c_f32p, C.c_void_p, c_i64, c_i64, c_i64, C.c_int, C.c_float, c_f32p, C.c_void_p]),
    "anyloc_pca_gram_f64": (C.c_int, [c_f32p, c_i64, c_i64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "anyloc_pca_axes_f64": (C.c_int, [C.c_void_p, c_i64, c_i64, c_i64, c_f32p, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "anyloc_x3_bytes": (C.c_size_t, [c_i64, c_i64]),
    "anyloc_split_x3": (C.c_int, [c_f32p, c_i64, c_i64, c_i64, C.c_void_p, C.c_void_p]),
    "anyloc_gemm_nt_x6": (C.c_int, [C.c_void_p, C.c_void_p, c_f32p, c_f32p, c_i64, c_i64, c_i64, c_i64, C.c_void_p]),
    "anyloc_h2_bytes": (C.c_size_t, [c_i64, c_i64]),
    "anyloc_split_h2": (C.c_int, [c_f32p, c_i64, c_i64, c_i64, C.c_void_p, c_f32p, C.c_void_p]),
    "anyloc_gemm_nt_h3": (C.c_int, [C.c_void_p, c_f32p, C.c_void_p, c_f32p, c_f32p, c_f32p, c_i64, c_i64, c_i64, c_i64,
                                    C.c_void_p]),
    "anyloc_h3_lead_plan_check": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, c_i64, C.POINTER(C.c_uint32)]),
    "anyloc_gemm_nt": (C.c_int, [c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_f32p, c_i64,
                                 c_i64, c_i64, c_i64, C.c_void_p]),
    "anyloc_layernorm": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_i64, C.c_float, C.c_void_p]),
    "anyloc_attention": (C.c_int, [c_f32p, c_f32p, c_i64, c_i64, c_i64, c_i64, C.c_void_p]),
    "anyloc_attention_h3_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "anyloc_attention_h3": (C.c_int, [c_f32p, C.c_void_p, c_f32p, c_i64, c_i64, c_i64, c_i64, C.c_void_p, c_sz, C.c_void_p]),
    "anyloc_vlad_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i64]),
    "anyloc_vlad_auto_parts": (C.c_int, [c_i64, c_i64, c_i64, c_i64]),
    "anyloc_vlad_workspace_bytes_parts": (c_sz, [c_i64, c_i64, c_i64, c_i64, C.c_int32]),
    "anyloc_vlad_hard": (C.c_int, [c_f32p, c_i64p, c_i64, c_i64, c_i64, c_f32p, c_i64, C.c_uint,
                                   c_f32p, c_i64p, C.c_void_p, c_sz, C.c_void_p]),
    "anyloc_vlad_soft": (C.c_int, [c_f32p, c_i64p, c_i64, c_i64, c_i64, c_f32p, c_i64, C.c_float,
                                   C.c_uint, c_f32p, C.c_void_p, c_sz, C.c_void_p]),
    "anyloc_vlad_soft_weights": (C.c_int, [c_f32p, c_i64, c_i64, c_f32p, c_i64, C.c_float, c_f32p, C.c_void_p, c_sz,
                                           C.c_void_p]),
    "anyloc_vlad_residuals": (C.c_int, [c_f32p, c_i64, c_i64, c_f32p, c_i64, C.c_uint, c_f32p, C.c_void_p]),
    "anyloc_vlad_assigned": (C.c_int, [c_f32p, c_i64, c_i64, c_f32p, c_i64, c_i64p, c_f32p, C.c_uint, c_f32p,
                                       C.c_void_p, c_sz, C.c_void_p]),
    "anyloc_kmeans_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "anyloc_kmeans_step": (C.c_int, [c_f32p, c_i64, c_i64, c_f32p, c_i64, C.c_int, c_f32p, c_f32p,
                                     c_i64p, C.c_void_p, c_sz, C.c_void_p]),
    "anyloc_kmeans_update": (C.c_int, [c_f32p, c_f32p, c_f32p, c_i64, c_i64, c_f32p, C.c_void_p, C.c_void_p]),
    "anyloc_topk_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i64]),
    "anyloc_topk": (C.c_int, [c_f32p, c_i64, c_f32p, c_i64, c_i64, c_i64, C.c_int, C.c_uint, c_i64, c_f32p,
                              c_i64p, C.c_void_p, c_sz, C.c_void_p]),
    "anyloc_topk_path": (C.c_int, [c_i64, c_i64, c_i64]),
    "anyloc_topk_index_bytes": (c_sz, [c_i64, c_i64]),
    "anyloc_topk_index_build": (C.c_int, [c_f32p, c_i64, c_i64, C.c_void_p, c_sz, C.c_void_p]),
    "anyloc_topk_index_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i64]),
    "anyloc_topk_search_index": (C.c_int, [c_f32p, c_i64, C.c_void_p, c_i64, c_i64, c_i64, C.c_int, C.c_uint, c_i64, c_f32p,
                                           c_i64p, C.c_void_p, c_sz, C.c_void_p]),
    "anyloc_topk_search_index_rows": (C.c_int, [c_f32p, c_i64, c_f32p, C.c_void_p, c_i64, c_i64, c_i64, C.c_int, C.c_uint, c_i64,
                                                c_f32p, c_i64p, C.c_void_p, c_sz, C.c_void_p]),
    "anyloc_vit_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(VitConfig), c_f32p, c_f32p,
                                    c_f32p, C.POINTER(VitBlockWeights)]),
    "anyloc_vit_destroy": (None, [C.c_void_p]),
    "anyloc_vit_attach_x3": (C.c_int, [C.c_void_p, C.POINTER(VitBlockX3)]),
    "anyloc_vit_attach_h2": (C.c_int, [C.c_void_p, C.POINTER(VitBlockH2)]),
    "anyloc_vit_set_telemetry": (C.c_int, [C.c_void_p, c_f32p, C.c_int32]),
    "anyloc_vit_block_ffn_exact": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "anyloc_vit_workspace_bytes": (c_sz, [C.c_void_p, c_i64, c_i64, c_i64]),
    "anyloc_vit_forward": (C.c_int, [C.c_void_p, c_f32p, c_i64, c_i64, c_i64, c_f32p, C.c_int32,
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_uint, c_f32p,
                                     C.c_void_p, c_sz, C.c_void_p]),
    "anyloc_profile_enable": (C.c_int, [C.c_int]),
    "anyloc_profile_filter": (C.c_int, [C.c_char_p]),
    "anyloc_profile_reset": (C.c_int, []),
    "anyloc_profile_dump": (C.c_int, [C.c_char_p, c_sz]),
}

ABI_VERSION = 9          # include/anyloc_hip.h ANYLOC_ABI_VERSION the structs and signatures above were written for

_lib = None


def load():
    """Load (once) and return the ctypes library; raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise AnylocHipError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run `python -m anyloc_amd.build`). anyloc_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a symbol is missing
        fn.restype = res
        fn.argtypes = args
    have = lib.anyloc_version()
    if have != ABI_VERSION:
        # the .so is git-ignored and built separately: a stale one still exports every symbol but lays the structs out
        # differently (anyloc_vit_block_h2 grew in ABI 4) -- refuse it instead of handing the GPU garbage pointers
        raise AnylocHipError(f"{LIB_PATH} has ABI version {have}, this package expects {ABI_VERSION}: "
                             "rebuild it (`python -m anyloc_amd.build --force`)")
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        msg = load().anyloc_last_error().decode("utf-8", "replace")
        raise AnylocHipError(f"{what} failed with status {status}: {msg}")


def require_gpu():
    if not torch.cuda.is_available():
        raise AnylocHipError("anyloc_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                             "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


# ---- caller-owned workspace: one growing byte buffer per device ---------------
_workspaces = {}


def workspace(nbytes, device, tag="default"):
    """Growing scratch buffer per (device, stream, tag): two streams never share one (kernels of different streams
    would race on it), consecutive calls on one stream reuse it (stream order serialises them)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream().cuda_stream, tag)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            del _workspaces[key]
            buf = None
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def release_workspaces():
    _workspaces.clear()
