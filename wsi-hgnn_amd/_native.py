"""ctypes binding of csrc/libwsi_hgnn.so (the C-ABI declared in include/wsi_hgnn.h).

There is NO fallback: if the shared object is missing or a symbol is absent, importing fails
loudly — the product path never silently drops to eager PyTorch (or to the CPU oracle).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libwsi_hgnn.so")

WSI_GEMM_NT, WSI_GEMM_NN, WSI_GEMM_TN = 0, 1, 2
WSI_EPI_BIAS, WSI_EPI_ACCUMULATE, WSI_EPI_SCALE_GATE, WSI_EPI_GELU, WSI_EPI_ADD_R, WSI_EPI_R_1MG, WSI_EPI_MUL_M, WSI_EPI_BACKGROUND, WSI_EPI_DROPOUT = 1, 2, 4, 8, 16, 32, 64, 128, 256
WSI_EPI_GATED_SKIP = WSI_EPI_BIAS | WSI_EPI_SCALE_GATE | WSI_EPI_ADD_R | WSI_EPI_R_1MG
WSI_RED_SUM, WSI_RED_MEAN, WSI_RED_MAX = 0, 1, 2
WSI_GEMM_MAX_GROUPS = 24
WSI_ABI_VERSION = 23
WSI_GEMM_FP32, WSI_GEMM_BF16X6, WSI_GEMM_FP16X3, WSI_GEMM_AUTO = 0, 1, 2, 3
WSI_ATTN_XCD_CONTIGUOUS = 1


class AttnPool(ctypes.Structure):
    """wsi_attn_pool_t (include/wsi_hgnn.h)."""
    _fields_ = [("row_seg", ctypes.c_void_p), ("segs_per_type", ctypes.c_int32), ("n_types", ctypes.c_int32),
                ("y", ctypes.c_void_p), ("g_row", ctypes.c_void_p), ("omg", ctypes.c_void_p),
                ("r_out", ctypes.c_void_p), ("ldr", ctypes.c_int64), ("ctab", ctypes.c_void_p), ("ctab_ready", ctypes.c_int32),
                ("h", ctypes.c_void_p), ("ldh", ctypes.c_int64), ("beta", ctypes.c_void_p),
                ("gtab", ctypes.c_void_p), ("edge_seg", ctypes.c_void_p), ("seg_dst", ctypes.c_void_p)]


WSI_ATTN_MAX_SPANS = 200


class AttnTiles(ctypes.Structure):
    """wsi_attn_tiles_t (include/wsi_hgnn.h)."""
    _fields_ = [("part_ptr", c_int32 * 9), ("begin", c_int32 * WSI_ATTN_MAX_SPANS), ("end", c_int32 * WSI_ATTN_MAX_SPANS)]


class AdamTensor(ctypes.Structure):
    """wsi_adam_tensor_t (include/wsi_hgnn.h)."""
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p), ("n", ctypes.c_int64)]


class GemmGroup(ctypes.Structure):
    """struct wsi_gemm_group (include/wsi_hgnn.h)."""
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p),
        ("bias", c_void_p), ("R", c_void_p), ("gate", c_void_p),
        ("B1", c_void_p), ("B2", c_void_p),
        ("lda", c_int64), ("ldb", c_int64), ("ldc", c_int64), ("ldr", c_int64),
        ("M", c_int32), ("N", c_int32), ("K", c_int32), ("b_chunk", c_int32),
        ("Mm", c_void_p), ("ldm", c_int64),
        ("colsum_out", c_void_p),
        ("a_absmax", c_void_p), ("c_absmax", c_void_p),
        ("a_absmax_parts", c_int32), ("c_absmax_parts", c_int32), ("c_absmax_first", c_int32), ("reserved", c_int32),
        ("drop_seed", ctypes.c_uint32), ("drop_threshold", ctypes.c_uint32), ("drop_scale", c_float),
        ("drop_row0", c_int32), ("drop_cols", c_int32), ("drop_col0", c_int32),
        ("drop_seed_base", c_void_p),
        ("c_colmax", c_void_p), ("c_colsum", c_void_p), ("c_col_ld", c_int64),
        ("a_colmax", c_void_p), ("a_colsum", c_void_p), ("b_colmax", c_void_p),
        ("a_col_ld", c_int64), ("b_col_ld", c_int64), ("a_col_parts", c_int32), ("b_col_parts", c_int32),
        ("b_packed", c_void_p),
    ]


def gemm_absmax_parts(n_cols: int) -> int:
    """WSI_GEMM_ABSMAX_PARTS: slots per row a group of ``n_cols`` output columns writes into c_absmax."""
    return 2 * ((int(n_cols) + 127) // 128)


EXPORTS = {
    "wsi_abi_version": (ctypes.c_int, []),
    "wsi_last_error": (c_char_p, []),
    "wsi_heat_attn_fwd": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                         c_int32, c_int32, c_int32,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                         c_void_p, c_void_p,
                                         c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wsi_heat_attn_scores_fwd": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_void_p]),
    "wsi_heat_attn_tiled_fwd": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                               c_int32, c_int32, c_int32, c_int32, c_int32,
                                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(AttnTiles), c_int32,
                                               c_void_p, c_void_p,
                                               c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wsi_heat_attn_tiled_bwd": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                               c_int32, c_int32, c_int32, c_int32, c_int32,
                                               c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, POINTER(AttnTiles), c_int32,
                                               c_void_p, c_void_p,
                                               c_void_p, c_int64, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                               c_void_p, c_void_p, c_void_p]),
    "wsi_heat_pool_gtab": (ctypes.c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                          c_int32, c_int32, c_void_p, c_void_p]),
    "wsi_heat_pool_coeff": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "wsi_heat_attn_bwd": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                         c_int32, c_int32, c_int32, c_int32, c_int32,
                                         c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_int32, c_void_p, c_int32,
                                         c_void_p, c_void_p,
                                         c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wsi_context_create": (ctypes.c_int, [POINTER(c_void_p)]),
    "wsi_context_destroy": (None, [c_void_p]),
    "wsi_gemm_workspace_bytes": (c_int64, [c_int32, c_int32, POINTER(GemmGroup), c_int32]),
    "wsi_gemm_kernel_precision": (c_int32, [c_int32, c_int32, POINTER(GemmGroup), c_int32]),
    "wsi_row_absmax": (ctypes.c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "wsi_gemm_grouped": (ctypes.c_int, [c_int32, c_int32, c_int32, POINTER(GemmGroup), c_int32, c_void_p, c_int64, c_void_p]),
    "wsi_gemm_writes_colstats": (c_int32, [c_int32, c_int32, POINTER(GemmGroup), c_int32]),
    "wsi_gemm_packed_b_bytes": (c_int64, [c_int32, c_int32]),
    "wsi_gemm_pack_b": (ctypes.c_int, [c_int32, POINTER(GemmGroup), c_int32, c_void_p]),
    "wsi_col_stats_parts": (c_int32, [c_int32]),
    "wsi_col_stats": (ctypes.c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
    "wsi_col_absmax_workspace_bytes": (c_int64, [c_int32, c_int32]),
    "wsi_col_absmax": (ctypes.c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
    "wsi_segment_reduce_fwd": (ctypes.c_int, [c_void_p, c_int64, c_int32, c_int32,
                                              c_void_p, c_int32, c_void_p, c_int32,
                                              c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "wsi_segment_dot_diff": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int32,
                                            c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "wsi_segment_weighted_sums": (ctypes.c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int32,
                                                 c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "wsi_gate_grad": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int32,
                                     c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "wsi_pool_factors": (ctypes.c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                        c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wsi_pool_tmean": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wsi_pool_bwd_prep": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                         c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wsi_pool_bwd_bias": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wsi_gemm_small_pair": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p]),
    "wsi_plan_assemble": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_void_p]),
    "wsi_cross_entropy": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wsi_adam_step": (ctypes.c_int, [c_void_p, c_int32, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                     c_int64, c_void_p]),
    "wsi_layernorm_fwd": (ctypes.c_int, [c_void_p, c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_int64, c_void_p, c_void_p]),
    "wsi_layernorm_bwd": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "wsi_dropout_apply": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32,
                                         ctypes.c_uint32, c_void_p, ctypes.c_uint32, c_float, c_void_p]),
    "wsi_gelu_fwd": (ctypes.c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "wsi_gelu_bwd": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "wsi_spmm_sum": (ctypes.c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int32, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "wsi_row_sqnorm": (ctypes.c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "wsi_knn_select": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "wsi_pair_stats": (ctypes.c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int32, c_int32,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "wsi_csr_gather_max_fwd": (ctypes.c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "wsi_csr_gather_max_bwd": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_int64, c_void_p]),
    "wsi_asap_attend_fwd": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_float,
                                           c_void_p, c_void_p, c_int64, c_void_p]),
    "wsi_asap_attend_bwd": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int64,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "wsi_graph_topk": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wsi_stas": (ctypes.c_int, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wsi_segment_reduce_bwd": (ctypes.c_int, [c_void_p, c_int64, c_int32, c_int32,
                                              c_void_p, c_void_p, c_int32, c_void_p, c_int32,
                                              c_void_p, c_void_p, c_int64, c_void_p]),
}

_lib = None
_ablate = False


def use_measurement_library() -> None:
    """tools/ only: bind the -DWSI_ABLATE flavour of the library (csrc/libwsi_hgnn_ablate.so: the dominant kernel's ablation variants and the
    WSI_* environment knobs of csrc/common.h::knob compiled in) instead of the product library.  Must be called before the first ``load()``;
    the package itself never calls it - the product library reads no environment variable."""
    global _ablate, LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_measurement_library(): the product library is already loaded in this process")
    from .build import LIB_ABLATE
    _ablate, LIB_PATH = True, LIB_ABLATE


def load() -> ctypes.CDLL:
    """Load the shared object and bind every declared symbol; raise RuntimeError when impossible."""
    global _lib
    if _lib is not None:
        return _lib
    try:   # (re)build in-tree when the shared object is missing or older than its sources (hipcc cross-compiles anywhere)
        from .build import build_native
        build_native(force=False, verbose=True, ablate=_ablate)
    except Exception as exc:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing and could not be built ({exc}). Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'`. There is no PyTorch/CPU fallback for the hot path.") from exc
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: the HIP extension is not built; there is no PyTorch/CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:  # pragma: no cover
            raise RuntimeError(f"{LIB_PATH} does not export {name}; rebuild it") from exc
        fn.restype = res
        fn.argtypes = args
    if lib.wsi_abi_version() != WSI_ABI_VERSION:
        raise RuntimeError(f"libwsi_hgnn.so ABI {lib.wsi_abi_version()} != expected {WSI_ABI_VERSION}; rebuild it")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().wsi_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


_contexts: dict = {}


def context() -> int:
    """The caller-owned wsi_context_t of the current device (side stream of the hub kernels), created on first use and kept
    for the life of the process: the LIBRARY holds no state, this host-side mirror owns one context per device."""
    dev = torch._C._cuda_getDevice()            # (torch.cuda.current_device() re-checks the lazy init on every call: ~10 us)
    h = _contexts.get(dev)
    if h is None:
        out = c_void_p()
        check(load().wsi_context_create(ctypes.byref(out)), "wsi_context_create")
        h = _contexts[dev] = out.value
    return h


def stream() -> int:
    """Raw hipStream_t of torch's current stream on the current device (the C accessors: the torch.cuda wrappers cost ~15 us a call,
    paid once per kernel launch)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def ptr(t, byte_offset: int = 0):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr() + byte_offset


def require_cuda(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("wsi_hgnn_amd ops run on the GPU only (HIP kernels); got a CPU tensor. "
                               "The CPU oracle under oracle/ is test infrastructure, not a fallback.")
