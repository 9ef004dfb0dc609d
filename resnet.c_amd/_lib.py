"""ctypes binding of librn_hip.so (the C-ABI of include/rn_hip.h).

The library is the product: if it is missing, or an entry point is missing
from it, loading fails loudly.  There is no CPU fallback anywhere in this
package.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# RN_HIP_LIB: another build of the same library (A/B timing of kernel changes)
LIB_PATH = os.environ.get("RN_HIP_LIB") or os.path.join(_HERE, "librn_hip.so")

RN_OK = 0
RN_ERR_INVALID, RN_ERR_HIP, RN_ERR_IO, RN_ERR_NOMEM, RN_ERR_UNSUPPORTED = 1, 2, 3, 4, 5
RN_LAYOUT_NCHW, RN_LAYOUT_NHWC = 0, 1
RN_FWD_REFERENCE_OPS, RN_FWD_FUSED = 0, 1
RN_DTYPE_F32, RN_DTYPE_BF16 = 0, 1

u64 = c_uint64
fptr = c_void_p  # device pointers travel as plain addresses


class ConvSecond(ctypes.Structure):
    _fields_ = [("inp", c_void_p), ("in_channels", c_uint64), ("H", c_uint64), ("W", c_uint64),
                ("stride", c_uint64)]


class Epilogue(ctypes.Structure):
    _fields_ = [("scale", c_void_p), ("shift", c_void_p), ("residual", c_void_p),
                ("relu", c_int)]


# name -> (restype, argtypes); every symbol include/rn_hip.h declares
SIGNATURES = {
    "rn_ctx_create": (c_int, [POINTER(c_void_p), c_int, c_void_p]),
    "rn_ctx_destroy": (c_int, [c_void_p]),
    "rn_ctx_set_layout": (c_int, [c_void_p, c_int]),
    "rn_ctx_get_layout": (c_int, [c_void_p]),
    "rn_ctx_set_sync_each_op": (c_int, [c_void_p, c_int]),
    "rn_ctx_set_deferred": (c_int, [c_void_p, c_int]),
    "rn_ctx_get_deferred": (c_int, [c_void_p]),
    "rn_flush": (c_int, [c_void_p]),
    "rn_observe": (c_int, [c_void_p, c_void_p]),
    "rn_ctx_deferred_stats": (c_int, [c_void_p] + [POINTER(u64)] * 5),
    "rn_ctx_set_weight_cache": (c_int, [c_void_p, c_int]),
    "rn_ctx_launch_count": (u64, [c_void_p]),
    "rn_conv_tile_candidates": (c_int, []),
    "rn_ctx_set_conv_tile": (c_int, [c_void_p, c_int]),
    "rn_ctx_set_split_k": (c_int, [c_void_p, c_int]),
    "rn_ctx_set_stem_items": (c_int, [c_void_p, c_int]),
    "rn_ctx_set_xcd_groups": (c_int, [c_void_p, c_int]),
    "rn_ctx_set_nchw_taps": (c_int, [c_void_p, c_int]),
    "rn_ctx_set_debug_stamps": (c_int, [c_void_p, c_void_p]),
    "rn_ctx_stream": (c_void_p, [c_void_p]),
    "rn_ctx_device": (c_int, [c_void_p]),
    "rn_sync": (c_int, [c_void_p]),
    "rn_last_error": (c_char_p, [c_void_p]),
    "rn_status_string": (c_char_p, [c_int]),
    "rn_device_count": (c_int, [POINTER(c_int)]),
    "rn_device_locality": (c_int, [c_int, c_char_p, u64, POINTER(c_int), c_char_p, u64]),
    "rn_version": (c_char_p, []),
    "rn_malloc": (c_int, [c_void_p, POINTER(c_void_p), u64]),
    "rn_free": (c_int, [c_void_p, c_void_p]),
    "rn_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_void_p, u64]),
    "rn_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_void_p, u64]),
    "rn_memcpy_d2d": (c_int, [c_void_p, c_void_p, c_void_p, u64]),
    "rn_memset": (c_int, [c_void_p, c_void_p, c_int, u64]),
    "rn_load_f32_file": (c_int, [c_void_p, c_char_p, POINTER(c_void_p), POINTER(u64)]),
    "rn_save_f32_file": (c_int, [c_void_p, c_char_p, c_void_p, u64]),
    "rn_event_create": (c_int, [c_void_p, POINTER(c_void_p)]),
    "rn_event_destroy": (c_int, [c_void_p]),
    "rn_event_record": (c_int, [c_void_p, c_void_p]),
    "rn_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "rn_conv_output_size": (u64, [u64, u64, u64, u64]),
    "rn_conv2d_forward": (c_int, [c_void_p, fptr, fptr, fptr] + [u64] * 10),
    "rn_maxpool2d_forward": (c_int, [c_void_p, fptr, fptr] + [u64] * 9),
    "rn_avgpool2d_forward": (c_int, [c_void_p, fptr, fptr] + [u64] * 9),
    "rn_linear_forward": (c_int, [c_void_p, fptr, fptr, fptr, fptr] + [u64] * 3),
    "rn_relu_forward": (c_int, [c_void_p, fptr, fptr, u64]),
    "rn_batchnorm2d_forward": (c_int, [c_void_p] + [fptr] * 6 + [u64] * 3),
    "rn_add_forward": (c_int, [c_void_p, fptr, fptr, fptr, u64]),
    "rn_argmax_forward": (c_int, [c_void_p, fptr, c_void_p, u64, u64]),
    "rn_nchw_to_nhwc": (c_int, [c_void_p, fptr, fptr] + [u64] * 4),
    "rn_nhwc_to_nchw": (c_int, [c_void_p, fptr, fptr] + [u64] * 4),
    "rn_conv2d_input_channels": (u64, [u64]),
    "rn_nchw_to_nhwc_pad": (c_int, [c_void_p, fptr, fptr] + [u64] * 5),
    "rn_conv2d_packed_weight_numel": (u64, [u64, u64, u64]),
    "rn_conv2d_pack_weight": (c_int, [c_void_p, fptr, fptr, u64, u64, u64]),
    "rn_batchnorm2d_fold": (c_int, [c_void_p] + [fptr] * 6 + [u64]),
    "rn_conv2d_nhwc_forward": (c_int, [c_void_p, fptr, fptr, fptr] + [u64] * 10
                               + [POINTER(Epilogue)]),
    "rn_conv2d_packed_weight_numel_dt": (u64, [c_int, u64, u64, u64]),
    "rn_conv2d_pack_weight_dt": (c_int, [c_void_p, c_int, fptr, fptr, u64, u64, u64]),
    "rn_nchw_to_nhwc_pad_dt": (c_int, [c_void_p, c_int, fptr, fptr] + [u64] * 6),
    "rn_conv2d_nhwc_forward_dt": (c_int, [c_void_p, c_int, c_int, fptr, fptr, fptr] + [u64] * 10
                                  + [POINTER(Epilogue)]),
    "rn_conv2d_packed_weight_numel_exact": (u64, [u64, u64, u64]),
    "rn_conv2d_pack_weight_exact": (c_int, [c_void_p, fptr, fptr, u64, u64, u64]),
    "rn_conv2d_nhwc_exact_forward": (c_int, [c_void_p, fptr, fptr, fptr] + [u64] * 9
                                     + [POINTER(Epilogue)]),
    "rn_conv2d_packed_pair_weight_numel": (u64, [u64, u64, u64, u64]),
    "rn_conv2d_pack_weight_pair_dt": (c_int, [c_void_p, c_int, fptr, fptr, fptr, fptr, fptr]
                                      + [u64] * 4),
    "rn_conv2d_nhwc_pair_forward_dt": (c_int, [c_void_p, c_int, c_int, fptr, fptr, fptr] + [u64] * 10
                                       + [POINTER(ConvSecond), POINTER(Epilogue)]),
    "rn_model_set_pair_fusion": (c_int, [c_void_p, c_int]),
    "rn_model_set_stem_exact": (c_int, [c_void_p, c_int]),
    "rn_stem_pool_packed_weight_numel": (u64, [c_int]),
    "rn_stem_pool_pack_weight_dt": (c_int, [c_void_p, c_int, fptr, fptr, u64]),
    "rn_stem_pool_forward_dt": (c_int, [c_void_p, c_int, fptr, fptr, fptr, fptr, fptr, c_int, u64, u64, u64]),
    "rn_stem_pool_nchw_forward_dt": (c_int, [c_void_p, c_int, fptr, fptr, fptr, fptr, fptr, c_int, u64, u64, u64, u64]),
    "rn_stem_conv_pool_nchw_forward": (c_int, [c_void_p] + [fptr] * 6 + [u64] * 4),
    "rn_conv_chain_forward_dt": (c_int, [c_void_p, c_int] + [fptr] * 10 + [u64] * 4),
    "rn_conv_chain_pair_forward_dt": (c_int, [c_void_p, c_int] + [fptr] * 9 + [u64] * 5),
    "rn_model_set_chain": (c_int, [c_void_p, c_int]),
    "rn_model_set_stem_pool_fusion": (c_int, [c_void_p, c_int]),
    "rn_model_set_streams": (c_int, [c_void_p, c_int]),
    "rn_model_get_streams": (c_int, [c_void_p]),
    "rn_model_parts": (c_int, [c_void_p, u64]),
    "rn_model_set_front_parts": (c_int, [c_void_p, c_int]),
    "rn_maxpool2d_nhwc_forward_dt": (c_int, [c_void_p, c_int, fptr, fptr] + [u64] * 9),
    "rn_avgpool2d_nhwc_forward_dt": (c_int, [c_void_p, c_int, fptr, fptr] + [u64] * 9),
    "rn_model_create": (c_int, [c_void_p, POINTER(c_void_p), c_int]),
    "rn_model_set_dtype": (c_int, [c_void_p, c_int]),
    "rn_model_destroy": (c_int, [c_void_p]),
    "rn_model_set_tensor": (c_int, [c_void_p, c_char_p, c_void_p, u64]),
    "rn_model_load_dir": (c_int, [c_void_p, c_char_p]),
    "rn_model_finalize": (c_int, [c_void_p]),
    "rn_model_tensor_key": (c_char_p, [c_void_p, u64, POINTER(u64)]),
    "rn_model_forward": (c_int, [c_void_p, fptr, u64, fptr, c_int]),
    "rn_model_tune": (c_int, [c_void_p, fptr, u64, fptr, c_int]),
    "rn_model_export_tuning": (c_int, [c_void_p, POINTER(u64), u64, POINTER(u64)]),
    "rn_model_import_tuning": (c_int, [c_void_p, POINTER(u64), u64]),
    "rn_model_set_profiling": (c_int, [c_void_p, c_int]),
    "rn_model_profile_count": (u64, [c_void_p]),
    "rn_model_profile_get": (c_int, [c_void_p, u64, POINTER(c_char_p), POINTER(c_char_p),
                                     POINTER(c_float), POINTER(c_double), POINTER(c_double)]),
    "rn_model_activation_bytes": (u64, [c_void_p]),
    "rn_model_capture": (c_int, [c_void_p, fptr, u64, fptr, c_int, POINTER(c_void_p)]),
    "rn_graph_launch": (c_int, [c_void_p]),
    "rn_graph_destroy": (c_int, [c_void_p]),
    "rn_graph_node_count": (u64, [c_void_p]),
    "rn_pipeline_create": (c_int, [c_void_p, POINTER(c_void_p), u64, c_int]),
    "rn_pipeline_destroy": (c_int, [c_void_p]),
    "rn_pipeline_input_buffer": (c_int, [c_void_p, POINTER(c_void_p)]),
    "rn_pipeline_submit": (c_int, [c_void_p, c_void_p]),
    "rn_pipeline_collect": (c_int, [c_void_p, c_void_p]),
    "rn_pipeline_in_flight": (u64, [c_void_p]),
    "rn_pipeline_submit_n": (c_int, [c_void_p, c_void_p, u64]),
    "rn_pipeline_collect_n": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(u64)]),
    "rn_shard_bounds": (None, [u64, c_int, c_int, POINTER(u64), POINTER(u64)]),
    "rn_shard_create": (c_int, [POINTER(c_void_p), POINTER(c_int), c_int, c_int]),
    "rn_shard_destroy": (c_int, [c_void_p]),
    "rn_shard_count": (c_int, [c_void_p]),
    "rn_shard_last_error": (c_char_p, [c_void_p]),
    "rn_shard_set_tensor": (c_int, [c_void_p, c_char_p, c_void_p, u64]),
    "rn_shard_load_dir": (c_int, [c_void_p, c_char_p]),
    "rn_shard_set_dtype": (c_int, [c_void_p, c_int]),
    "rn_shard_finalize": (c_int, [c_void_p]),
    "rn_shard_forward": (c_int, [c_void_p, c_void_p, u64, c_void_p, c_void_p, c_int]),
    "rn_shard_tune": (c_int, [c_void_p, c_void_p, u64, c_int]),
    "rn_shard_model": (c_void_p, [c_void_p, c_int]),
    "rn_shard_placement": (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_int), c_char_p, u64]),
    "rn_shard_stream_open": (c_int, [c_void_p, u64, c_int]),
    "rn_shard_stream_buffer": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(u64), POINTER(u64)]),
    "rn_shard_submit": (c_int, [c_void_p, c_void_p]),
    "rn_shard_collect": (c_int, [c_void_p, c_void_p, c_void_p]),
    "rn_shard_in_flight": (c_int, [c_void_p]),
    "rn_shard_stream_close": (c_int, [c_void_p]),
}

_lib = None


class RnError(RuntimeError):
    """A C-ABI call returned a non-zero status (the reference would have aborted:
    gpuAssert, cuda/helpers.cuh:13-22)."""

    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: status {status}" + (f" ({detail})" if detail else ""))


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C resnet.c_amd/csrc`). There is no CPU fallback.")
        # torch ships its own libamdhip64.so.7; importing it first makes this library
        # bind to the same HIP runtime instance instead of loading a second one.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is plumbing, not required
            pass
        L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def source_digest() -> str:
    """sha256 (16 hex digits) over the sources librn_hip.so is built from: what a measurement of
    the kernels (a PMC pass, a trace) is stamped with, so that a later build can tell whether
    the figure still describes it.  (The GPU box has no .git; a commit id would also change with
    every documentation commit.)"""
    import glob
    import hashlib

    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                   glob.glob(os.path.join(_HERE, "csrc", "*.c")) + glob.glob(os.path.join(_HERE, "csrc", "Makefile")) +
                   [os.path.join(os.path.dirname(_HERE), "include", "rn_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def check(status: int, where: str, ctx=None) -> None:
    if status != RN_OK:
        detail = ""
        if ctx:
            msg = lib().rn_last_error(ctx)
            detail = msg.decode() if msg else ""
        if not detail:
            detail = lib().rn_status_string(status).decode()
        raise RnError(status, where, detail)
