"""ctypes binding of lib/libmi355x_match.so (the C ABI in include/mi355x_match.h).

PyTorch is plumbing only: it owns device memory and the HIP stream; every compute call
goes through the C ABI with raw device pointers.  There is no CPU or eager fallback: if the
library is missing or a tensor is not on the GPU the call raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_size_t, c_uint32, c_void_p

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# MI355X_MATCH_LIB: development hook (A/B timing of two builds on one box); the default is the in-tree build
LIB_PATH = os.environ.get("MI355X_MATCH_LIB") or os.path.join(_PKG, "lib", "libmi355x_match.so")

# name -> argtypes (return type is int unless listed in _RESTYPE); mirrors include/mi355x_match.h
SIGNATURES = {
    "mi_abi_version": [],
    "mi_sinkhorn_dots_schedule": [c_void_p, c_int, c_int, c_int, c_int],
    "mi_sinkhorn_dots_set_schedule": [c_void_p, c_int],
    "mi_error_string": [c_int],
    "mi_release_stream_resources": [c_void_p],
    "mi_corner_response": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "mi_corner_response_u8": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "mi_convert_u8_f32": [c_void_p, ctypes.c_longlong, c_void_p, c_void_p],
    "mi_corner_response_balanced": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "mi_corner_response_pair": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "mi_nms_mask": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "mi_candidate_layout": [c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)],
    "mi_nms_candidates": [c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p],
    "mi_select_candidates": [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p],
    "mi_topk_keypoints": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "mi_bad_plan_bytes": [c_int],
    "mi_bad_plan_build": [c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "mi_sparse_bad": [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_int,
                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "mi_sparse_bad_u8": [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_int,
                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "mi_sparse_bad_pair": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                           c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "mi_bad_dense": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p],
    "mi_bad_dense_oriented": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p,
                              c_void_p],
    "mi_gather_descriptors": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p],
    "mi_angle_map": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "mi_angle_at_keypoints": [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "mi_sparse_bad_oriented": [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_int, c_int, c_float, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    "mi_angle_at_keypoints_pair": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                   c_void_p],
    "mi_sparse_bad_oriented_pair": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_float, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    "mi_cost_logscores_bits": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_double, c_void_p, c_int, c_void_p],
    "mi_cost_logscores_f32": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_double, c_void_p, c_int, c_void_p],
    "mi_sinkhorn_workspace_bytes": [c_int, c_int, c_int],
    "mi_sinkhorn": [c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                    c_size_t, c_void_p],
    "mi_cost_dots_bits": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                          c_void_p],
    "mi_sinkhorn_dots_workspace_bytes": [c_int, c_int, c_int],
    "mi_sinkhorn_dots": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_double, c_int,
                         c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p],
    "mi_sinkhorn_dots_status_word": [c_void_p, c_int, c_int, c_int],
    "mi_match_filters": [c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p],
    "mi_match_filter_masks": [c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p],
    "mi_mnn_extract": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p,
                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "mi_mnn_duals_workspace_bytes": [c_int, c_int, c_int],
    "mi_mnn_from_duals": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                          c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "mi_mnn_from_duals_dots": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_double, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_int, c_float, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p],
    "mi_core_maxima": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "mi_normalise_keypoints": [c_void_p, ctypes.c_longlong, c_void_p, c_void_p, c_void_p],
    "mi_essential_matrix_workspace_bytes": [c_int, c_int, c_int, c_int],
    "mi_essential_matrix": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                            c_void_p, c_void_p, c_size_t, c_void_p],
    "mi_essential_matrix_dots": [c_void_p, c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                 c_void_p],
    "mi_fast_score": [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p],
    "mi_dog_responses": [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "mi_akaze_diffuse": [c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p],
    "mi_akaze_scale_fused": [c_int, c_int],
    "mi_akaze_scale": [c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p,
                       c_void_p],
    "mi_akaze_scale_sets": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_int, c_void_p, c_void_p,
                            c_void_p, c_void_p],
    "mi_akaze_scale_select": [c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_int, c_void_p, c_void_p,
                              c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "mi_akaze_orientation_from_attain": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                         c_void_p],
    "mi_akaze_orientation_select": [c_void_p, c_size_t, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int,
                                    c_void_p, c_void_p, c_void_p],
    "mi_akaze_hessian_scores": [c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p],
    "mi_akaze_combine": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "mi_akaze_orientation_at_keypoints": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                          c_void_p],
}


class MatchParams(ctypes.Structure):
    """mi_match_params (include/mi355x_match.h): the hyper-parameters of the whole path for mi_match_pairs."""
    _fields_ = [("block_size", c_int), ("nms_radius", c_int), ("max_keypoints", c_int), ("score_threshold", c_float),
                ("border_margin", c_int), ("num_pairs", c_int), ("pair_geom", c_void_p), ("pair_thr", c_void_p),
                ("bad_plan", c_void_p), ("normalize_descriptors", c_int), ("epsilon", c_double),
                ("unused_score", c_double), ("sinkhorn_iterations", c_int), ("max_matches", c_int),
                ("match_threshold", c_float), ("flags", c_int)]


SIGNATURES["mi_match_pairs_workspace_bytes"] = [c_int, c_int, c_int, ctypes.POINTER(MatchParams)]
SIGNATURES["mi_match_pairs"] = [c_void_p, c_void_p, c_int, c_int, c_int, ctypes.POINTER(MatchParams), c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
# include/mi355x_match_debug.h: test / tooling hooks -- exported only by lib/libmi355x_match_debug.so (debug_library()),
# not part of the product ABI; nothing in this package calls them
DEBUG_SIGNATURES = {"mi_debug_set": [c_int, c_int], "mi_debug_topk_stamps": [c_void_p], "mi_debug_clock_probe": [c_void_p],
                    "mi_debug_bad_plan_passes": [c_void_p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)],
                    "mi_debug_sinkhorn_dots_form": [c_int, c_int, c_int, c_int, c_int, c_int],
                    "mi_debug_tuner_script": [c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int)],
                    "mi_debug_akaze_math_check": [c_int, c_float, c_uint32, c_uint32, c_void_p, c_void_p, c_void_p]}
SIGNATURES["mi_match_pairs_u8"] = SIGNATURES["mi_match_pairs"]
_RESTYPE = {"mi_essential_matrix_workspace_bytes": c_size_t, "mi_sinkhorn_dots_status_word": c_void_p, "mi_match_pairs_workspace_bytes": c_size_t, "mi_error_string": c_char_p, "mi_sinkhorn_workspace_bytes": c_size_t, "mi_bad_plan_bytes": c_size_t,
            "mi_sinkhorn_dots_workspace_bytes": c_size_t, "mi_mnn_duals_workspace_bytes": c_size_t}

MI_BAD_RAW, MI_BAD_SOFT, MI_BAD_HARD = 0, 1, 2
MI_DIST_L2, MI_DIST_L1 = 0, 1

DEBUG_LIB_PATH = os.path.join(_PKG, "lib", "libmi355x_match_debug.so")

_lib = None            # the library every call() goes through: the product, unless inside debug_library()
_product = None
_debug = None


def _open(path: str, signatures: dict) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP library is not built. Run "
            "`python -m onnx_image_processing_amd.build` (there is no CPU fallback)."
        )
    lib = ctypes.CDLL(path)
    for name, argtypes in signatures.items():
        fn = getattr(lib, name)            # AttributeError if the ABI and the binding diverge
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, c_int)
    return lib


def load() -> ctypes.CDLL:
    """The library in use (the product library unless a debug_library() block is active), loaded once; raises loudly
    when it has not been built."""
    global _lib, _product
    if _lib is None:
        if _product is None:
            _product = _open(LIB_PATH, SIGNATURES)
        _lib = _product
    return _lib


class debug_library:
    """with debug_library() as lib: ... -- tests / tools only.  Inside the block every call of this package goes through
    lib/libmi355x_match_debug.so (same sources, -DMI_DEBUG_HOOKS), whose mi_debug_* hooks select between equivalent
    kernel implementations; on exit the hooks are back at their defaults and the product library is in use again."""

    def __enter__(self) -> ctypes.CDLL:
        global _lib, _debug
        load()
        if _debug is None:
            _debug = _open(DEBUG_LIB_PATH, {**SIGNATURES, **DEBUG_SIGNATURES})
        self._outer = _lib
        _lib = _debug
        return _debug

    def __exit__(self, *exc) -> None:
        global _lib
        _debug.mi_debug_set(0, 0)                  # key 0: every selector back to the product's value, probes off
        _lib = self._outer


def use_debug_library() -> ctypes.CDLL:
    """tools/ only: switch this process to the debug library for good (debug_library() without the exit)."""
    return debug_library().__enter__()


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().mi_error_string(code)
        raise RuntimeError(f"{what} failed ({code}): {msg.decode() if msg else '?'}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def dev(t: torch.Tensor, dtype: torch.dtype, what: str) -> int:
    """Device pointer of a contiguous GPU tensor of the expected dtype."""
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} must live on the GPU (got device {t.device}); this package has no CPU path"
        )
    if t.dtype != dtype:
        raise RuntimeError(f"{what} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{what} must be contiguous")
    return t.data_ptr()


_timers: dict | None = None


_timed_names = None


def enable_timing(on: bool, only=None) -> None:
    """Bracket C-ABI calls with HIP events on the launch stream (bench.py's per-kernel clock): every call, or
    only the entry points named in `only`.  Off by default: no events, no overhead."""
    global _timers, _timed_names
    _timers = {} if on else None
    _timed_names = set(only) if (on and only) else None


def timings_ms() -> dict:
    """name -> list of elapsed milliseconds, one per call (synchronises)."""
    torch.cuda.synchronize()
    return {k: [s.elapsed_time(e) for s, e in v] for k, v in (_timers or {}).items()}


def call(name: str, *args) -> None:
    fn = getattr(load(), name)
    if _timers is None or (_timed_names is not None and name not in _timed_names):
        check(fn(*args), name)
        return
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()                     # torch's current stream == the stream passed to the ABI
    rc = fn(*args)
    end.record()
    check(rc, name)
    _timers.setdefault(name, []).append((start, end))
