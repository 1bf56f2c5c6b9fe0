"""Functional layer over the C ABI: shape checks, output/workspace allocation, launch.

Everything here is asynchronous on torch's current HIP stream.  Outputs are fresh torch
tensors on the input's device (the reference's ownership convention); workspaces come from
torch's caching allocator, so a steady-state step performs no hipMalloc.
"""
from __future__ import annotations

import torch

from . import _native as N

F32 = torch.float32


U8 = torch.uint8


def convert_u8(image: torch.Tensor) -> torch.Tensor:
    """uint8 device tensor -> float32 of the same shape (`mi_convert_u8_f32`): what the reference's hosts do on the CPU
    before calling the model (sample/visual_odometry.py:65-92)."""
    src = image.contiguous()
    out = torch.empty(src.shape, dtype=F32, device=src.device)
    if src.numel():
        N.call("mi_convert_u8_f32", N.dev(src, U8, "image"), src.numel(), out.data_ptr(), N.stream_ptr())
    return out


class ImagePair:
    """image1 / image2 of a matcher (two equally shaped (B,1,H,W) batches) as ONE batch of 2B images, batch a first, for
    the front-end entry points that take two base pointers (`mi_corner_response_pair`, `mi_sparse_bad_pair`,
    `mi_angle_at_keypoints_pair`, `mi_sparse_bad_oriented_pair`): nothing is concatenated, every stage is one launch
    for both images, and what follows (NMS, top-k, ...) runs on ordinary (2B, ...) tensors.  Accepted wherever those four
    ops take `image`."""

    def __init__(self, a: torch.Tensor, b: torch.Tensor):
        if a.shape != b.shape or a.dtype != b.dtype or a.device != b.device:
            raise RuntimeError(f"image batches differ: {tuple(a.shape)} {a.dtype} {a.device} vs {tuple(b.shape)} {b.dtype} {b.device}")
        self.a, self.b = a, b

    @property
    def device(self):
        return self.a.device

    @property
    def dtype(self):
        return self.a.dtype

    @property
    def shape(self):
        return torch.Size((2 * self.a.shape[0],) + tuple(self.a.shape[1:]))


def _images(image, what: str, keep_u8: bool = False):
    """(N,1,H,W) image batch as a contiguous float32 tensor.  uint8 frames (the u8 ingest path): kept as they are for
    the entry points that have a uint8 form (keep_u8), converted on the device for the others.  An ImagePair comes
    back as an ImagePair of two such tensors."""
    if isinstance(image, ImagePair):
        return ImagePair(_images(image.a, what, keep_u8), _images(image.b, what, keep_u8))
    if image.dim() != 4 or image.shape[1] != 1:
        raise RuntimeError(f"{what} must have shape (N, 1, H, W), got {tuple(image.shape)}")
    if image.dtype == U8:
        return image.contiguous() if keep_u8 else convert_u8(image)
    return image.float().contiguous()


TILE_COUNTER_BYTES = 16640          # include/mi355x_match.h MI_TILE_COUNTER_BYTES

# Sinkhorn solver flags handed to mi_sinkhorn_dots / mi_match_pairs (include/mi355x_match.h, "co-residency"):
# 0 = MI_SOLVER_DEFAULT; MI_SOLVER_MULTI_LAUNCH for a process that shares its GPU with long-running foreign kernels;
# MI_SOLVER_NO_FORK keeps the >= 64-pair Sinkhorn on the caller's stream (no helper streams, no events, no tuning).
# A preference of this Python layer (the C library has no process-wide switch); results are identical either way.
MI_SOLVER_DEFAULT, MI_SOLVER_MULTI_LAUNCH, MI_SOLVER_NO_FORK = 0, 1, 2
# (not a preference: sinkhorn_bits sets it by itself when the descriptors have fewer than 1024 bits, which is what lets
# mi_sinkhorn_dots' row kernel read the uint16 dot products as fp16 denormals)
MI_SOLVER_DOTS_BELOW_1024 = 4
_solver_flags = MI_SOLVER_DEFAULT


def set_solver_flags(flags: int) -> None:
    global _solver_flags
    if not isinstance(flags, int) or flags & ~(MI_SOLVER_MULTI_LAUNCH | MI_SOLVER_NO_FORK):
        raise ValueError("solver flags must be MI_SOLVER_DEFAULT (0) or an OR of MI_SOLVER_MULTI_LAUNCH (1) and "
                         f"MI_SOLVER_NO_FORK (2), got {flags}")
    _solver_flags = int(flags)


MI_SCHEDULE_UNDECIDED, MI_SCHEDULE_CALLER_HELPER, MI_SCHEDULE_TWO_HELPERS, MI_SCHEDULE_UNSPLIT = -1, 0, 1, 2


def sinkhorn_schedule(batch: int, n: int, m: int, iterations: int) -> int:
    """The stream schedule in force for mi_sinkhorn_dots calls of this shape on torch's current stream
    (`mi_sinkhorn_dots_schedule`): MI_SCHEDULE_UNDECIDED until the tuner has decided or a schedule is pinned."""
    return int(N.load().mi_sinkhorn_dots_schedule(N.stream_ptr(), int(batch), int(n), int(m), int(iterations)))


def set_sinkhorn_schedule(schedule: int) -> None:
    """Pin a stream schedule for torch's current stream, or MI_SCHEDULE_UNDECIDED to start tuning over
    (`mi_sinkhorn_dots_set_schedule`)."""
    N.check(N.load().mi_sinkhorn_dots_set_schedule(N.stream_ptr(), int(schedule)), "mi_sinkhorn_dots_set_schedule")


def corner_response(image: torch.Tensor, block_size: int) -> torch.Tensor:
    img = _images(image, "image", keep_u8=True)
    n, _, h, w = img.shape
    out = torch.empty(img.shape, dtype=F32, device=img.device)
    u8 = img.dtype == U8
    # K1's ticket counters: a fresh block per call from the caching allocator (stream-ordered: no sharing between
    # streams, nothing cached per stream); the library clears it on the stream before the kernel draws from it
    ctr = torch.empty(TILE_COUNTER_BYTES // 4, dtype=torch.int32, device=img.device)
    if isinstance(img, ImagePair):
        N.call("mi_corner_response_pair", N.dev(img.a, U8 if u8 else F32, "image"), N.dev(img.b, U8 if u8 else F32, "image"),
               int(u8), n // 2, h, w, int(block_size), N.dev(out, F32, "score"), ctr.data_ptr(), N.stream_ptr())
        return out
    N.call("mi_corner_response_balanced", N.dev(img, U8 if u8 else F32, "image"), int(u8), n, h, w, int(block_size),
           N.dev(out, F32, "score"), ctr.data_ptr(), N.stream_ptr())
    return out


def _score_maps(scores: torch.Tensor) -> torch.Tensor:
    if scores.dim() != 3:
        raise RuntimeError(f"scores must have shape (B, H, W), got {tuple(scores.shape)}")
    return scores.float().contiguous()


def nms_mask(scores: torch.Tensor, radius: int) -> torch.Tensor:
    s = _score_maps(scores)
    b, h, w = s.shape
    mask = torch.empty_like(s)
    N.call("mi_nms_mask", N.dev(s, F32, "scores"), b, h, w, int(radius), N.dev(mask, F32, "mask"), N.stream_ptr())
    return mask


def candidate_layout(h: int, w: int) -> tuple[int, int]:
    """(segments per image, slots per segment) of the K2 candidate buffer."""
    import ctypes
    seg, cap = ctypes.c_int(0), ctypes.c_int(0)
    N.check(N.load().mi_candidate_layout(int(h), int(w), ctypes.byref(seg), ctypes.byref(cap)), "mi_candidate_layout")
    return seg.value, cap.value


def _candidate_buffers(b: int, h: int, w: int, device):
    seg, cap = candidate_layout(h, w)
    cand = torch.empty((b, seg, cap), dtype=torch.int64, device=device)
    count = torch.empty((b, seg), dtype=torch.int32, device=device)      # fully written by K2
    return cand, count, seg, cap


def _topk_from_candidates(cand, count, seg, cap, b, h, w, k):
    if h * w < k:
        raise RuntimeError(f"selected index k out of range (k={k} > H*W={h * w})")  # torch.topk's failure mode
    kpts = torch.empty((b, k, 2), dtype=F32, device=cand.device)
    ksc = torch.empty((b, k), dtype=F32, device=cand.device)
    N.call("mi_topk_keypoints", cand.data_ptr(), count.data_ptr(), seg, cap, b, w, int(k), kpts.data_ptr(),
           ksc.data_ptr(), N.stream_ptr())
    return kpts, ksc


def nms_topk(scores: torch.Tensor, radius: int, k: int, score_threshold: float = 0.0, border_margin: int = 0):
    """Fused NMS + border + threshold + top-k (the mask is never materialised)."""
    s = _score_maps(scores)
    b, h, w = s.shape
    cand, count, seg, cap = _candidate_buffers(b, h, w, s.device)
    N.call("mi_nms_candidates", N.dev(s, F32, "scores"), b, h, w, int(radius), float(score_threshold),
           int(border_margin), cand.data_ptr(), count.data_ptr(), N.stream_ptr())
    return _topk_from_candidates(cand, count, seg, cap, b, h, w, k)


def select_topk(scores: torch.Tensor, mask: torch.Tensor, k: int, score_threshold: float = 0.0,
                border_margin: int = 0):
    s = _score_maps(scores)
    mk = _score_maps(mask)
    if mk.shape != s.shape:
        raise RuntimeError(f"nms_mask shape {tuple(mk.shape)} != scores shape {tuple(s.shape)}")
    b, h, w = s.shape
    cand, count, seg, cap = _candidate_buffers(b, h, w, s.device)
    N.call("mi_select_candidates", N.dev(s, F32, "scores"), N.dev(mk, F32, "nms_mask"), b, h, w,
           float(score_threshold), int(border_margin), cand.data_ptr(), count.data_ptr(), N.stream_ptr())
    return _topk_from_candidates(cand, count, seg, cap, b, h, w, k)


def bad_plan(pair_geom: torch.Tensor, pair_thr: torch.Tensor) -> torch.Tensor:
    """Device-resident fast-path plan for a pair table (built once; see mi_bad_plan_build)."""
    p = pair_geom.numel()
    nbytes = int(N.load().mi_bad_plan_bytes(p))
    plan = torch.empty(((nbytes + 15) // 16 * 2,), dtype=torch.int64, device=pair_geom.device)
    N.call("mi_bad_plan_build", N.dev(pair_geom, torch.int32, "pair_geom"), N.dev(pair_thr, F32, "pair_thr"), p,
           plan.data_ptr(), N.stream_ptr())
    return plan


def sparse_bad(image: torch.Tensor, keypoints: torch.Tensor, pair_geom: torch.Tensor, pair_thr: torch.Tensor,
               mode: int, temperature: float, normalize: bool, want_desc: bool = True, want_bits: bool = False,
               plan: torch.Tensor | None = None):
    img = _images(image, "image", keep_u8=True)
    n, _, h, w = img.shape
    if keypoints.dim() != 3 or keypoints.shape[0] != n or keypoints.shape[2] != 2:
        raise RuntimeError(f"keypoints must have shape ({n}, K, 2), got {tuple(keypoints.shape)}")
    kp = keypoints.float().contiguous()
    k = kp.shape[1]
    p = pair_geom.numel()
    desc = torch.empty((n, k, p), dtype=F32, device=img.device) if want_desc else None
    bits = torch.empty((n, k, p // 32), dtype=torch.int32, device=img.device) if want_bits else None
    u8 = img.dtype == U8
    if isinstance(img, ImagePair):
        N.call("mi_sparse_bad_pair", N.dev(img.a, U8 if u8 else F32, "image"), N.dev(img.b, U8 if u8 else F32, "image"),
               int(u8), n // 2, h, w, N.dev(kp, F32, "keypoints"), k,
               N.dev(pair_geom, torch.int32, "pair_geom"), N.dev(pair_thr, F32, "pair_thr"), p, int(mode),
               float(temperature), int(bool(normalize)), desc.data_ptr() if want_desc else None,
               bits.data_ptr() if want_bits else None, plan.data_ptr() if plan is not None else None,
               torch.empty((n * k,), dtype=torch.uint8, device=img.device).data_ptr() if plan is not None else None,
               N.stream_ptr())
        return desc, bits
    N.call("mi_sparse_bad_u8" if u8 else "mi_sparse_bad", N.dev(img, U8 if u8 else F32, "image"), n, h, w,
           N.dev(kp, F32, "keypoints"), k,
           N.dev(pair_geom, torch.int32, "pair_geom"), N.dev(pair_thr, F32, "pair_thr"), p, int(mode),
           float(temperature), int(bool(normalize)), desc.data_ptr() if want_desc else None,
           bits.data_ptr() if want_bits else None, plan.data_ptr() if plan is not None else None,
           torch.empty((n * k,), dtype=torch.uint8, device=img.device).data_ptr() if plan is not None else None,
           N.stream_ptr())
    return desc, bits


def bad_dense(image: torch.Tensor, pair_geom: torch.Tensor, pair_thr: torch.Tensor, mode: int,
              temperature: float) -> torch.Tensor:
    img = _images(image, "x")
    n, _, h, w = img.shape
    p = pair_geom.numel()
    out = torch.empty((n, p, h, w), dtype=F32, device=img.device)
    N.call("mi_bad_dense", N.dev(img, F32, "x"), n, h, w, N.dev(pair_geom, torch.int32, "pair_geom"),
           N.dev(pair_thr, F32, "pair_thr"), p, int(mode), float(temperature), out.data_ptr(), N.stream_ptr())
    return out


def bad_dense_oriented(image: torch.Tensor, orientation: torch.Tensor, pair_geom: torch.Tensor, pair_thr: torch.Tensor,
                       mode: int, temperature: float) -> torch.Tensor:
    img = _images(image, "x")
    n, _, h, w = img.shape
    ori = orientation.float().contiguous()
    if tuple(ori.shape) != (n, 1, h, w):
        raise RuntimeError(f"orientation must have shape ({n}, 1, {h}, {w}), got {tuple(ori.shape)}")
    p = pair_geom.numel()
    out = torch.empty((n, p, h, w), dtype=F32, device=img.device)
    N.call("mi_bad_dense_oriented", N.dev(img, F32, "x"), N.dev(ori, F32, "orientation"), n, h, w,
           N.dev(pair_geom, torch.int32, "pair_geom"), N.dev(pair_thr, F32, "pair_thr"), p, int(mode),
           float(temperature), out.data_ptr(), N.stream_ptr())
    return out


def gather_descriptors(descriptor_map: torch.Tensor, keypoints: torch.Tensor, bilinear: bool) -> torch.Tensor:
    if descriptor_map.dim() != 4 or keypoints.dim() != 3 or keypoints.shape[0] != descriptor_map.shape[0]:
        raise RuntimeError(f"expected (B,D,H,W) and (B,N,2), got {tuple(descriptor_map.shape)} {tuple(keypoints.shape)}")
    dm = descriptor_map.float().contiguous()
    kp = keypoints.float().contiguous()
    b, d, h, w = dm.shape
    out = torch.empty((b, kp.shape[1], d), dtype=F32, device=dm.device)
    N.call("mi_gather_descriptors", N.dev(dm, F32, "descriptor_map"), b, d, h, w, N.dev(kp, F32, "keypoints"),
           kp.shape[1], int(bool(bilinear)), out.data_ptr(), N.stream_ptr())
    return out


def angle_map(image: torch.Tensor, moment_kernels: torch.Tensor, patch_size: int) -> torch.Tensor:
    img = _images(image, "image")
    n, _, h, w = img.shape
    out = torch.empty_like(img)
    N.call("mi_angle_map", N.dev(img, F32, "image"), n, h, w, int(patch_size),
           N.dev(moment_kernels.contiguous(), F32, "moment_kernels"), out.data_ptr(), N.stream_ptr())
    return out


def angle_at_keypoints(image: torch.Tensor, keypoints: torch.Tensor, moment_kernels: torch.Tensor,
                       patch_size: int) -> torch.Tensor:
    img = _images(image, "image")
    n, _, h, w = img.shape
    kp = keypoints.float().contiguous()
    theta = torch.empty((n, kp.shape[1]), dtype=F32, device=img.device)
    if isinstance(img, ImagePair):
        N.call("mi_angle_at_keypoints_pair", N.dev(img.a, F32, "image"), N.dev(img.b, F32, "image"), n // 2, h, w,
               N.dev(kp, F32, "keypoints"), kp.shape[1], int(patch_size),
               N.dev(moment_kernels.contiguous(), F32, "moment_kernels"), theta.data_ptr(), N.stream_ptr())
        return theta
    N.call("mi_angle_at_keypoints", N.dev(img, F32, "image"), n, h, w, N.dev(kp, F32, "keypoints"), kp.shape[1],
           int(patch_size), N.dev(moment_kernels.contiguous(), F32, "moment_kernels"), theta.data_ptr(),
           N.stream_ptr())
    return theta


def sparse_bad_oriented(image: torch.Tensor, keypoints: torch.Tensor, orientation: torch.Tensor,
                        pair_geom: torch.Tensor, pair_thr: torch.Tensor, mode: int, temperature: float,
                        normalize: bool, want_desc: bool = True, want_bits: bool = False, bilinear: bool = False,
                        max_reach: float = 0.0):
    """orientation: dense map (B,1,H,W) -- the reference's argument -- or per-keypoint angles (B,K).
    bilinear: sampling_mode="bilinear" of the box-mean maps instead of "nearest".
    max_reach: upper bound of |pair offset| + box radius over the table in pixels (0 = unknown: the 60-pixel window)."""
    img = _images(image, "image")
    n, _, h, w = img.shape
    if keypoints.dim() != 3 or keypoints.shape[0] != n or keypoints.shape[2] != 2:
        raise RuntimeError(f"keypoints must have shape ({n}, K, 2), got {tuple(keypoints.shape)}")
    kp = keypoints.float().contiguous()
    k = kp.shape[1]
    ang = orientation.float().contiguous()
    if ang.dim() == 4 and tuple(ang.shape) == (n, 1, h, w):
        amap, akp = N.dev(ang, F32, "orientation"), None
    elif ang.dim() == 2 and tuple(ang.shape) == (n, k):
        amap, akp = None, N.dev(ang, F32, "orientation")
    else:
        raise RuntimeError(f"orientation must be ({n}, 1, {h}, {w}) or ({n}, {k}), got {tuple(ang.shape)}")
    p = pair_geom.numel()
    desc = torch.empty((n, k, p), dtype=F32, device=img.device) if want_desc else None
    bits = torch.empty((n, k, p // 32), dtype=torch.int32, device=img.device) if want_bits else None
    status = torch.empty((n, k), dtype=torch.uint8, device=img.device)      # int32-table fast path bookkeeping
    if isinstance(img, ImagePair):
        if akp is None:
            raise RuntimeError("an ImagePair takes per-keypoint angles (2B, K), not a dense orientation map")
        N.call("mi_sparse_bad_oriented_pair", N.dev(img.a, F32, "image"), N.dev(img.b, F32, "image"), n // 2, h, w,
               N.dev(kp, F32, "keypoints"), k, akp, N.dev(pair_geom, torch.int32, "pair_geom"),
               N.dev(pair_thr, F32, "pair_thr"), p, int(mode), float(temperature), int(bool(normalize)),
               int(bool(bilinear)), float(max_reach), desc.data_ptr() if want_desc else None,
               bits.data_ptr() if want_bits else None, status.data_ptr(), N.stream_ptr())
        return desc, bits
    N.call("mi_sparse_bad_oriented", N.dev(img, F32, "image"), n, h, w, N.dev(kp, F32, "keypoints"), k, amap, akp,
           N.dev(pair_geom, torch.int32, "pair_geom"), N.dev(pair_thr, F32, "pair_thr"), p, int(mode),
           float(temperature), int(bool(normalize)), int(bool(bilinear)), float(max_reach),
           desc.data_ptr() if want_desc else None, bits.data_ptr() if want_bits else None, status.data_ptr(),
           N.stream_ptr())
    return desc, bits


def _pitch(m: int) -> int:
    return (m + 3) // 4 * 4


def cost_logscores_bits(bits1: torch.Tensor, bits2: torch.Tensor, normalized: bool, epsilon: float):
    b, n, words = bits1.shape
    m = bits2.shape[1]
    pitch = _pitch(m)
    z = torch.empty((b, n, pitch), dtype=F32, device=bits1.device)
    N.call("mi_cost_logscores_bits", N.dev(bits1, torch.int32, "bits1"), N.dev(bits2, torch.int32, "bits2"), b, n, m,
           words * 32, int(bool(normalized)), float(epsilon), z.data_ptr(), pitch, N.stream_ptr())
    return z, pitch


def cost_logscores_f32(desc1: torch.Tensor, desc2: torch.Tensor, distance: int, epsilon: float):
    if desc1.dim() != 3 or desc2.dim() != 3 or desc1.shape[0] != desc2.shape[0] or desc1.shape[2] != desc2.shape[2]:
        raise RuntimeError(f"descriptor shapes do not match: {tuple(desc1.shape)} vs {tuple(desc2.shape)}")
    d1 = desc1.float().contiguous()
    d2 = desc2.float().contiguous()
    b, n, d = d1.shape
    m = d2.shape[1]
    pitch = _pitch(m)
    z = torch.empty((b, n, pitch), dtype=F32, device=d1.device)
    N.call("mi_cost_logscores_f32", N.dev(d1, F32, "desc1"), N.dev(d2, F32, "desc2"), b, n, m, d, int(distance),
           float(epsilon), z.data_ptr(), pitch, N.stream_ptr())
    return z, pitch


def sinkhorn(z: torch.Tensor, m: int, pitch: int, dustbin_logscore: float, iterations: int,
             return_duals: bool = False, use_workspace: bool = True, want_p: bool = True):
    """want_p=False: duals only (P is not written); returns (None, u, v)."""
    b, n, _ = z.shape
    u = torch.empty((b, n + 1), dtype=F32, device=z.device)
    v = torch.empty((b, m + 1), dtype=F32, device=z.device)
    p = torch.empty((b, n + 1, m + 1), dtype=F32, device=z.device) if want_p else None
    wbytes = int(N.load().mi_sinkhorn_workspace_bytes(b, n, m))
    work = torch.empty((max(wbytes, 8) // 8,), dtype=torch.int64, device=z.device) if use_workspace else None
    N.call("mi_sinkhorn", N.dev(z, F32, "z"), b, n, m, pitch, float(dustbin_logscore), int(iterations),
           u.data_ptr(), v.data_ptr(), p.data_ptr() if p is not None else None,
           work.data_ptr() if work is not None else None, wbytes if work is not None else 0, N.stream_ptr())
    return (p, u, v) if (return_duals or not want_p) else p


DOTS_MIN_EPSILON = 0.005      # MI_DOTS_MIN_EPSILON (include/mi355x_match.h): below it the clamped fp32-Z form runs


def dots_supported(b: int, n: int, m: int, epsilon: float = 1.0) -> bool:
    return epsilon >= DOTS_MIN_EPSILON and int(N.load().mi_sinkhorn_dots_workspace_bytes(b, n, m)) > 0


def sinkhorn_bits(bits1: torch.Tensor, bits2: torch.Tensor, normalized: bool, epsilon: float, unused_score: float,
                  iterations: int, return_duals: bool = False, want_p: bool = True, return_state: bool = False):
    """Cost + Sinkhorn for packed hard-bit descriptors (B,N,D/32),(B,M,D/32) int32 -> P (B,N+1,M+1),
    uint16 dot-product form (M <= 1024; half the bytes per iteration).  return_state: also the
    (dots, row_info, col_info, pitch, (workspace, status word address)) the duals refer to (for mnn_from_duals_dots;
    the workspace is kept alive because the call's status word lives in it)."""
    b, n, words = bits1.shape
    m = bits2.shape[1]
    dev = bits1.device
    wbytes = int(N.load().mi_sinkhorn_dots_workspace_bytes(b, n, m)) if epsilon >= DOTS_MIN_EPSILON else 0
    if wbytes == 0:
        if return_state or not want_p:
            raise RuntimeError(f"the dot-product Sinkhorn form supports M <= 1024 and epsilon >= {DOTS_MIN_EPSILON}, "
                               f"got M = {m}, epsilon = {epsilon}")
        z, pitch = cost_logscores_bits(bits1, bits2, normalized, epsilon)
        return sinkhorn(z, m, pitch, -unused_score / epsilon, iterations, return_duals=return_duals)
    pitch = (m + 7) // 8 * 8
    dots = torch.empty((b, n, pitch), dtype=torch.int16, device=dev)
    row_info = torch.empty((b, n, 2), dtype=F32, device=dev)
    col_info = torch.empty((b, m, 2), dtype=F32, device=dev)
    N.call("mi_cost_dots_bits", N.dev(bits1, torch.int32, "bits1"), N.dev(bits2, torch.int32, "bits2"), b, n, m,
           words * 32, int(bool(normalized)), dots.data_ptr(), pitch, row_info.data_ptr(), col_info.data_ptr(),
           N.stream_ptr())
    u = torch.empty((b, n + 1), dtype=F32, device=dev)
    v = torch.empty((b, m + 1), dtype=F32, device=dev)
    p = torch.empty((b, n + 1, m + 1), dtype=F32, device=dev) if want_p else None
    work = torch.empty(((wbytes + 7) // 8,), dtype=torch.int64, device=dev)
    N.call("mi_sinkhorn_dots", dots.data_ptr(), row_info.data_ptr(), col_info.data_ptr(), b, n, m, pitch,
           float(epsilon), float(unused_score), 1.0 if normalized else float(words * 32), int(iterations),
           u.data_ptr(), v.data_ptr(), p.data_ptr() if p is not None else None, work.data_ptr(), wbytes,
           _solver_flags | (MI_SOLVER_DOTS_BELOW_1024 if words * 32 < 1024 else 0), N.stream_ptr())
    if return_state:
        status = N.load().mi_sinkhorn_dots_status_word(work.data_ptr(), b, n, m)
        return p, u, v, (dots, row_info, col_info, pitch, (work, status))
    return (p, u, v) if (return_duals or not want_p) else p


def match_filters(p: torch.Tensor, ratio_threshold: float, dustbin_margin: float):
    """In-place outlier filters on P (B,N+1,M+1) -> (P, valid (B,N) bool)."""
    b, n1, m1 = p.shape
    valid = torch.empty((b, n1 - 1), dtype=torch.bool, device=p.device)       # the kernel writes 0/1 bytes
    N.call("mi_match_filters", N.dev(p, F32, "P"), b, n1 - 1, m1 - 1, float(ratio_threshold), float(dustbin_margin),
           valid.data_ptr(), N.stream_ptr())
    return p, valid


def match_filter_masks(p: torch.Tensor, has_dustbin: bool, ratio_threshold: float, dustbin_margin: float) -> torch.Tensor:
    """Masks of the two outlier tests, P untouched: P (B,N+1,M+1) with has_dustbin, else the core (B,N,M) -> (B,N) bool."""
    pp = p.float().contiguous()
    b, n, m = pp.shape[0], pp.shape[1] - int(has_dustbin), pp.shape[2] - int(has_dustbin)
    valid = torch.empty((b, n), dtype=torch.bool, device=pp.device)
    N.call("mi_match_filter_masks", N.dev(pp, F32, "P"), b, n, m, int(has_dustbin), float(ratio_threshold),
           float(dustbin_margin), valid.data_ptr(), N.stream_ptr())
    return valid


def mnn_extract(p: torch.Tensor, kpts1: torch.Tensor, kpts2: torch.Tensor, max_matches: int, threshold: float,
                return_indices: bool = False):
    if p.dim() != 3:
        raise RuntimeError(f"P must have shape (B, N+1, M+1), got {tuple(p.shape)}")
    pp = p.float().contiguous()
    k1 = kpts1.float().contiguous()
    k2 = kpts2.float().contiguous()
    b, n, m = pp.shape[0], k1.shape[1], k2.shape[1]
    if pp.shape[1] != n + 1 or pp.shape[2] != m + 1:
        raise RuntimeError(f"P shape {tuple(pp.shape)} does not match keypoints ({n}, {m})")
    dev = pp.device
    row_best = torch.empty((b, n), dtype=torch.int64, device=dev)
    col_best = torch.empty((b, m), dtype=torch.int64, device=dev)
    mk1 = torch.empty((b, max_matches, 2), dtype=F32, device=dev)
    mk2 = torch.empty((b, max_matches, 2), dtype=F32, device=dev)
    sc = torch.empty((b, max_matches), dtype=F32, device=dev)
    valid = torch.empty((b, max_matches), dtype=torch.bool, device=dev)          # the kernel writes 0/1 bytes
    ij = torch.empty((b, max_matches, 2), dtype=torch.int32, device=dev)
    N.call("mi_mnn_extract", N.dev(pp, F32, "P"), b, n, m, N.dev(k1, F32, "keypoints1"), N.dev(k2, F32, "keypoints2"),
           int(max_matches), float(threshold), row_best.data_ptr(), col_best.data_ptr(), mk1.data_ptr(),
           mk2.data_ptr(), sc.data_ptr(), valid.data_ptr(), ij.data_ptr(), N.stream_ptr())
    out = (mk1, mk2, sc, valid)
    return out + (ij,) if return_indices else out


def _mnn_outputs(b, max_matches, dev):
    return (torch.empty((b, max_matches, 2), dtype=F32, device=dev), torch.empty((b, max_matches, 2), dtype=F32, device=dev),
            torch.empty((b, max_matches), dtype=F32, device=dev), torch.empty((b, max_matches), dtype=torch.bool, device=dev),
            torch.empty((b, max_matches, 2), dtype=torch.int32, device=dev))


def mnn_duals_supported(b: int, n: int, m: int) -> bool:
    return int(N.load().mi_mnn_duals_workspace_bytes(b, n, m)) > 0


def mnn_from_duals(z: torch.Tensor, m: int, pitch: int, u: torch.Tensor, v: torch.Tensor, kpts1: torch.Tensor,
                   kpts2: torch.Tensor, max_matches: int, threshold: float, return_indices: bool = False):
    """mnn_extract(P) with P = exp(Z + u + v) evaluated on the fly (P never written)."""
    b, n, _ = z.shape
    k1, k2 = kpts1.float().contiguous(), kpts2.float().contiguous()
    wbytes = int(N.load().mi_mnn_duals_workspace_bytes(b, n, m))
    if wbytes == 0:
        raise RuntimeError(f"mnn_from_duals supports N <= 4096 and M <= 1024, got ({n}, {m})")
    work = torch.empty((wbytes // 8,), dtype=torch.int64, device=z.device)
    mk1, mk2, sc, valid, ij = _mnn_outputs(b, max_matches, z.device)
    N.call("mi_mnn_from_duals", N.dev(z, F32, "z"), b, n, m, pitch, N.dev(u, F32, "u"), N.dev(v, F32, "v"),
           N.dev(k1, F32, "keypoints1"), N.dev(k2, F32, "keypoints2"), int(max_matches), float(threshold),
           work.data_ptr(), wbytes, mk1.data_ptr(), mk2.data_ptr(), sc.data_ptr(), valid.data_ptr(), ij.data_ptr(),
           N.stream_ptr())
    out = (mk1, mk2, sc, valid)
    return out + (ij,) if return_indices else out


def mnn_from_duals_dots(state, m: int, epsilon: float, u: torch.Tensor, v: torch.Tensor, kpts1: torch.Tensor,
                        kpts2: torch.Tensor, max_matches: int, threshold: float, return_indices: bool = False):
    dots, row_info, col_info, pitch = state[:4]
    status = state[4][1] if len(state) > 4 else None          # the producing Sinkhorn call's status word (device address)
    b, n, _ = dots.shape
    k1, k2 = kpts1.float().contiguous(), kpts2.float().contiguous()
    wbytes = int(N.load().mi_mnn_duals_workspace_bytes(b, n, m))
    if wbytes == 0:
        raise RuntimeError(f"mnn_from_duals_dots supports N <= 4096 and M <= 1024, got ({n}, {m})")
    work = torch.empty((wbytes // 8,), dtype=torch.int64, device=dots.device)
    mk1, mk2, sc, valid, ij = _mnn_outputs(b, max_matches, dots.device)
    N.call("mi_mnn_from_duals_dots", dots.data_ptr(), row_info.data_ptr(), col_info.data_ptr(), b, n, m, pitch,
           float(epsilon), N.dev(u, F32, "u"), N.dev(v, F32, "v"), N.dev(k1, F32, "keypoints1"),
           N.dev(k2, F32, "keypoints2"), int(max_matches), float(threshold), work.data_ptr(), wbytes, status,
           mk1.data_ptr(), mk2.data_ptr(), sc.data_ptr(), valid.data_ptr(), ij.data_ptr(), N.stream_ptr())
    out = (mk1, mk2, sc, valid)
    return out + (ij,) if return_indices else out


# ---- AKAZE (detector/akaze.py) ------------------------------------------------------------------

def akaze_diffuse(image: torch.Tensor, iterations: int, kappa: float, dt: float = 0.25) -> torch.Tensor:
    """NonLinearDiffusion.forward: `iterations` explicit steps, ping-ponging two buffers."""
    img = _images(image, "image")
    n, _, h, w = img.shape
    if iterations <= 0:
        return img
    bufs = [torch.empty_like(img), torch.empty_like(img) if iterations > 1 else None]
    cur = img
    for i in range(int(iterations)):
        dst = bufs[i & 1]
        N.call("mi_akaze_diffuse", N.dev(cur, F32, "image"), n, h, w, float(kappa), float(dt), dst.data_ptr(),
               N.stream_ptr())
        cur = dst
    return cur


def akaze_scale(image: torch.Tensor, iterations: int, kappa: float, dt: float, threshold: float, nms_size: int,
                scores_out: torch.Tensor | None = None, image_out: torch.Tensor | None = None):
    """One AKAZE scale in one launch: (diffused image, Hessian score map) = (NonLinearDiffusion(image),
    HessianDetector(diffused)); `mi_akaze_scale`."""
    img = _images(image, "image")
    n, _, h, w = img.shape
    out = image_out if image_out is not None else torch.empty_like(img)
    scores = scores_out if scores_out is not None else torch.empty_like(img)
    fused = bool(N.load().mi_akaze_scale_fused(int(iterations), int(nms_size)))
    tmp = None if fused or iterations <= 1 else torch.empty_like(img)
    N.call("mi_akaze_scale", N.dev(img, F32, "image"), n, h, w, int(iterations), float(kappa), float(dt), float(threshold),
           int(nms_size), N.dev(out, F32, "image_out"), N.dev(scores, F32, "scores"), tmp.data_ptr() if tmp is not None else None,
           N.stream_ptr())
    return out, scores


def akaze_scale_sets(image1: torch.Tensor, image2: torch.Tensor, iterations: int, kappa: float, dt: float, threshold: float,
                     nms_size: int, scores_out: torch.Tensor, image_out: torch.Tensor):
    """The first AKAZE scale of two equally shaped batches in one launch (`mi_akaze_scale_sets`): image_out / scores_out
    (2N,1,H,W) receive batch 1 then batch 2."""
    a, b = _images(image1, "image1"), _images(image2, "image2")
    if a.shape != b.shape:
        raise RuntimeError(f"image shapes differ: {tuple(a.shape)} vs {tuple(b.shape)}")
    n, _, h, w = a.shape
    if tuple(image_out.shape) != (2 * n, 1, h, w) or tuple(scores_out.shape) != (2 * n, 1, h, w):
        raise RuntimeError("image_out / scores_out must be (2N,1,H,W)")
    fused = bool(N.load().mi_akaze_scale_fused(int(iterations), int(nms_size)))
    tmp = None if fused or iterations <= 1 else torch.empty_like(a)
    N.call("mi_akaze_scale_sets", N.dev(a, F32, "image1"), N.dev(b, F32, "image2"), n, h, w, int(iterations), float(kappa),
           float(dt), float(threshold), int(nms_size), N.dev(image_out, F32, "image_out"), N.dev(scores_out, F32, "scores"),
           tmp.data_ptr() if tmp is not None else None, N.stream_ptr())
    return image_out, scores_out


AKAZE_KAPPA_MIN, AKAZE_KAPPA_MAX = 1e-3, 1e6          # include/mi355x_match.h MI_AKAZE_KAPPA_MIN / _MAX


def akaze_kappa_fused(kappa: float) -> bool:
    """kappa inside the range the fused scale kernels accept (their exactly rounded division helpers are verified for
    it); outside, the modules run the per-step kernels (IEEE operators)."""
    return AKAZE_KAPPA_MIN <= float(kappa) <= AKAZE_KAPPA_MAX


def akaze_attain(scale_scores: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
    """(S,N,1,H,W) per-scale maps + their maximum -> attain (N,1,H,W) uint8, bit s = scale s reaches it (the general-
    parameter route of AKAZE.detect_select; elementwise torch ops on the device)."""
    bits = torch.zeros(scores.shape, dtype=torch.int32, device=scores.device)
    for s in range(scale_scores.shape[0]):
        bits |= (scale_scores[s] == scores).to(torch.int32) << s
    return bits.to(U8)


def akaze_scale_select(image: torch.Tensor, iterations: int, kappa: float, dt: float, threshold: float, nms_size: int,
                       prev_scores: torch.Tensor | None, image_out: torch.Tensor | None = None):
    """The last AKAZE scale with the selection across scales folded in (`mi_akaze_scale_select`): (diffused image,
    best = max over prev_scores (S-1,N,1,H,W) and this scale's score map, attain (N,1,H,W) uint8: bit s = scale s
    reaches best)."""
    img = _images(image, "image")
    n, _, h, w = img.shape
    num_prev = 0 if prev_scores is None else int(prev_scores.shape[0])
    if num_prev > 7:
        raise RuntimeError(f"at most 8 scales, got {num_prev + 1}")
    if num_prev and tuple(prev_scores.shape[1:]) != (n, 1, h, w):
        raise RuntimeError(f"prev_scores must be (S-1,{n},1,{h},{w}), got {tuple(prev_scores.shape)}")
    out = image_out if image_out is not None else torch.empty_like(img)
    best = torch.empty_like(img)
    attain = torch.empty((n, 1, h, w), dtype=U8, device=img.device)
    fused = bool(N.load().mi_akaze_scale_fused(int(iterations), int(nms_size)))
    tmp = None if fused or iterations <= 1 else torch.empty_like(img)
    N.call("mi_akaze_scale_select", N.dev(img, F32, "image"), n, h, w, int(iterations), float(kappa), float(dt),
           float(threshold), int(nms_size), N.dev(out, F32, "image_out"),
           N.dev(prev_scores, F32, "prev_scores") if num_prev else None, num_prev, best.data_ptr(), attain.data_ptr(),
           tmp.data_ptr() if tmp is not None else None, N.stream_ptr())
    return out, best, attain


def akaze_orientation_from_attain(attain: torch.Tensor, scale_theta: torch.Tensor, keypoints: torch.Tensor) -> torch.Tensor:
    n, _, h, w = attain.shape
    s = int(scale_theta.shape[0])
    kp = keypoints.float().contiguous()
    k = kp.shape[1]
    theta = torch.empty((n, k), dtype=F32, device=attain.device)
    N.call("mi_akaze_orientation_from_attain", N.dev(attain, U8, "attain"), N.dev(scale_theta, F32, "scale_theta"), s, n,
           h, w, N.dev(kp, F32, "keypoints"), k, theta.data_ptr(), N.stream_ptr())
    return theta


def akaze_orientation_select(scale_images: torch.Tensor, attain: torch.Tensor, keypoints: torch.Tensor,
                             moment_kernels: torch.Tensor, patch_size: int) -> torch.Tensor:
    """AKAZE.forward's orientation at keypoints in one launch (`mi_akaze_orientation_select`): scale_images (S,N,1,H,W)
    stacked diffused images, attain (N,1,H,W) uint8 from akaze_scale_select."""
    s, n, _, h, w = scale_images.shape
    kp = keypoints.float().contiguous()
    k = kp.shape[1]
    theta = torch.empty((n, k), dtype=F32, device=attain.device)
    N.call("mi_akaze_orientation_select", N.dev(scale_images, F32, "scale_images"), n * h * w, s,
           N.dev(attain, U8, "attain"), n, h, w, N.dev(kp, F32, "keypoints"), k, int(patch_size),
           N.dev(moment_kernels.float().contiguous(), F32, "moment_kernels"), theta.data_ptr(), N.stream_ptr())
    return theta


def akaze_hessian_scores(image: torch.Tensor, threshold: float, nms_size: int,
                         out: torch.Tensor | None = None) -> torch.Tensor:
    img = _images(image, "image")
    n, _, h, w = img.shape
    if out is None:
        out = torch.empty_like(img)
    N.call("mi_akaze_hessian_scores", N.dev(img, F32, "image"), n, h, w, float(threshold), int(nms_size),
           N.dev(out, F32, "scores"), N.stream_ptr())
    return out


def akaze_combine(scale_scores: torch.Tensor, scale_orientations: torch.Tensor | None):
    """(S,N,1,H,W) stacks -> (scores, orientations) of AKAZE.forward; orientations None -> scores only."""
    s, n, _, h, w = scale_scores.shape
    scores = torch.empty((n, 1, h, w), dtype=F32, device=scale_scores.device)
    oris = torch.empty_like(scores) if scale_orientations is not None else None
    N.call("mi_akaze_combine", N.dev(scale_scores, F32, "scale_scores"),
           N.dev(scale_orientations, F32, "scale_orientations") if scale_orientations is not None else None,
           s, n, h, w, scores.data_ptr(), oris.data_ptr() if oris is not None else None, N.stream_ptr())
    return scores, oris


def akaze_orientation_at_keypoints(scale_scores: torch.Tensor, scale_theta: torch.Tensor,
                                   keypoints: torch.Tensor) -> torch.Tensor:
    s, n, _, h, w = scale_scores.shape
    kp = keypoints.float().contiguous()
    k = kp.shape[1]
    theta = torch.empty((n, k), dtype=F32, device=scale_scores.device)
    N.call("mi_akaze_orientation_at_keypoints", N.dev(scale_scores, F32, "scale_scores"),
           N.dev(scale_theta, F32, "scale_theta"), s, n, h, w, N.dev(kp, F32, "keypoints"), k, theta.data_ptr(),
           N.stream_ptr())
    return theta


# ---- essential-matrix head (geometry/essential_matrix_estimator.py) ---------------------------------

def _validity_bytes(valid: torch.Tensor | None) -> torch.Tensor | None:
    """A validity mask as the uint8 array the C ABI reads; a bool tensor is VIEWED (one byte per element, 0 / 1): no copy
    kernel on the per-pair path."""
    if valid is None:
        return None
    if valid.dtype == torch.bool:
        return valid.contiguous().view(torch.uint8)
    return (valid != 0).contiguous().view(torch.uint8)


def essential_matrix(p: torch.Tensor, pts1_n: torch.Tensor, pts2_n: torch.Tensor, valid1: torch.Tensor | None,
                     valid2: torch.Tensor | None, top_k: int, n_iter: int, n_iter_manifold: int,
                     banded: bool = True) -> torch.Tensor:
    """P (B,N+1,M+1), normalised points (B,N,2)/(B,M,2) as (x,y), optional validity (B,N)/(B,M) -> E (B,3,3).
    banded: the two-launch form on a workspace (top_k <= 4); False: the single-launch dense form."""
    if p.dim() != 3:
        raise RuntimeError(f"P must have shape (B, N+1, M+1), got {tuple(p.shape)}")
    pp = p.float().contiguous()
    b, n, m = pp.shape[0], pp.shape[1] - 1, pp.shape[2] - 1
    q1, q2 = pts1_n.float().contiguous(), pts2_n.float().contiguous()
    if tuple(q1.shape) != (b, n, 2) or tuple(q2.shape) != (b, m, 2):
        raise RuntimeError(f"points must be ({b},{n},2) and ({b},{m},2), got {tuple(q1.shape)}, {tuple(q2.shape)}")
    if (valid1 is None) != (valid2 is None):
        raise RuntimeError("valid1 and valid2 must be given together")
    v1, v2 = _validity_bytes(valid1), _validity_bytes(valid2)
    e = torch.empty((b, 3, 3), dtype=F32, device=pp.device)
    wbytes = int(N.load().mi_essential_matrix_workspace_bytes(b, n, m, int(top_k))) if banded else 0
    work = torch.empty(((wbytes + 7) // 8,), dtype=torch.int64, device=pp.device) if wbytes else None
    N.call("mi_essential_matrix", N.dev(pp, F32, "P"), b, n, m, N.dev(q1, F32, "pts1"), N.dev(q2, F32, "pts2"),
           N.dev(v1, torch.uint8, "valid1") if v1 is not None else None,
           N.dev(v2, torch.uint8, "valid2") if v2 is not None else None, int(top_k), int(n_iter), int(n_iter_manifold),
           e.data_ptr(), work.data_ptr() if work is not None else None, wbytes, N.stream_ptr())
    return e


def essential_matrix_dots(state, m: int, epsilon: float, u: torch.Tensor, v: torch.Tensor, pts1_n: torch.Tensor,
                          pts2_n: torch.Tensor, valid1: torch.Tensor | None, valid2: torch.Tensor | None, top_k: int,
                          n_iter: int, n_iter_manifold: int, banded: bool = True) -> torch.Tensor:
    """essential_matrix() without a materialised P (`mi_essential_matrix_dots`): `state` = the (dots, row_info, col_info,
    pitch, ...) tuple of sinkhorn_bits(return_state=True), u / v its duals.  Same E bit for bit as essential_matrix on
    the P that solve would have written."""
    dots, row_info, col_info, pitch = state[:4]
    b, n = dots.shape[0], dots.shape[1]
    q1, q2 = pts1_n.float().contiguous(), pts2_n.float().contiguous()
    if tuple(q1.shape) != (b, n, 2) or tuple(q2.shape) != (b, m, 2):
        raise RuntimeError(f"points must be ({b},{n},2) and ({b},{m},2), got {tuple(q1.shape)}, {tuple(q2.shape)}")
    if (valid1 is None) != (valid2 is None):
        raise RuntimeError("valid1 and valid2 must be given together")
    v1, v2 = _validity_bytes(valid1), _validity_bytes(valid2)
    e = torch.empty((b, 3, 3), dtype=F32, device=dots.device)
    wbytes = int(N.load().mi_essential_matrix_workspace_bytes(b, n, m, int(top_k))) if banded else 0
    work = torch.empty(((wbytes + 7) // 8,), dtype=torch.int64, device=dots.device) if wbytes else None
    N.call("mi_essential_matrix_dots", dots.data_ptr(), row_info.data_ptr(), col_info.data_ptr(), int(pitch), float(epsilon),
           N.dev(u, F32, "u"), N.dev(v, F32, "v"), b, n, m, N.dev(q1, F32, "pts1"), N.dev(q2, F32, "pts2"),
           N.dev(v1, torch.uint8, "valid1") if v1 is not None else None,
           N.dev(v2, torch.uint8, "valid2") if v2 is not None else None, int(top_k), int(n_iter), int(n_iter_manifold),
           e.data_ptr(), work.data_ptr() if work is not None else None, wbytes, N.stream_ptr())
    return e


# ---- FAST / DoG detectors (detector/fast.py, detector/dog.py) -----------------------------------------

def fast_score(image: torch.Tensor, threshold: float) -> torch.Tensor:
    img = _images(image, "image")
    n, _, h, w = img.shape
    out = torch.empty_like(img)
    N.call("mi_fast_score", N.dev(img, F32, "image"), n, h, w, float(threshold), out.data_ptr(), N.stream_ptr())
    return out


def dog_responses(image: torch.Tensor, weights_1d: torch.Tensor, want_maps: bool = True, want_score: bool = False):
    """weights_1d (S, ks): 1-D factors of the normalised Gaussians -> DoG maps (N, S-1, H, W) and/or the
    score map max_s |DoG_s| (N, 1, H, W)."""
    img = _images(image, "image")
    n, _, h, w = img.shape
    s, ks = weights_1d.shape
    out = torch.empty((n, s - 1, h, w), dtype=F32, device=img.device) if want_maps else None
    score = torch.empty((n, 1, h, w), dtype=F32, device=img.device) if want_score else None
    N.call("mi_dog_responses", N.dev(img, F32, "image"), n, h, w, N.dev(weights_1d.contiguous(), F32, "weights_1d"), s,
           ks, out.data_ptr() if out is not None else None, score.data_ptr() if score is not None else None,
           N.stream_ptr())
    return out, score


def normalise_keypoints(keypoints: torch.Tensor, k_inv: torch.Tensor) -> torch.Tensor:
    """(..., 2) pixel keypoints (y, x) -> (..., 2) normalised (x, y) = K^-1 [x, y, 1] (first two rows)."""
    kp = keypoints.float().contiguous()
    ki = k_inv.float().contiguous()
    out = torch.empty_like(kp)
    N.call("mi_normalise_keypoints", N.dev(kp, F32, "keypoints"), kp.numel() // 2, N.dev(ki, F32, "K_inv"),
           out.data_ptr(), N.stream_ptr())
    return out


def core_maxima(p: torch.Tensor):
    """P (B,N+1,M+1) -> (row maxima (B,N), column maxima (B,M)) of the core P[:, :N, :M]."""
    pp = p.float().contiguous()
    b, n, m = pp.shape[0], pp.shape[1] - 1, pp.shape[2] - 1
    r = torch.empty((b, n), dtype=F32, device=pp.device)
    c = torch.empty((b, m), dtype=F32, device=pp.device)
    N.call("mi_core_maxima", N.dev(pp, F32, "P"), b, n, m, r.data_ptr(), c.data_ptr(), N.stream_ptr())
    return r, c


def match_pairs(image1: torch.Tensor, image2: torch.Tensor, *, block_size: int, nms_radius: int, max_keypoints: int,
                score_threshold: float, border_margin: int, pair_geom: torch.Tensor, pair_thr: torch.Tensor,
                plan: torch.Tensor | None, normalize_descriptors: bool, epsilon: float, unused_score: float,
                sinkhorn_iterations: int, max_matches: int, match_threshold: float, want_ij: bool = False):
    """The whole path in ONE C-ABI call (`mi_match_pairs`): what MatchExtractionWrapper(ShiTomasiSparseBADSinkhornMatcher)
    computes for hard-binarised descriptors and the L2 cost, bit-identical to the module path.
    -> (keypoints1, keypoints2, matched1, matched2, scores, valid[, match_ij])."""
    u8 = image1.dtype == U8 and image2.dtype == U8           # both uint8: the u8 ingest form; otherwise float32
    a, b2 = _images(image1, "image1", keep_u8=u8), _images(image2, "image2", keep_u8=u8)
    if a.shape != b2.shape:
        raise RuntimeError(f"image shapes differ: {tuple(a.shape)} vs {tuple(b2.shape)}")
    n, _, h, w = a.shape
    dev = a.device
    prm = N.MatchParams(int(block_size), int(nms_radius), int(max_keypoints), float(score_threshold), int(border_margin),
                        int(pair_geom.numel()), N.dev(pair_geom, torch.int32, "pair_geom"), N.dev(pair_thr, F32, "pair_thr"),
                        plan.data_ptr() if plan is not None else None, int(bool(normalize_descriptors)), float(epsilon),
                        float(unused_score), int(sinkhorn_iterations), int(max_matches), float(match_threshold),
                        _solver_flags)
    import ctypes
    wbytes = int(N.load().mi_match_pairs_workspace_bytes(n, h, w, ctypes.byref(prm)))
    if wbytes == 0:
        raise RuntimeError("mi_match_pairs does not cover these parameters (K <= 1024, P % 64 == 0, odd block size ...)")
    work = torch.empty(((wbytes + 7) // 8,), dtype=torch.int64, device=dev)
    k, mx = int(max_keypoints), int(max_matches)
    kp1 = torch.empty((n, k, 2), dtype=F32, device=dev)
    kp2 = torch.empty((n, k, 2), dtype=F32, device=dev)
    mk1 = torch.empty((n, mx, 2), dtype=F32, device=dev)
    mk2 = torch.empty((n, mx, 2), dtype=F32, device=dev)
    sc = torch.empty((n, mx), dtype=F32, device=dev)
    valid = torch.empty((n, mx), dtype=torch.bool, device=dev)
    ij = torch.empty((n, mx, 2), dtype=torch.int32, device=dev) if want_ij else None
    N.call("mi_match_pairs_u8" if u8 else "mi_match_pairs", N.dev(a, U8 if u8 else F32, "image1"),
           N.dev(b2, U8 if u8 else F32, "image2"), n, h, w, ctypes.byref(prm), kp1.data_ptr(),
           kp2.data_ptr(), mk1.data_ptr(), mk2.data_ptr(), sc.data_ptr(), valid.data_ptr(),
           ij.data_ptr() if ij is not None else None, work.data_ptr(), wbytes, N.stream_ptr())
    out = (kp1, kp2, mk1, mk2, sc, valid)
    return out + (ij,) if want_ij else out
