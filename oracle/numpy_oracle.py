"""CPU oracle for the image-matching hot path -- TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference's algorithm (the reference is
pure Python on torch; its arithmetic lives in ATen CPU kernels, torch==2.6.0 per
reference requirements.txt:1-3).  It exists to CHECK the HIP path.  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import it; the product package never does.

Parity pin: every function here is checked in tests/test_oracle_golden.py
against vectors produced by running the reference itself in the build
container (tests/golden/make_golden.py, torch 2.10.0+rocm7.0 CPU).  The
reference ships no golden vectors of its own for this path.

Deliberate differences from the reference, all documented in DESIGN.md:
  * top-k tie order: torch.topk's is implementation-defined; the oracle (and
    the HIP path) use (score descending, linear index ascending).
  * BAD box means are evaluated in exact arithmetic (float64 summed-area
    table) instead of the reference's fp32 conv with weights fp32(1/area);
    the two agree to <= ~2.1e-4 per response, so a hard bit can differ only
    where |response - threshold| is below that ("fragile" bits, reported).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------
# detector: reference pytorch_model/detector/shi_tomasi.py:66-112
# --------------------------------------------------------------------------
def shi_tomasi_score(image: np.ndarray, block_size: int = 3, return_sqrt_term: bool = False) -> np.ndarray:
    """lambda_min of the Sobel structure tensor; (N,1,H,W) f32 -> (N,1,H,W) f32.

    shi_tomasi.py:78-83  replicate pad 1 + unnormalised 3x3 Sobel (cross-correlation)
    shi_tomasi.py:88-96  products, replicate pad bs//2 OF THE PRODUCTS, box sum
    shi_tomasi.py:102-110 (a+c)/2 - sqrt(((a-c)/2)^2 + b^2 + 1e-10), clamp >= 0
    Every step is an individually rounded fp32 op, as in the reference; sqrt is the
    IEEE correctly-rounded one (what ONNX Runtime, torch-CUDA and torch-CPU's own
    vectorised path use).  torch-CPU with MKL routes large contiguous tensors through
    VML vsSqrt, which is 1 ulp off on ~0.7 % of inputs -- the golden check allows that.
    """
    if block_size <= 0 or block_size % 2 == 0:
        raise ValueError(f"block_size must be a positive odd integer, got {block_size}")
    img = np.asarray(image).astype(F32)
    n, c, h, w = img.shape
    assert c == 1
    p = np.pad(img[:, 0], ((0, 0), (1, 1), (1, 1)), mode="edge")
    top, mid, bot = p[:, :-2], p[:, 1:-1], p[:, 2:]
    # column difference per row, then 1-2-1 vertically (same taps as the 3x3 kernel)
    ix = (top[:, :, 2:] - top[:, :, :-2]) + F32(2) * (mid[:, :, 2:] - mid[:, :, :-2]) + (bot[:, :, 2:] - bot[:, :, :-2])
    iy = (bot[:, :, :-2] - top[:, :, :-2]) + F32(2) * (bot[:, :, 1:-1] - top[:, :, 1:-1]) + (bot[:, :, 2:] - top[:, :, 2:])
    r = block_size // 2

    def box(q: np.ndarray) -> np.ndarray:
        e = np.pad(q, ((0, 0), (r, r), (r, r)), mode="edge")
        acc = np.zeros_like(q)
        for dy in range(block_size):
            for dx in range(block_size):
                acc = acc + e[:, dy:dy + h, dx:dx + w]
        return acc

    a, cc, b = box(ix * ix), box(iy * iy), box(ix * iy)
    half_trace = (a + cc) / F32(2)
    half_diff = (a - cc) / F32(2)
    disc = half_diff * half_diff + b * b
    root = np.sqrt(disc + F32(1e-10))
    lam = half_trace - root
    out = np.maximum(lam, F32(0))[:, None].astype(F32)
    if return_sqrt_term:
        return out, root[:, None]
    return out


# --------------------------------------------------------------------------
# NMS + top-k: reference pytorch_model/utils/keypoint_utils.py:12-117
# --------------------------------------------------------------------------
def nms_mask(scores: np.ndarray, nms_radius: int) -> np.ndarray:
    """keypoint_utils.py:26-43: square (2r+1)^2 window max with -inf outside the
    image; keep where s >= max - 1e-7 (fp32).  (B,H,W) f32 -> (B,H,W) f32 {0,1}."""
    s = np.asarray(scores, F32)
    b, h, w = s.shape
    r = int(nms_radius)
    e = np.pad(s, ((0, 0), (r, r), (r, r)), mode="constant", constant_values=-np.inf)
    rowmax = e[:, :, 0:w]
    for dx in range(1, 2 * r + 1):
        rowmax = np.maximum(rowmax, e[:, :, dx:dx + w])
    win = rowmax[:, 0:h]
    for dy in range(1, 2 * r + 1):
        win = np.maximum(win, rowmax[:, dy:dy + h])
    return (s >= (win - F32(1e-7))).astype(F32)


def masked_scores(scores, mask, score_threshold=0.0, border_margin=0) -> np.ndarray:
    """keypoint_utils.py:77-92: s*mask*border, zero where <= threshold."""
    s = np.asarray(scores, F32)
    b, h, w = s.shape
    m = s * np.asarray(mask, F32)
    if border_margin > 0:
        g = int(border_margin)
        yv = ((np.arange(h) >= g) & (np.arange(h) < h - g)).astype(F32)
        xv = ((np.arange(w) >= g) & (np.arange(w) < w - g)).astype(F32)
        m = m * (yv[None, :, None] * xv[None, None, :])
    return np.where(m > F32(score_threshold), m, F32(0)).astype(F32)


def select_topk_keypoints(scores, mask, max_keypoints, score_threshold=0.0, border_margin=0):
    """keypoint_utils.py:94-115 with the tie policy (score desc, index asc).

    Returns keypoints (B,K,2) f32 (y,x; invalid = -1,-1), scores (B,K) f32 and the
    flat indices (B,K) int64 (-1 for invalid) for test bookkeeping."""
    m = masked_scores(scores, mask, score_threshold, border_margin)
    b, h, w = m.shape
    k = int(max_keypoints)
    flat = m.reshape(b, -1)
    if flat.shape[1] < k:
        raise RuntimeError("selected index k out of range")  # torch.topk's own failure mode
    kp = np.empty((b, k, 2), F32)
    sc = np.empty((b, k), F32)
    ids = np.empty((b, k), np.int64)
    for i in range(b):
        order = np.argsort(-flat[i].astype(np.float64), kind="stable")[:k]
        v = flat[i][order]
        ok = v > 0
        kp[i, :, 0] = np.where(ok, order // w, -1)
        kp[i, :, 1] = np.where(ok, order % w, -1)
        sc[i] = np.where(ok, v, 0)
        ids[i] = np.where(ok, order, -1)
    return kp, sc, ids


# --------------------------------------------------------------------------
# Sparse BAD: reference pytorch_model/descriptor/bad.py:436-576
# --------------------------------------------------------------------------
def _nearest_centre(pos: np.ndarray, size: int) -> np.ndarray:
    """bad.py:469-470,518-535 + ATen grid_sampler(nearest, border, align_corners):
    g = pos*fp32(2/(size-1+1e-8)) - 1 ; x = ((g+1)/2)*(size-1) ; clip ; nearbyint."""
    scale = F32(2.0 / (size - 1 + 1e-8))
    g = pos.astype(F32) * scale - F32(1)
    x = ((g + F32(1)) / F32(2)) * F32(size - 1)
    x = np.minimum(np.maximum(x, F32(0)), F32(size - 1))
    return np.rint(x).astype(np.int64)  # rint = round half to even = std::nearbyint


def sparse_bad(
    image: np.ndarray,
    keypoints: np.ndarray,
    box_params: np.ndarray,
    thresholds: np.ndarray,
    binarize: bool = False,
    soft_binarize: bool = True,
    temperature: float = 10.0,
    normalize_descriptors: bool = True,
    return_aux: bool = False,
):
    """Non-oriented, nearest-mode SparseBAD.forward (bad.py:458-574).

    image (B,1,H,W) f32, keypoints (B,K,2) f32 (y,x) -> desc (B,K,P) f32.
    box_params (P,5) = (x1,x2,y1,y2,r) in the 32x32 patch frame (offsets = value-16,
    bad.py:403-406); box content is taken from the replicate-padded image
    (bad.py:474-478), the box CENTRE is clamped to the image (padding_mode="border").
    aux = dict(centered f64 (B,K,P), bits bool (B,K,P), valid bool (B,K)).
    """
    img = np.asarray(image, F32)
    bsz, _, h, w = img.shape
    kp = np.asarray(keypoints, F32)
    box = np.asarray(box_params).astype(np.int64)
    thr = np.asarray(thresholds, F32).astype(np.float64)
    rmax = int(box[:, 4].max())
    valid = kp[:, :, 0] >= 0                                           # bad.py:461
    ky = np.clip(kp[:, :, 0], F32(0), F32(h - 1))                      # bad.py:464-465
    kx = np.clip(kp[:, :, 1], F32(0), F32(w - 1))
    ox1, ox2, oy1, oy2 = [(box[:, i] - 16).astype(F32) for i in range(4)]
    rad = box[:, 4]
    area = ((2 * rad + 1) ** 2).astype(np.float64)

    centered = np.empty((bsz, kp.shape[1], box.shape[0]), np.float64)
    for b in range(bsz):
        e = np.pad(img[b, 0].astype(np.float64), rmax, mode="edge")
        sat = np.zeros((e.shape[0] + 1, e.shape[1] + 1), np.float64)
        sat[1:, 1:] = e.cumsum(0).cumsum(1)

        def mean_box(cy, cx):
            y0 = cy - rad[None, :] + rmax
            y1 = cy + rad[None, :] + rmax + 1
            x0 = cx - rad[None, :] + rmax
            x1 = cx + rad[None, :] + rmax + 1
            s = sat[y1, x1] - sat[y0, x1] - sat[y1, x0] + sat[y0, x0]
            return s / area[None, :]

        c1y = _nearest_centre(ky[b][:, None] + oy1[None, :], h)
        c1x = _nearest_centre(kx[b][:, None] + ox1[None, :], w)
        c2y = _nearest_centre(ky[b][:, None] + oy2[None, :], h)
        c2x = _nearest_centre(kx[b][:, None] + ox2[None, :], w)
        centered[b] = mean_box(c1y, c1x) - mean_box(c2y, c2x) - thr[None, :]

    bits = centered <= 0
    if not binarize:
        desc = centered.astype(F32)
    elif soft_binarize:
        z = (-(centered.astype(F32)) * F32(temperature)).astype(F32)
        with np.errstate(over="ignore"):
            desc = (F32(1) / (F32(1) + np.exp(-z))).astype(F32)
    else:
        desc = bits.astype(F32)
    desc = desc * valid[:, :, None].astype(F32)
    if normalize_descriptors:                                          # F.normalize, eps 1e-12
        nrm = np.sqrt((desc * desc).sum(-1, dtype=F32, keepdims=True)).astype(F32)
        desc = (desc / np.maximum(nrm, F32(1e-12))).astype(F32)
    if return_aux:
        return desc, {"centered": centered, "bits": bits & valid[:, :, None], "valid": valid}
    return desc


def pack_bits(bits: np.ndarray) -> np.ndarray:
    """bool (...,P) -> uint32 (...,P/32); bit p lives in word p//32, position p%32."""
    b = np.asarray(bits, bool)
    p = b.shape[-1]
    assert p % 32 == 0
    w = b.reshape(*b.shape[:-1], p // 32, 32).astype(np.uint64)
    return (w << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)


# --------------------------------------------------------------------------
# cost + Sinkhorn: reference pytorch_model/matching/sinkhorn.py:79-208
# --------------------------------------------------------------------------
def cost_matrix(desc1: np.ndarray, desc2: np.ndarray, distance_type: str = "l2", dtype=F32) -> np.ndarray:
    """sinkhorn.py:95-108."""
    a = np.asarray(desc1, dtype)
    b = np.asarray(desc2, dtype)
    if distance_type == "l2":
        n1 = (a * a).sum(-1, keepdims=True)
        n2 = (b * b).sum(-1, keepdims=True)
        c = n1 + np.swapaxes(n2, -1, -2) - dtype(2) * (a @ np.swapaxes(b, -1, -2))
        return np.maximum(c, dtype(0))
    if distance_type == "l1":
        return np.abs(a[:, :, None, :] - b[:, None, :, :]).sum(-1)
    raise ValueError(f"distance_type must be 'l1' or 'l2', got {distance_type}")


def _lse(x: np.ndarray, axis: int) -> np.ndarray:
    m = x.max(axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0)
    return (np.log(np.exp(x - m).sum(axis=axis, keepdims=True)) + m).squeeze(axis)


def sinkhorn_from_cost(cost, iterations=20, epsilon=1.0, unused_score=1.0, dtype=F32, return_duals=False):
    """sinkhorn.py:170-206: Z = pad(-cost/eps, dustbin=-unused/eps); log mu / log nu;
    `iterations` x { u = log mu - LSE_j(Z+v) ; v = log nu - LSE_i(Z+u) } ; P = exp(Z+u+v)."""
    c = np.asarray(cost, dtype)
    bsz, n, m = c.shape
    z = np.full((bsz, n + 1, m + 1), dtype(-unused_score / epsilon), dtype)
    z[:, :n, :m] = -c / dtype(epsilon)
    log_mu = np.zeros((bsz, n + 1), dtype)
    log_nu = np.zeros((bsz, m + 1), dtype)
    log_mu[:, n] = np.log(dtype(m))
    log_nu[:, m] = np.log(dtype(n))
    u = np.zeros_like(log_mu)
    v = np.zeros_like(log_nu)
    for _ in range(int(iterations)):
        u = (log_mu - _lse(z + v[:, None, :], 2)).astype(dtype)
        v = (log_nu - _lse(z + u[:, :, None], 1)).astype(dtype)
    p = np.exp(z + u[:, :, None] + v[:, None, :]).astype(dtype)
    if return_duals:
        return p, z, u, v
    return p


def sinkhorn_match(desc1, desc2, iterations=20, epsilon=1.0, unused_score=1.0, distance_type="l2", dtype=F32):
    """SinkhornMatcher.forward (sinkhorn.py:149-208)."""
    if iterations <= 0:
        raise ValueError(f"iterations must be positive, got {iterations}")
    if epsilon <= 0:
        raise ValueError(f"epsilon must be positive, got {epsilon}")
    return sinkhorn_from_cost(cost_matrix(desc1, desc2, distance_type, dtype), iterations, epsilon, unused_score, dtype)


# --------------------------------------------------------------------------
# MNN extraction: reference pytorch_model/matching/match_extraction.py:72-181
# --------------------------------------------------------------------------
def mnn_extract(p, kpts1, kpts2, max_matches=100, threshold=0.1):
    """Mutual argmax on P[:N,:M] (first index on ties), score >= threshold, top
    `max_matches` by score (ties: lower row index first), zero padded.
    Returns mk1 (B,Mx,2), mk2 (B,Mx,2), scores (B,Mx), valid (B,Mx) bool, and the
    (i, j) index pairs (B,Mx,2) int64 (-1 where invalid)."""
    p = np.asarray(p, F32)
    k1 = np.asarray(kpts1, F32)
    k2 = np.asarray(kpts2, F32)
    bsz, n, m = p.shape[0], k1.shape[1], k2.shape[1]
    mx = int(max_matches)
    mk1 = np.zeros((bsz, mx, 2), F32)
    mk2 = np.zeros((bsz, mx, 2), F32)
    sc = np.zeros((bsz, mx), F32)
    ij = np.full((bsz, mx, 2), -1, np.int64)
    for b in range(bsz):
        core = p[b, :n, :m]
        jbest = core.argmax(1)
        ibest = core.argmax(0)
        best = core.max(1)
        keep = (ibest[jbest] == np.arange(n)) & (best >= F32(threshold))
        key = np.where(keep, best, F32(-1))
        order = np.argsort(-key.astype(np.float64), kind="stable")[: min(mx, n)]
        cnt = order.shape[0]
        sc[b, :cnt] = key[order]
        idx = np.zeros(mx, np.int64)
        idx[:cnt] = order
        mk1[b] = k1[b, idx]
        mk2[b] = k2[b, jbest[idx]]
        ok = sc[b] > 0
        ij[b, ok, 0] = idx[ok]
        ij[b, ok, 1] = jbest[idx][ok]
    return mk1, mk2, sc, sc > 0, ij


# --------------------------------------------------------------------------
# composite: reference feature_detection/shi_tomasi_sparse_bad_sinkhorn.py:134-182
# --------------------------------------------------------------------------
def match_pair(
    image1,
    image2,
    box_params,
    thresholds,
    max_keypoints,
    block_size=3,
    binarize=False,
    soft_binarize=True,
    temperature=10.0,
    sinkhorn_iterations=20,
    epsilon=1.0,
    unused_score=1.0,
    distance_type="l2",
    nms_radius=3,
    score_threshold=0.0,
    normalize_descriptors=True,
    border_margin=None,
    return_aux=False,
):
    """scores -> NMS -> top-k -> sparse BAD -> Sinkhorn, for both images.
    border_margin=None -> max radius of the table (7), as
    shi_tomasi_sparse_bad_sinkhorn.py:120-124."""
    if border_margin is None:
        border_margin = int(np.asarray(box_params)[:, 4].max())
    out = []
    aux = {}
    for tag, im in (("1", image1), ("2", image2)):
        s = shi_tomasi_score(im, block_size)[:, 0]
        mk = nms_mask(s, nms_radius)
        kp, ksc, ids = select_topk_keypoints(s, mk, max_keypoints, score_threshold, border_margin)
        d, a = sparse_bad(im, kp, box_params, thresholds, binarize, soft_binarize, temperature,
                          normalize_descriptors, return_aux=True)
        out.append((kp, d))
        aux["scores" + tag] = s
        aux["kscores" + tag] = ksc
        aux["ids" + tag] = ids
        aux["bad" + tag] = a
    p = sinkhorn_match(out[0][1], out[1][1], sinkhorn_iterations, epsilon, unused_score, distance_type)
    if return_aux:
        aux["desc1"], aux["desc2"] = out[0][1], out[1][1]
        return out[0][0], out[1][0], p, aux
    return out[0][0], out[1][0], p


# --------------------------------------------------------------------------
# outlier filters: reference pytorch_model/matching/sinkhorn.py:317-465
# --------------------------------------------------------------------------
def probability_ratio_filter(p, ratio_threshold=2.0):
    """matching/outlier_filters.py:11-64 on the core P (K, K): best / (second + 1e-8) >= ratio_threshold per row,
    second = second entry of the descending sort (ties count); K < 2 accepts every row (:44-47)."""
    p = np.asarray(p)
    k = p.shape[0]
    if k < 2:
        return np.ones(k, dtype=bool)
    srt = np.sort(p, axis=1)[:, ::-1]
    return srt[:, 0] / (srt[:, 1] + 1e-8) >= ratio_threshold


def dustbin_margin_filter(p, margin=0.3):
    """matching/outlier_filters.py:67-116 on the full P (K+1, K+1): max_j P[i, :K] - P[i, K] >= margin for i < K."""
    p = np.asarray(p)
    k = p.shape[0] - 1
    return p[:k, :k].max(axis=1) - p[:k, k] >= margin


def match_filters(p, ratio_threshold=None, dustbin_margin=None):
    """SinkhornMatcherWithFilters' filter stage on P (B,N+1,M+1) -> (P_filtered, valid (B,N) bool).
    :337-351 top-2 ratio (with multiplicity; second = 0 when M == 1), :370-387 best - dustbin,
    :441-463 failing rows: core * 0, dustbin entry 1."""
    p = np.asarray(p, F32).copy()
    bsz, n, m = p.shape[0], p.shape[1] - 1, p.shape[2] - 1
    rt = -1.0 if ratio_threshold is None else ratio_threshold
    dm = -1.0 if dustbin_margin is None else dustbin_margin
    core = p[:, :n, :m]
    srt = np.sort(core, axis=2)
    best = srt[:, :, -1]
    second = srt[:, :, -2] if m >= 2 else np.zeros_like(best)
    valid = np.ones((bsz, n), bool)
    if rt > 0:
        valid &= (best / (second + F32(1e-8))) >= F32(rt)
    if dm >= 0:
        valid &= (best - p[:, :n, m]) >= F32(dm)
    vf = valid.astype(F32)[:, :, None]
    p[:, :n, :m] = core * vf
    p[:, :n, m:m + 1] = (F32(1) - vf) + vf * p[:, :n, m:m + 1]
    return p, valid


# --------------------------------------------------------------------------
# orientation: reference pytorch_model/orientation/angle_estimation.py:86-172
# --------------------------------------------------------------------------
def moment_kernels(patch_size=15, sigma=2.5):
    """angle_estimation.py:97-112: (2, ps, ps) fp32 weights x*G, y*G, G = exp(-(x^2+y^2)/(2 sigma^2))."""
    if patch_size % 2 == 0:
        raise ValueError(f"patch_size must be odd, got {patch_size}")
    if sigma <= 0:
        raise ValueError(f"sigma must be positive, got {sigma}")
    c = np.arange(-(patch_size // 2), patch_size // 2 + 1, dtype=F32)
    y, x = np.meshgrid(c, c, indexing="ij")
    g = np.exp(-(x ** 2 + y ** 2) / F32(2 * sigma ** 2)).astype(F32)
    return np.stack([x * g, y * g]).astype(F32)


def angle_map(image, patch_size=15, sigma=2.5, return_moments=False):
    """angle_estimation.py:155-170: zero-padded correlation with the two moment kernels, atan2(m01, m10).
    Accumulated in float64 (the reference's fp32 conv order is oneDNN's: tolerance parity)."""
    img = np.asarray(image, F32)[:, 0].astype(np.float64)
    wk = moment_kernels(patch_size, sigma).astype(np.float64)
    n, h, w = img.shape
    half = patch_size // 2
    e = np.pad(img, ((0, 0), (half, half), (half, half)))
    m10 = np.zeros_like(img)
    m01 = np.zeros_like(img)
    for dy in range(patch_size):
        for dx in range(patch_size):
            win = e[:, dy:dy + h, dx:dx + w]
            m10 += wk[0, dy, dx] * win
            m01 += wk[1, dy, dx] * win
    ang = np.arctan2(m01, m10).astype(F32)[:, None]
    if return_moments:
        return ang, m10, m01
    return ang


def sample_nearest(field, keypoints):
    """descriptor/bad.py:490-500: grid_sample(nearest, border, align_corners) of a (B,1,H,W) map at the
    clamped keypoints -> (B,K)."""
    f = np.asarray(field, F32)
    _, _, h, w = f.shape
    kp = np.asarray(keypoints, F32)
    cy = _nearest_centre(np.clip(kp[:, :, 0], F32(0), F32(h - 1)), h)
    cx = _nearest_centre(np.clip(kp[:, :, 1], F32(0), F32(w - 1)), w)
    return np.stack([f[b, 0, cy[b], cx[b]] for b in range(f.shape[0])])


def sparse_bad_oriented(image, keypoints, theta, box_params, thresholds, binarize=False, soft_binarize=True,
                        temperature=10.0, normalize_descriptors=True, return_aux=False, sampling_mode="nearest"):
    """Oriented SparseBAD.forward (bad.py:487-574); theta (B,K) = orientation sampled at the keypoints.
    rot_dy = ox*sin + oy*cos, rot_dx = ox*cos - oy*sin in fp32 (bad.py:505-509), then as sparse_bad.
    sampling_mode="bilinear": the box-mean maps are interpolated as ATen's grid_sampler_2d does
    (align_corners, border padding); theta = 0 gives the non-oriented bilinear descriptor."""
    img = np.asarray(image, F32)
    bsz, _, h, w = img.shape
    kp = np.asarray(keypoints, F32)
    th = np.asarray(theta, F32)
    box = np.asarray(box_params).astype(np.int64)
    thr = np.asarray(thresholds, F32).astype(np.float64)
    rmax = int(box[:, 4].max())
    pad = rmax
    valid = kp[:, :, 0] >= 0
    ky = np.clip(kp[:, :, 0], F32(0), F32(h - 1))
    kx = np.clip(kp[:, :, 1], F32(0), F32(w - 1))
    ox1, ox2, oy1, oy2 = [(box[:, i] - 16).astype(F32) for i in range(4)]
    rad = box[:, 4]
    area = ((2 * rad + 1) ** 2).astype(np.float64)
    cos_t = np.cos(th).astype(F32)[:, :, None]
    sin_t = np.sin(th).astype(F32)[:, :, None]
    centered = np.empty((bsz, kp.shape[1], box.shape[0]), np.float64)
    for b in range(bsz):
        e = np.pad(img[b, 0].astype(np.float64), pad, mode="edge")
        sat = np.zeros((e.shape[0] + 1, e.shape[1] + 1), np.float64)
        sat[1:, 1:] = e.cumsum(0).cumsum(1)

        def mean_box(cy, cx):
            y0 = cy - rad[None, :] + pad
            y1 = cy + rad[None, :] + pad + 1
            x0 = cx - rad[None, :] + pad
            x1 = cx + rad[None, :] + pad + 1
            return (sat[y1, x1] - sat[y0, x1] - sat[y1, x0] + sat[y0, x0]) / area[None, :]

        def centre(oxv, oyv):
            dy = (oxv[None, :] * sin_t[b] + oyv[None, :] * cos_t[b]).astype(F32)
            dx = (oxv[None, :] * cos_t[b] - oyv[None, :] * sin_t[b]).astype(F32)
            return _nearest_centre(ky[b][:, None] + dy, h), _nearest_centre(kx[b][:, None] + dx, w)

        def bilinear(oxv, oyv):
            dy = (oxv[None, :] * sin_t[b] + oyv[None, :] * cos_t[b]).astype(F32)
            dx = (oxv[None, :] * cos_t[b] - oyv[None, :] * sin_t[b]).astype(F32)
            sy, sx = F32(2.0 / (h - 1 + 1e-8)), F32(2.0 / (w - 1 + 1e-8))
            iy = np.clip(((((ky[b][:, None] + dy) * sy - F32(1)) + F32(1)) / F32(2)) * F32(h - 1), F32(0), F32(h - 1))
            ix = np.clip(((((kx[b][:, None] + dx) * sx - F32(1)) + F32(1)) / F32(2)) * F32(w - 1), F32(0), F32(w - 1))
            y0f, x0f = np.floor(iy), np.floor(ix)
            y0, x0 = y0f.astype(np.int64), x0f.astype(np.int64)
            wy1, wx1 = (iy - y0f).astype(F32), (ix - x0f).astype(F32)
            wy0, wx0 = ((y0f + F32(1)) - iy).astype(F32), ((x0f + F32(1)) - ix).astype(F32)
            y1ok, x1ok = y0 + 1 <= h - 1, x0 + 1 <= w - 1
            y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
            acc = mean_box(y0, x0).astype(F32) * (wx0 * wy0)
            acc = acc + mean_box(y0, x1).astype(F32) * (wx1 * wy0) * x1ok
            acc = acc + mean_box(y1, x0).astype(F32) * (wx0 * wy1) * y1ok
            acc = acc + mean_box(y1, x1).astype(F32) * (wx1 * wy1) * (y1ok & x1ok)
            return acc.astype(F32)

        if sampling_mode == "bilinear":
            centered[b] = ((bilinear(ox1, oy1) - bilinear(ox2, oy2)).astype(F32) - thr[None, :].astype(F32)).astype(F32)
            continue
        c1y, c1x = centre(ox1, oy1)
        c2y, c2x = centre(ox2, oy2)
        centered[b] = mean_box(c1y, c1x) - mean_box(c2y, c2x) - thr[None, :]
    bits = centered <= 0
    if not binarize:
        desc = centered.astype(F32)
    elif soft_binarize:
        z = (-(centered.astype(F32)) * F32(temperature)).astype(F32)
        with np.errstate(over="ignore"):
            desc = (F32(1) / (F32(1) + np.exp(-z))).astype(F32)
    else:
        desc = bits.astype(F32)
    desc = desc * valid[:, :, None].astype(F32)
    if normalize_descriptors:
        nrm = np.sqrt((desc * desc).sum(-1, dtype=F32, keepdims=True)).astype(F32)
        desc = (desc / np.maximum(nrm, F32(1e-12))).astype(F32)
    if return_aux:
        return desc, {"centered": centered, "bits": bits & valid[:, :, None], "valid": valid}
    return desc


def match_pair_angle(image1, image2, box_params, thresholds, max_keypoints, block_size=5, patch_size=15, sigma=2.5,
                     binarize=False, soft_binarize=True, temperature=10.0, sinkhorn_iterations=20, epsilon=1.0,
                     unused_score=1.0, distance_type="l2", ratio_threshold=None, dustbin_margin=None, nms_radius=3,
                     score_threshold=0.0, normalize_descriptors=True, border_margin=None, with_filters=False,
                     return_aux=False):
    """ShiTomasiAngleSparseBADSinkhornMatcher[WithFilters].forward
    (feature_detection/shi_tomasi_angle_sparse_bad_sinkhorn.py:148-180, :312-340)."""
    if border_margin is None:
        border_margin = int(np.asarray(box_params)[:, 4].max())
    out, aux = [], {}
    for tag, im in (("1", image1), ("2", image2)):
        s = shi_tomasi_score(im, block_size)[:, 0]
        kp, ksc, _ = select_topk_keypoints(s, nms_mask(s, nms_radius), max_keypoints, score_threshold, border_margin)
        theta = sample_nearest(angle_map(im, patch_size, sigma), kp)
        d, a = sparse_bad_oriented(im, kp, theta, box_params, thresholds, binarize, soft_binarize, temperature,
                                   normalize_descriptors, return_aux=True)
        out.append((kp, d))
        aux["theta" + tag], aux["kscores" + tag], aux["bad" + tag] = theta, ksc, a
    p = sinkhorn_match(out[0][1], out[1][1], sinkhorn_iterations, epsilon, unused_score, distance_type)
    res = (out[0][0], out[1][0], p)
    if with_filters:
        p, valid = match_filters(p, ratio_threshold, dustbin_margin)
        res = (out[0][0], out[1][0], p, valid)
    if return_aux:
        aux["desc1"], aux["desc2"] = out[0][1], out[1][1]
        return res + (aux,)
    return res


# --------------------------------------------------------------------------
# dense BAD: reference pytorch_model/descriptor/bad.py:62-110,189-218,221-333
# --------------------------------------------------------------------------
def bad_dense(image, box_params, thresholds, binarize=False, soft_binarize=True, temperature=10.0):
    """BADDescriptor.forward (non-oriented): (B,1,H,W) -> (B,P,H,W).  Exact box sums (fp64 summed-area
    table) where the reference uses an fp32 integral image."""
    img = np.asarray(image, F32)
    bsz, _, h, w = img.shape
    box = np.asarray(box_params).astype(np.int64)
    thr = np.asarray(thresholds, F32).astype(np.float64)
    rmax = int(box[:, 4].max())
    rad = box[:, 4][:, None, None]
    area = ((2 * rad + 1) ** 2).astype(np.float64)
    yy = np.arange(h)[None, :, None]
    xx = np.arange(w)[None, None, :]
    out = np.empty((bsz, box.shape[0], h, w), np.float64)
    for b in range(bsz):
        e = np.pad(img[b, 0].astype(np.float64), rmax, mode="edge")
        sat = np.zeros((e.shape[0] + 1, e.shape[1] + 1), np.float64)
        sat[1:, 1:] = e.cumsum(0).cumsum(1)

        def mean_box(oy, ox):
            cy = np.clip(yy + oy[:, None, None], 0, h - 1) + rmax       # bad.py:81-82: centre clamped
            cx = np.clip(xx + ox[:, None, None], 0, w - 1) + rmax
            y0, y1, x0, x1 = cy - rad, cy + rad + 1, cx - rad, cx + rad + 1
            return (sat[y1, x1] - sat[y0, x1] - sat[y1, x0] + sat[y0, x0]) / area

        out[b] = mean_box(box[:, 2] - 16, box[:, 0] - 16) - mean_box(box[:, 3] - 16, box[:, 1] - 16) \
            - thr[:, None, None]
    if not binarize:
        return out.astype(F32)
    if soft_binarize:
        z = (-(out.astype(F32)) * F32(temperature)).astype(F32)
        with np.errstate(over="ignore"):
            return (F32(1) / (F32(1) + np.exp(-z))).astype(F32)
    return (out <= 0).astype(F32)


def gather_descriptors(descriptor_map, keypoints, bilinear=False):
    """bad.py:221-274 (integer truncation) / :277-333 (bilinear grid_sample, border, align_corners)."""
    dm = np.asarray(descriptor_map, F32)
    kp = np.asarray(keypoints, F32)
    bsz, d, h, w = dm.shape
    out = np.empty((bsz, kp.shape[1], d), F32)
    for b in range(bsz):
        if not bilinear:
            yi, xi = kp[b, :, 0].astype(np.int64), kp[b, :, 1].astype(np.int64)
            out[b] = dm[b][:, yi, xi].T
            continue
        gy = kp[b, :, 0] / F32(h - 1 + 1e-8) * F32(2) - F32(1)
        gx = kp[b, :, 1] / F32(w - 1 + 1e-8) * F32(2) - F32(1)
        fy = np.clip(((gy + F32(1)) / F32(2)) * F32(h - 1), F32(0), F32(h - 1))
        fx = np.clip(((gx + F32(1)) / F32(2)) * F32(w - 1), F32(0), F32(w - 1))
        y0, x0 = np.floor(fy), np.floor(fx)
        wy1, wx1 = (fy - y0).astype(F32), (fx - x0).astype(F32)
        wy0, wx0 = F32(1) - wy1, F32(1) - wx1
        iy0, ix0 = y0.astype(np.int64), x0.astype(np.int64)
        iy1, ix1 = iy0 + 1, ix0 + 1
        y1ok, x1ok = iy1 <= h - 1, ix1 <= w - 1
        iy1c, ix1c = np.minimum(iy1, h - 1), np.minimum(ix1, w - 1)
        acc = dm[b][:, iy0, ix0] * (wy0 * wx0)
        acc = acc + dm[b][:, iy0, ix1c] * (wy0 * wx1 * x1ok)
        acc = acc + dm[b][:, iy1c, ix0] * (wy1 * wx0 * y1ok)
        acc = acc + dm[b][:, iy1c, ix1c] * (wy1 * wx1 * (y1ok & x1ok))
        out[b] = acc.T
    return out


def match_pair_dense(image1, image2, box_params, thresholds, max_keypoints, block_size=3, binarize=False,
                     soft_binarize=True, temperature=10.0, sinkhorn_iterations=20, epsilon=1.0, unused_score=1.0,
                     distance_type="l2", nms_radius=3, score_threshold=0.0, normalize_descriptors=True):
    """ShiTomasiBADSinkhornMatcher.forward (feature_detection/shi_tomasi_bad_sinkhorn.py:190-219): no border
    margin; descriptors = the dense response at the keypoints (here evaluated there directly, exactly)."""
    out = []
    for im in (image1, image2):
        s = shi_tomasi_score(im, block_size)[:, 0]
        kp, _, _ = select_topk_keypoints(s, nms_mask(s, nms_radius), max_keypoints, score_threshold, 0)
        d = sparse_bad(im, kp, box_params, thresholds, binarize, soft_binarize, temperature, normalize_descriptors)
        out.append((kp, d))
    p = sinkhorn_match(out[0][1], out[1][1], sinkhorn_iterations, epsilon, unused_score, distance_type)
    return out[0][0], out[1][0], p


# --------------------------------------------------------------------------
# AKAZE (BASELINE config 4): reference pytorch_model/detector/akaze.py
# --------------------------------------------------------------------------
def _pad1(a, value=0.0):
    return np.pad(a, ((0, 0), (1, 1), (1, 1)), constant_values=value)


def _corr3(a, k):
    """3x3 cross-correlation of (N,H,W) with zero padding, fp32, row-major tap order."""
    e = _pad1(a)
    n, h, w = a.shape
    acc = np.zeros_like(a)
    for dy in range(3):
        for dx in range(3):
            if k[dy][dx] != 0:
                acc = acc + F32(k[dy][dx]) * e[:, dy:dy + h, dx:dx + w]
    return acc


_SOBEL_X = (np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], F32) / F32(8)).tolist()
_SOBEL_Y = (np.array([[-1, -2, -1], [0, 0, 0], [1, 2, 1]], F32) / F32(8)).tolist()
_HXX = (np.array([[1, -2, 1], [2, -4, 2], [1, -2, 1]], F32) / F32(16)).tolist()
_HYY = (np.array([[1, 2, 1], [-2, -4, -2], [1, 2, 1]], F32) / F32(16)).tolist()
_HXY = (np.array([[1, 0, -1], [0, 0, 0], [-1, 0, 1]], F32) / F32(4)).tolist()


def akaze_diffuse(image, iterations=3, kappa=0.05, dt=0.25):
    """NonLinearDiffusion.forward (akaze.py:98-131): (N,1,H,W) -> (N,1,H,W), fp32."""
    cur = np.asarray(image, F32)[:, 0]
    for _ in range(iterations):
        gx, gy = _corr3(cur, _SOBEL_X), _corr3(cur, _SOBEL_Y)                    # :82
        mag = np.sqrt(gx * gx + gy * gy + F32(1e-8))                             # :116
        q = mag / F32(kappa)
        c = F32(1) / (F32(1) + q * q)                                            # :96
        div = _corr3(c * gx, _SOBEL_X) + _corr3(c * gy, _SOBEL_Y)                # :125-126
        cur = (cur + F32(dt) * div).astype(F32)                                  # :129
    return cur[:, None]


def akaze_hessian_response(image):
    """HessianDetector.compute_hessian_response (akaze.py:173-198)."""
    a = np.asarray(image, F32)[:, 0]
    lxx, lyy, lxy = _corr3(a, _HXX), _corr3(a, _HYY), _corr3(a, _HXY)
    return (lxx * lyy - lxy * lxy).astype(F32)[:, None]


def akaze_hessian_scores(image, threshold=0.001, nms_size=5):
    """HessianDetector.forward (akaze.py:227-254)."""
    resp = akaze_hessian_response(image)[:, 0]
    n, h, w = resp.shape
    half = nms_size // 2
    e = np.pad(resp, ((0, 0), (half, half), (half, half)), constant_values=-np.inf)
    mx = np.full_like(resp, -np.inf)
    for dy in range(nms_size):
        for dx in range(nms_size):
            mx = np.maximum(mx, e[:, dy:dy + h, dx:dx + w])
    keep = ((resp == mx) & (resp > F32(threshold))).astype(F32)                  # :223,:245-246
    return np.maximum(resp * keep, F32(0))[:, None]                              # :249-252


def akaze_select(scale_scores, scale_orientations):
    """AKAZE.forward scale selection (akaze.py:442-451) on (S,N,1,H,W) stacks."""
    ss = np.asarray(scale_scores, F32)
    scores = ss.max(axis=0)
    mask = (ss == scores[None]).astype(F32)
    mask = mask / np.maximum(mask.sum(axis=0, keepdims=True), F32(1))
    oris = (np.asarray(scale_orientations, F32) * mask).sum(axis=0, dtype=F32)
    return scores, oris


def akaze(image, num_scales=3, diffusion_iterations=3, kappa=0.05, threshold=0.001, nms_size=5,
          orientation_patch_size=15, orientation_sigma=2.5, return_scales=False):
    """AKAZE.forward (akaze.py:384-453): (N,1,H,W) -> (scores, orientations)."""
    cur = np.asarray(image, F32)
    ss, so, ls = [], [], []
    for _ in range(num_scales):
        cur = akaze_diffuse(cur, diffusion_iterations, kappa)
        ss.append(akaze_hessian_scores(cur, threshold, nms_size))
        so.append(angle_map(cur, orientation_patch_size, orientation_sigma))
        ls.append(cur)
    scores, oris = akaze_select(np.stack(ss), np.stack(so))
    if return_scales:
        return scores, oris, np.stack(ss), np.stack(so), np.stack(ls)
    return scores, oris


def match_pair_akaze(image1, image2, box_params, thresholds, max_keypoints, num_scales=3, diffusion_iterations=3,
                     kappa=0.05, threshold=0.001, akaze_nms_size=5, orientation_patch_size=15,
                     orientation_sigma=2.5, binarize=False, soft_binarize=True, temperature=10.0,
                     sinkhorn_iterations=20, epsilon=1.0, unused_score=1.0, distance_type="l2", nms_radius=3,
                     score_threshold=0.0, normalize_descriptors=True, border_margin=None, return_aux=False,
                     scores_override=None):
    """AKAZESparseBADSinkhornMatcher.forward (feature_detection/akaze_sparse_bad_sinkhorn.py:148-196).
    scores_override: optional pair of (scores, orientations) maps to run the stages after the
    detector from (tests use it to separate detector tolerance from selection order)."""
    if border_margin is None:
        border_margin = int(np.asarray(box_params)[:, 4].max())
    out, aux = [], {}
    for idx, im in enumerate((image1, image2)):
        if scores_override is not None:
            sc, ori = scores_override[idx]
        else:
            sc, ori = akaze(im, num_scales, diffusion_iterations, kappa, threshold, akaze_nms_size,
                            orientation_patch_size, orientation_sigma)
        s = np.asarray(sc, F32)[:, 0]
        kp, ksc, _ = select_topk_keypoints(s, nms_mask(s, nms_radius), max_keypoints, score_threshold, border_margin)
        theta = sample_nearest(ori, kp)
        d = sparse_bad_oriented(im, kp, theta, box_params, thresholds, binarize, soft_binarize, temperature,
                                normalize_descriptors)
        out.append((kp, d))
        aux[f"theta{idx + 1}"], aux[f"kscores{idx + 1}"] = theta, ksc
        aux[f"scores{idx + 1}"], aux[f"ori{idx + 1}"] = sc, ori
    p = sinkhorn_match(out[0][1], out[1][1], sinkhorn_iterations, epsilon, unused_score, distance_type)
    res = (out[0][0], out[1][0], p)
    if return_aux:
        aux["desc1"], aux["desc2"] = out[0][1], out[1][1]
        return res + (aux,)
    return res


# --------------------------------------------------------------------------
# Essential-matrix head: reference pytorch_model/geometry/essential_matrix_estimator.py and
# feature_detection/shi_tomasi_angle_sparse_bad_sinkhorn_essential_matrix.py:184-271
# --------------------------------------------------------------------------
def _kth_largest(a, k, axis):
    """values[k-1] of torch.topk(a, k, dim=axis, sorted=True): the k-th largest with multiplicity."""
    return np.sort(a, axis=axis).take(-k, axis=axis)


def essential_weights(p, valid1=None, valid2=None, top_k=3):
    """Bidirectional top-k mask AND P > 0.01 on the core of P (essential_matrix_estimator.py:333-358;
    validity masking of the composite, ..._essential_matrix.py:213-218).  (N+1,M+1) -> (N,M) fp32."""
    core = np.asarray(p, F32)[:-1, :-1].copy()
    if valid1 is not None:
        core = core * np.asarray(valid1, F32)[:, None] * np.asarray(valid2, F32)[None, :]
    tr = _kth_largest(core, top_k, 1)[:, None]
    tc = _kth_largest(core, top_k, 0)[None, :]
    mask = (core >= tr) & (core >= tc) & (core > F32(0.01))
    return core * mask.astype(F32)


def _hartley(pts, w):
    """_hartley_normalization (essential_matrix_estimator.py:250-300)."""
    w_sum = w.sum(dtype=F32) + F32(1e-8)
    c = (w[:, None] * pts).sum(axis=0, dtype=F32) / w_sum
    pc = pts - c
    dist_sq = (pc ** 2).sum(axis=-1, dtype=F32)
    mean_dist = np.sqrt((w * dist_sq).sum(dtype=F32) / w_sum + F32(1e-8))
    s = np.sqrt(F32(2.0)) / (mean_dist + F32(1e-8))
    t = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1]], F32)
    return t, F32(s), c.astype(F32)


def _unit(v):
    return (v / (np.sqrt((v * v).sum(dtype=F32)) + F32(1e-8))).astype(F32)


def _det3(m):
    return (m[0, 0] * (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0])
            + m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]))


def essential_from_weights(weights, pts1_n, pts2_n, n_iter=30, n_iter_manifold=10, dtype=F32):
    """Steps 5-10 of EssentialMatrixEstimator.forward (essential_matrix_estimator.py:369-431) /
    _estimate_essential_matrix (:238-271): Hartley normalisation, Kronecker-factored normal equations,
    shifted power iteration for the minimum eigenvector, denormalisation, manifold projection."""
    w = np.asarray(weights, dtype)
    p1, p2 = np.asarray(pts1_n, dtype), np.asarray(pts2_n, dtype)
    n, m = w.shape
    t1, s1, c1 = _hartley(p1, w.sum(axis=1, dtype=dtype))
    t2, s2, c2 = _hartley(p2, w.sum(axis=0, dtype=dtype))
    f1 = np.concatenate([(p1 - c1) * s1, np.ones((n, 1), dtype)], axis=-1)
    f2 = np.concatenate([(p2 - c2) * s2, np.ones((m, 1), dtype)], axis=-1)
    big1 = (f1[:, :, None] * f1[:, None, :]).reshape(n, 9)
    big2 = (f2[:, :, None] * f2[:, None, :]).reshape(m, 9)
    m_flat = big1.T @ (w @ big2)
    m_mat = m_flat.reshape(3, 3, 3, 3).transpose(0, 2, 1, 3).reshape(9, 9).astype(dtype)
    lam = np.trace(m_mat).astype(dtype)
    m_s = (lam * np.eye(9, dtype=dtype) - m_mat).astype(dtype)
    v = (np.ones(9, dtype) / dtype(3.0)).astype(dtype)
    for _ in range(n_iter):
        v = _unit((m_s @ v).astype(dtype))
    e = (t2.T @ v.reshape(3, 3) @ t1).astype(dtype)
    # manifold projection (:175-248)
    b = (e.T @ e).astype(dtype)
    lam = np.trace(b).astype(dtype)
    v1 = (np.ones(3, dtype) / np.sqrt(dtype(3.0))).astype(dtype)
    for _ in range(n_iter_manifold):
        v1 = _unit((b @ v1).astype(dtype))
    b_s = (lam * np.eye(3, dtype=dtype) - b).astype(dtype)
    v3 = (np.ones(3, dtype) / np.sqrt(dtype(3.0))).astype(dtype)
    for _ in range(n_iter_manifold):
        v3 = _unit((b_s @ v3).astype(dtype))
    v2 = _unit(np.cross(v3, v1).astype(dtype))
    vm = np.stack([v1, v2, v3], axis=-1).astype(dtype)
    vm = (vm @ np.diag(np.array([1, 1, np.sign(_det3(vm))], dtype))).astype(dtype)
    ev0, ev1 = (e @ vm[:, 0]).astype(dtype), (e @ vm[:, 1]).astype(dtype)
    sg1, sg2 = np.sqrt((ev0 * ev0).sum(dtype=dtype)), np.sqrt((ev1 * ev1).sum(dtype=dtype))
    s_avg = (sg1 + sg2) / dtype(2.0)
    u1, u2 = ev0 / (sg1 + dtype(1e-8)), ev1 / (sg2 + dtype(1e-8))
    um = np.stack([u1, u2, np.cross(u1, u2)], axis=-1).astype(dtype)
    um = (um @ np.diag(np.array([1, 1, np.sign(_det3(um))], dtype))).astype(dtype)
    return (um @ np.diag(np.array([s_avg, s_avg, 0], dtype)) @ vm.T).astype(dtype)


def essential_matrix_grid(p, k_mat, image_shape=(32, 32), top_k=3, n_iter=30, n_iter_manifold=10):
    """EssentialMatrixEstimator.forward: feature i sits at pixel (i % W, i // W) (:103-118)."""
    h, w = image_shape
    idx = np.arange(h * w, dtype=F32)
    ph = np.stack([idx % w, idx // w, np.ones(h * w, F32)], axis=-1).astype(F32)
    pn = (ph @ np.linalg.inv(np.asarray(k_mat, F32)).astype(F32).T)[:, :2].astype(F32)
    n, m = p.shape[0] - 1, p.shape[1] - 1
    return essential_from_weights(essential_weights(p, None, None, top_k), pn[:n], pn[:m], n_iter, n_iter_manifold)


def essential_matrix_keypoints(p, kpts1, kpts2, valid1, valid2, k_mat, top_k=3, n_iter=30, n_iter_manifold=10):
    """The composites' head (..._essential_matrix.py:334-360): keypoints (K,2) as (y,x) pixels -> K^-1."""
    kinv = np.linalg.inv(np.asarray(k_mat, F32)).astype(F32)
    def norm(kp):
        kp = np.asarray(kp, F32)
        hom = np.stack([kp[:, 1], kp[:, 0], np.ones(len(kp), F32)], axis=-1)
        return (hom @ kinv.T)[:, :2].astype(F32)
    w = essential_weights(p, valid1, valid2, top_k)
    return essential_from_weights(w, norm(kpts1), norm(kpts2), n_iter, n_iter_manifold)


# --------------------------------------------------------------------------
# FAST and DoG detectors: reference pytorch_model/detector/fast.py, dog.py
# --------------------------------------------------------------------------
_FAST_OFFSETS = [(0, -3), (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3),
                 (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1), (-2, -2), (-1, -3)]     # (dy, dx), fast.py:47-52


def fast_score(image, threshold=20):
    """FASTScore.forward (fast.py:198-239): (N,1,H,W) -> {0,1} map, by the reference's own arithmetic
    (24-bit buffer, 16 windows of 9 bits)."""
    img = np.asarray(image, F32)[:, 0]
    n, h, w = img.shape
    e = np.pad(img, ((0, 0), (3, 3), (3, 3)), mode="edge")
    dark = np.zeros((n, h, w), np.int64)
    bright = np.zeros((n, h, w), np.int64)
    t = F32(float(threshold))
    for i, (dy, dx) in enumerate(_FAST_OFFSETS):
        diff = e[:, 3 + dy:3 + dy + h, 3 + dx:3 + dx + w] - img
        dark += (diff >= t).astype(np.int64) << i
        bright += (diff <= -t).astype(np.int64) << i

    def nine(bits):
        buf = bits + (bits % 256) * 65536
        hit = np.zeros(bits.shape, bool)
        for s in range(16):
            hit |= ((buf // (1 << s)) % 512) == 511
        return hit
    return (nine(dark) | nine(bright)).astype(F32)[:, None]


def gaussian_kernels_2d(num_scales=5, sigma_base=1.6, sigma_ratio=2 ** 0.5, kernel_size=None):
    """DoGDetector.__init__ (dog.py:54-98): (S, ks, ks) fp32 normalised Gaussians."""
    sigmas = [sigma_base * (sigma_ratio ** i) for i in range(num_scales)]
    if kernel_size is None:
        kernel_size = int(6 * sigmas[-1] + 1)
        if kernel_size % 2 == 0:
            kernel_size += 1
    half = kernel_size // 2
    c = np.arange(-half, half + 1, dtype=F32)
    yy, xx = np.meshgrid(c, c, indexing="ij")
    out = []
    for s in sigmas:
        k = np.exp(-(xx ** 2 + yy ** 2) / F32(2 * s ** 2)).astype(F32)
        out.append((k / k.sum(dtype=F32)).astype(F32))
    return np.stack(out)


def dog_responses(image, num_scales=5, sigma_base=1.6, sigma_ratio=2 ** 0.5, kernel_size=None):
    """DoGDetector.forward (dog.py:100-142), accumulated in float64 (separable: the normalised 2-D kernel is
    the outer product of its row sums)."""
    img = np.asarray(image, F32)[:, 0].astype(np.float64)
    k2 = gaussian_kernels_2d(num_scales, sigma_base, sigma_ratio, kernel_size).astype(np.float64)
    ks = k2.shape[-1]
    half = ks // 2
    n, h, w = img.shape
    e = np.pad(img, ((0, 0), (half, half), (half, half)), mode="edge")
    pyr = []
    for s in range(num_scales):
        w1 = k2[s].sum(axis=-1)
        tmp = np.zeros((n, h + 2 * half, w))
        for k in range(ks):
            tmp += w1[k] * e[:, :, k:k + w]
        acc = np.zeros((n, h, w))
        for k in range(ks):
            acc += w1[k] * tmp[:, k:k + h, :]
        pyr.append(acc)
    pyr = np.stack(pyr, axis=1)
    return (pyr[:, 1:] - pyr[:, :-1]).astype(F32)


def dog_score(image, **kw):
    """DoGDetectorWithScore.forward (dog.py:182-204)."""
    return np.abs(dog_responses(image, **kw)).max(axis=1, keepdims=True)


def bad_dense_oriented(image, orientation, box_params, thresholds, binarize=False, soft_binarize=True, temperature=10.0):
    """BADDescriptor.forward(x, orientation) (bad.py:112-218): every pixel is a keypoint whose angle is the
    orientation map's value there, box means sampled bilinearly: the sparse bilinear oracle over the pixel grid."""
    img = np.asarray(image, F32)
    bsz, _, h, w = img.shape
    yy, xx = np.meshgrid(np.arange(h, dtype=F32), np.arange(w, dtype=F32), indexing="ij")
    kp = np.broadcast_to(np.stack([yy.ravel(), xx.ravel()], -1)[None], (bsz, h * w, 2)).astype(F32)
    theta = np.asarray(orientation, F32).reshape(bsz, h * w)
    d = sparse_bad_oriented(img, kp, theta, box_params, thresholds, binarize, soft_binarize, temperature,
                            normalize_descriptors=False, sampling_mode="bilinear")
    return d.reshape(bsz, h, w, -1).transpose(0, 3, 1, 2).copy()
