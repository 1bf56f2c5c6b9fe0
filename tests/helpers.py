"""Shared checker utilities (test infrastructure)."""
from __future__ import annotations

import ast
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def cfg_of(g, key="cfg") -> dict:
    return ast.literal_eval(str(g[key]))


def tie_canonical_perm(kpts: np.ndarray, kscores: np.ndarray, width: int) -> np.ndarray:
    """Permutation that re-orders one image's K keypoints into the build's tie policy
    (score descending, linear index ascending).  torch.topk's order inside a group of
    equal scores is implementation-defined; everything else about the
    order is already fixed, so this only moves entries inside tie groups.  Invalid
    entries (-1,-1 / score 0) sort last, among themselves by original position."""
    k = kpts.shape[0]
    lin = kpts[:, 0].astype(np.int64) * width + kpts[:, 1].astype(np.int64)
    invalid = kpts[:, 0] < 0
    lin = np.where(invalid, np.iinfo(np.int64).max - (k - np.arange(k)), lin)
    return np.lexsort((lin, -kscores.astype(np.float64)))


def permute_p(p: np.ndarray, perm_rows: np.ndarray, perm_cols: np.ndarray) -> np.ndarray:
    """Apply keypoint permutations to a (N+1, M+1) assignment matrix (dustbins stay last)."""
    r = np.concatenate([perm_rows, [p.shape[0] - 1]])
    c = np.concatenate([perm_cols, [p.shape[1] - 1]])
    return p[np.ix_(r, c)]


def unpack_bits(words: np.ndarray, nbits: int) -> np.ndarray:
    w = np.asarray(words, np.uint32)
    b = (w[..., None] >> np.arange(32, dtype=np.uint32)) & 1
    return b.reshape(*w.shape[:-1], -1)[..., :nbits].astype(bool)


def p_close(p: np.ndarray, ref: np.ndarray, atol: float = 1e-4):
    """north_star tolerance: |dP| <= 1e-4 on core entries; dustbin row/column entries can be
    as large as M, so there the bound is relative: 1e-4 * max(1, |P|)."""
    err = np.abs(p.astype(np.float64) - ref.astype(np.float64))
    bound = atol * np.maximum(1.0, np.abs(ref.astype(np.float64)))
    return bool((err <= bound).all()), float((err / bound).max())


def bad_tables(num_pairs: int):
    path = os.path.join(os.path.dirname(GOLDEN), "..", "onnx_image_processing_amd", "data", "bad_tables.npz")
    t = np.load(path)
    return t[f"box_{num_pairs}"], t[f"thr_{num_pairs}"]


def bits_mismatch(x, y, allowed: int, what: str = "") -> int:
    """Number of differing entries of two boolean arrays; asserts it is <= `allowed` -- the count MEASURED for that
    fixture (no blanket agreement rates).  MI_REPORT=1 prints the measured count."""
    x, y = np.asarray(x, bool), np.asarray(y, bool)
    assert x.shape == y.shape, (x.shape, y.shape)
    n = int((x != y).sum())
    if os.environ.get("MI_REPORT"):
        print(f"[bits_mismatch] {what}: {n} of {x.size} (allowed {allowed})")
    assert n <= allowed, f"{what}: {n} differing bits of {x.size}, measured allowance {allowed}"
    return n


# Measured per-fixture counts (oracle vs the recorded reference output, and HIP vs the same): the number of
# descriptor bits that may differ.  Exact-arithmetic bits vs the reference's fp32 box means ("fragile" bits, DESIGN.md
# section 2.3) and rotated box centres within rounding of x.5 (section 2.4).  Anything above these is a regression.
class _Allow(dict):
    def __missing__(self, key):
        if os.environ.get("MI_REPORT") == "measure":      # measuring a new fixture: report, do not fail
            return 1 << 60
        raise KeyError(f"no measured bit-mismatch allowance recorded for {key!r} (run once with MI_REPORT=measure)")


ALLOW = _Allow({
    # oracle vs recorded reference output (tests/test_oracle_golden.py): measured 0 everywhere
    "angle_hard_desc1": 0, "angle_hard_desc2": 0, "akaze_hard_u8": 0, "bilinear_int": 0, "bilinear_frac": 0,
    "bilinear_ori": 0, "dense_oriented_hard": 0,
    # HIP path vs oracle / vs recorded reference output (tests/test_gpu_parity.py), measured on MI355X in round 2
    # (gpurun_out/r2_test1.log): 0 everywhere
    "gpu_oriented_vs_oracle": 0, "gpu_angle_detector_vs_reference": 0, "gpu_dense_oriented_vs_reference": 0,
    "gpu_bilinear_int_vs_oracle": 0, "gpu_bilinear_int_vs_reference": 0, "gpu_bilinear_frac_vs_oracle": 0,
    "gpu_bilinear_frac_vs_reference": 0, "gpu_bilinear_ori_vs_oracle": 0, "gpu_bilinear_ori_vs_reference": 0,
    # the 64 bench pairs against the reference's recorded match sets (test_bench_pairs_match_sets_equal_the_reference):
    # pairs whose ONLY difference is a tie at the max_matches cut, each checked match by match; measured 3 (rounds 3, 4)
    "gpu_bench_pairs_cut_ties": 3,
})


def match_dict(mk1, mk2, scores, valid) -> dict:
    """{(y1, x1, y2, x2): score} of the valid matches of ONE pair."""
    return {(*map(float, a), *map(float, b)): float(s) for a, b, s, v in zip(mk1, mk2, scores, valid) if v}


def check_match_sets(got: dict, want: dict, max_matches: int, tol: float = 1e-4, mutual: dict | None = None) -> str:
    """Match-set parity of one pair, strictly.  Equal sets with scores within `tol`: "same".
    Otherwise the ONLY legitimate difference is a tie at the max_matches cut: both sets are full, and every match that
    is in one set but not the other has a score within `tol` of the cut score (the smallest kept score) -- two scores
    closer than the parity bound may swap places there.  `mutual`: the reference's complete set of mutual matches
    above the threshold with their scores; when given, a match only `got` has must be one of them, with the same score
    to `tol`.  Returns "same" or "cut"; anything else is an AssertionError naming the offending matches."""
    for k in set(got) & set(want):
        assert abs(got[k] - want[k]) <= tol, f"score of {k}: {got[k]} vs {want[k]}"
    if set(got) == set(want):
        return "same"
    assert len(got) == max_matches and len(want) == max_matches, \
        f"sets differ ({sorted(set(got) ^ set(want))}) without both being full: {len(got)} / {len(want)} of {max_matches}"
    cut = min(min(got.values()), min(want.values()))
    both = {**want, **got}
    odd = {k: both[k] for k in set(got) ^ set(want) if both[k] > cut + tol}
    assert not odd, f"matches differ away from the cut score {cut}: {odd}"
    if mutual is not None:
        for k in set(got) - set(want):
            assert k in mutual and abs(mutual[k] - got[k]) <= tol, f"{k} (score {got[k]}) is not a mutual match of the reference"
    return "cut"
