"""MatchExtractionWrapper -- mirror of reference
pytorch_model/feature_detection/match_extraction_wrapper.py:14-113."""
import torch
from torch import nn

from ... import ops
from ..matching.match_extraction import MutualNearestNeighborMatcher


class MatchExtractionWrapper(nn.Module):
    """Wraps any matcher returning (keypoints1, keypoints2, matching_probs, ...) and appends
    mutual-nearest-neighbour extraction: forward(image1, image2) ->
    (matched_kpts1, matched_kpts2, scores, valid_mask).

    When the wrapped matcher offers `match_solution` (the Sinkhorn duals instead of P) and the
    problem size is supported, the matches are taken straight from the duals: same outputs bit for
    bit, without the (B,N+1,M+1) matrix ever being written (`fuse_extraction = False` restores the
    two-step form)."""

    def __init__(self, feature_matcher: nn.Module, max_matches: int = 100, match_threshold: float = 0.1) -> None:
        super().__init__()
        self.feature_matcher = feature_matcher
        self.match_extractor = MutualNearestNeighborMatcher(max_matches=max_matches, threshold=match_threshold)

        self.fuse_extraction = True

    @torch.no_grad()
    def forward(self, image1: torch.Tensor, image2: torch.Tensor):
        k = getattr(self.feature_matcher, "max_keypoints", None)
        if (self.fuse_extraction and hasattr(self.feature_matcher, "match_solution") and k is not None
                and ops.mnn_duals_supported(image1.shape[0], k, k)):
            k1, k2, sol = self.feature_matcher.match_solution(image1, image2)
            return sol.mutual_matches(k1, k2, self.match_extractor.max_matches, self.match_extractor.threshold)
        out = self.feature_matcher(image1, image2)
        return self.match_extractor(out[2], out[0], out[1])

    @torch.no_grad()
    def forward_single_call(self, image1: torch.Tensor, image2: torch.Tensor, want_keypoints: bool = False):
        """forward() as ONE C-ABI call (`mi_match_pairs`): the entry point a non-Python host binds.  Requires a
        ShiTomasiSparseBADSinkhornMatcher with hard-binarised descriptors and the L2 cost; same outputs bit for bit.
        want_keypoints: also return (keypoints1, keypoints2)."""
        fm = self.feature_matcher
        from .shi_tomasi_sparse_bad_sinkhorn import ShiTomasiSparseBADSinkhornMatcher
        from ... import _native as N
        if not isinstance(fm, ShiTomasiSparseBADSinkhornMatcher) or fm.descriptor.mode != N.MI_BAD_HARD \
                or fm.matcher.distance_type != "l2" or fm.descriptor.sampling_mode != "nearest":
            raise RuntimeError("forward_single_call covers ShiTomasiSparseBADSinkhornMatcher(binarize=True, soft_binarize=False, "
                               "distance_type='l2', sampling_mode='nearest')")
        d = fm.descriptor
        d._check(image1, None)
        out = ops.match_pairs(image1, image2, block_size=fm.corner_detector.block_size, nms_radius=fm.nms_radius,
                              max_keypoints=fm.max_keypoints, score_threshold=fm.score_threshold,
                              border_margin=fm.border_margin, pair_geom=d.pair_geom, pair_thr=d.pair_thr, plan=d._get_plan(),
                              normalize_descriptors=d.normalize_descriptors, epsilon=fm.matcher.epsilon,
                              unused_score=fm.matcher.unused_score, sinkhorn_iterations=fm.matcher.iterations,
                              max_matches=self.match_extractor.max_matches, match_threshold=self.match_extractor.threshold)
        return out if want_keypoints else out[2:]
