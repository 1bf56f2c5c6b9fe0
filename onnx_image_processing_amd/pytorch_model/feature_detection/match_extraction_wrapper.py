"""MatchExtractionWrapper -- mirror of reference
pytorch_model/feature_detection/match_extraction_wrapper.py:14-113."""
import torch
from torch import nn

from ..matching.match_extraction import MutualNearestNeighborMatcher


class MatchExtractionWrapper(nn.Module):
    """Wraps any matcher returning (keypoints1, keypoints2, matching_probs, ...) and appends
    mutual-nearest-neighbour extraction: forward(image1, image2) ->
    (matched_kpts1, matched_kpts2, scores, valid_mask)."""

    def __init__(self, feature_matcher: nn.Module, max_matches: int = 100, match_threshold: float = 0.1) -> None:
        super().__init__()
        self.feature_matcher = feature_matcher
        self.match_extractor = MutualNearestNeighborMatcher(max_matches=max_matches, threshold=match_threshold)

    @torch.no_grad()
    def forward(self, image1: torch.Tensor, image2: torch.Tensor):
        out = self.feature_matcher(image1, image2)
        return self.match_extractor(out[2], out[0], out[1])
