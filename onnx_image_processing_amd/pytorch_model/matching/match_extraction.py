"""MutualNearestNeighborMatcher -- mirror of reference pytorch_model/matching/match_extraction.py."""
import torch
from torch import nn

from ... import ops


class MutualNearestNeighborMatcher(nn.Module):
    """forward(P (B,N+1,M+1), keypoints1 (B,N,2), keypoints2 (B,M,2)) ->
    (matched_kpts1 (B,Mx,2), matched_kpts2 (B,Mx,2), scores (B,Mx), valid_mask (B,Mx) bool)
    -- mutual argmax, score >= threshold, best `max_matches` (match_extraction.py:37-184).
    K7 `mi_mnn_extract`."""

    def __init__(self, max_matches: int = 100, threshold: float = 0.1) -> None:
        super().__init__()
        self.max_matches = max_matches
        self.threshold = threshold

    @torch.no_grad()
    def forward(self, P: torch.Tensor, keypoints1: torch.Tensor, keypoints2: torch.Tensor):
        return ops.mnn_extract(P, keypoints1, keypoints2, self.max_matches, self.threshold)
