"""NMS + top-k keypoint selection -- mirror of reference pytorch_model/utils/keypoint_utils.py."""
import torch

from ... import ops


@torch.no_grad()
def apply_nms_maxpool(scores: torch.Tensor, nms_radius: int) -> torch.Tensor:
    """scores (B,H,W) -> float mask (B,H,W), 1.0 where the pixel is a (2r+1)^2 window maximum
    (keypoint_utils.py:12-44).  K2 `mi_nms_mask`."""
    return ops.nms_mask(scores, nms_radius)


@torch.no_grad()
def select_topk_keypoints(
    scores: torch.Tensor,
    nms_mask: torch.Tensor,
    max_keypoints: int,
    score_threshold: float = 0.0,
    border_margin: int = 0,
) -> tuple[torch.Tensor, torch.Tensor]:
    """(scores, nms_mask) -> keypoints (B,K,2) as (y,x) padded with (-1,-1), scores (B,K)
    (keypoint_utils.py:47-117).  Ties are ordered (score desc, linear index asc); torch.topk
    leaves that order unspecified.  K2 `mi_select_candidates` + K3 `mi_topk_keypoints`."""
    return ops.select_topk(scores, nms_mask, max_keypoints, score_threshold, border_margin)


@torch.no_grad()
def detect_keypoints(
    scores: torch.Tensor,
    nms_radius: int,
    max_keypoints: int,
    score_threshold: float = 0.0,
    border_margin: int = 0,
) -> tuple[torch.Tensor, torch.Tensor]:
    """apply_nms_maxpool + select_topk_keypoints in one pass over the score map (the mask is
    never written).  Same result as calling the two functions above."""
    return ops.nms_topk(scores, nms_radius, max_keypoints, score_threshold, border_margin)
