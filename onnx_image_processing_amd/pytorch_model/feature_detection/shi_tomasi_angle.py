"""Shi-Tomasi + angle composites -- mirrors of reference
pytorch_model/feature_detection/shi_tomasi_angle.py:23-356."""
import torch
from torch import nn

from ..descriptor.bad import SparseBAD
from ..detector.shi_tomasi import ShiTomasiScore
from ..orientation.angle_estimation import AngleEstimator
from ..utils.keypoint_utils import detect_keypoints


class ShiTomasiWithAngle(nn.Module):
    """forward(image) -> (scores (N,1,H,W), angles (N,1,H,W)); sub-modules `shi_tomasi`,
    `angle_estimator` (shi_tomasi_angle.py:52-98)."""

    def __init__(self, block_size: int = 5, sobel_size: int = 3, patch_size: int = 15, sigma: float = 2.5):
        super().__init__()
        self.shi_tomasi = ShiTomasiScore(block_size=block_size, sobel_size=sobel_size)
        self.angle_estimator = AngleEstimator(patch_size=patch_size, sigma=sigma)

    @torch.no_grad()
    def forward(self, image: torch.Tensor):
        return self.shi_tomasi(image), self.angle_estimator(image)


class ShiTomasiAngleSparseBAD(nn.Module):
    """forward(image, keypoints) -> (scores, angles, descriptors); `detect_and_orient`, `describe`
    (shi_tomasi_angle.py:148-243)."""

    def __init__(self, block_size: int = 5, patch_size: int = 15, sigma: float = 2.5, num_pairs: int = 256,
                 binarize: bool = False, soft_binarize: bool = True, temperature: float = 10.0,
                 normalize_descriptors: bool = True, sampling_mode: str = "nearest"):
        super().__init__()
        self.detector = ShiTomasiWithAngle(block_size=block_size, patch_size=patch_size, sigma=sigma)
        self.descriptor = SparseBAD(num_pairs=num_pairs, binarize=binarize, soft_binarize=soft_binarize,
                                    temperature=temperature, normalize_descriptors=normalize_descriptors,
                                    sampling_mode=sampling_mode)

    def detect_and_orient(self, image: torch.Tensor):
        return self.detector(image)

    def describe(self, image: torch.Tensor, keypoints: torch.Tensor, orientation: torch.Tensor):
        return self.descriptor(image, keypoints, orientation)

    @torch.no_grad()
    def forward(self, image: torch.Tensor, keypoints: torch.Tensor):
        scores, angles = self.detect_and_orient(image)
        return scores, angles, self.describe(image, keypoints, angles)


class ShiTomasiAngleSparseBADDetector(nn.Module):
    """forward(image) -> (keypoints (B,K,2), scores (B,K), descriptors (B,K,P)); no border margin
    (shi_tomasi_angle.py:289-356).  The dense angle map is not built: angles are computed at the
    selected keypoints only."""

    def __init__(self, max_keypoints: int, block_size: int = 5, patch_size: int = 15, sigma: float = 2.5,
                 num_pairs: int = 256, binarize: bool = False, soft_binarize: bool = True, temperature: float = 10.0,
                 normalize_descriptors: bool = True, sampling_mode: str = "nearest", nms_radius: int = 3,
                 score_threshold: float = 0.0) -> None:
        super().__init__()
        self.max_keypoints = max_keypoints
        self.nms_radius = nms_radius
        self.score_threshold = score_threshold
        self.model = ShiTomasiAngleSparseBAD(block_size=block_size, patch_size=patch_size, sigma=sigma,
                                             num_pairs=num_pairs, binarize=binarize, soft_binarize=soft_binarize,
                                             temperature=temperature, normalize_descriptors=normalize_descriptors,
                                             sampling_mode=sampling_mode)

    @torch.no_grad()
    def forward(self, image: torch.Tensor):
        score_map = self.model.detector.shi_tomasi(image).squeeze(1)
        keypoints, scores = detect_keypoints(score_map, self.nms_radius, self.max_keypoints, self.score_threshold)
        theta = self.model.detector.angle_estimator.at_keypoints(image, keypoints)
        return keypoints, scores, self.model.descriptor(image, keypoints, theta)
