"""Rotation-aware matchers -- mirrors of reference
pytorch_model/feature_detection/shi_tomasi_angle_sparse_bad_sinkhorn.py:26-340."""
import torch
from torch import nn

from ... import _native as N
from ... import ops
from ..descriptor.bad import SparseBAD
from ..matching.sinkhorn import SinkhornMatcher, SinkhornMatcherWithFilters
from ..utils.keypoint_utils import detect_keypoints
from .shi_tomasi_angle import ShiTomasiWithAngle


class _AngleMatcherBase(nn.Module):
    """Shared composition (:148-180 / :312-340): scores -> NMS -> top-k -> angle at the keypoints ->
    oriented sparse BAD -> matcher.  The reference builds a dense angle map and samples it at the
    keypoints; here only those K angles are computed (same values, same nearest rounding)."""

    def _init_common(self, max_keypoints, block_size, patch_size, sigma, num_pairs, binarize, soft_binarize,
                     temperature, nms_radius, score_threshold, normalize_descriptors, sampling_mode, border_margin):
        self.max_keypoints = max_keypoints
        self.pair_launches = True                 # image1 / image2 share the front end's launches (False: one call per image)
        self.nms_radius = nms_radius
        self.score_threshold = score_threshold
        self.detector = ShiTomasiWithAngle(block_size=block_size, patch_size=patch_size, sigma=sigma)
        self.descriptor = SparseBAD(num_pairs=num_pairs, binarize=binarize, soft_binarize=soft_binarize,
                                    temperature=temperature, normalize_descriptors=normalize_descriptors,
                                    sampling_mode=sampling_mode)
        self.border_margin = self.descriptor.max_radius if border_margin is None else border_margin

    def _detect_describe(self, image):
        packed = self.descriptor.mode == N.MI_BAD_HARD and self.matcher.distance_type == "l2"
        scores = self.detector.shi_tomasi(image).squeeze(1)
        kp, _ = detect_keypoints(scores, self.nms_radius, self.max_keypoints, self.score_threshold,
                                 self.border_margin)
        theta = self.detector.angle_estimator.at_keypoints(image, kp)
        d = self.descriptor.forward_bits(image, kp, theta) if packed else self.descriptor(image, kp, theta)
        return kp, d, packed

    def _detect_describe_pair(self, image1, image2):
        """-> (k1, d1, k2, d2, packed).  Both images go through every front-end kernel in ONE launch (ops.ImagePair: two
        base pointers, nothing is concatenated); the kernels treat images independently."""
        if image1.shape != image2.shape:
            raise RuntimeError(f"image shapes differ: {tuple(image1.shape)} vs {tuple(image2.shape)}")
        b = image1.shape[0]
        if self.pair_launches and image1.dtype == image2.dtype:
            kp, d, packed = self._detect_describe(ops.ImagePair(image1, image2))
            return kp[:b], d[:b], kp[b:], d[b:], packed
        k1, d1, packed = self._detect_describe(image1)
        k2, d2, _ = self._detect_describe(image2)
        return k1, d1, k2, d2, packed

    def _match(self, image1, image2):
        k1, d1, k2, d2, packed = self._detect_describe_pair(image1, image2)
        if packed:
            out = self.matcher.forward_bits(d1, d2, self.descriptor.normalize_descriptors)
        else:
            out = self.matcher(d1, d2)
        return k1, k2, out


class ShiTomasiAngleSparseBADSinkhornMatcher(_AngleMatcherBase):
    """forward(image1, image2) -> (keypoints1, keypoints2, matching_probs (B,K+1,K+1)); sub-modules
    `detector`, `descriptor`, `matcher` (:79-180)."""

    def __init__(self, max_keypoints: int, block_size: int = 5, patch_size: int = 15, sigma: float = 2.5,
                 num_pairs: int = 256, binarize: bool = False, soft_binarize: bool = True, temperature: float = 10.0,
                 sinkhorn_iterations: int = 20, epsilon: float = 1.0, unused_score: float = 1.0,
                 distance_type: str = "l2", nms_radius: int = 3, score_threshold: float = 0.0,
                 normalize_descriptors: bool = True, sampling_mode: str = "nearest",
                 border_margin: int | None = None) -> None:
        super().__init__()
        self._init_common(max_keypoints, block_size, patch_size, sigma, num_pairs, binarize, soft_binarize,
                          temperature, nms_radius, score_threshold, normalize_descriptors, sampling_mode, border_margin)
        self.matcher = SinkhornMatcher(iterations=sinkhorn_iterations, epsilon=epsilon, unused_score=unused_score,
                                       distance_type=distance_type)

    @torch.no_grad()
    def forward(self, image1: torch.Tensor, image2: torch.Tensor):
        return self._match(image1, image2)

    @torch.no_grad()
    def match_solution(self, image1: torch.Tensor, image2: torch.Tensor):
        """forward() up to the Sinkhorn duals (see MatchExtractionWrapper)."""
        k1, d1, k2, d2, packed = self._detect_describe_pair(image1, image2)
        if packed:
            return k1, k2, self.matcher.solve_bits(d1, d2, self.descriptor.normalize_descriptors)
        return k1, k2, self.matcher.solve(d1, d2)


class ShiTomasiAngleSparseBADSinkhornMatcherWithFilters(_AngleMatcherBase):
    """forward(image1, image2) -> (keypoints1, keypoints2, matching_probs, valid_mask (B,K) bool)
    (:233-340): the matcher is SinkhornMatcherWithFilters(ratio_threshold, dustbin_margin)."""

    def __init__(self, max_keypoints: int, block_size: int = 5, patch_size: int = 15, sigma: float = 2.5,
                 num_pairs: int = 256, binarize: bool = False, soft_binarize: bool = True, temperature: float = 10.0,
                 sinkhorn_iterations: int = 20, epsilon: float = 1.0, unused_score: float = 1.0,
                 distance_type: str = "l2", ratio_threshold: float = None, dustbin_margin: float = None,
                 nms_radius: int = 3, score_threshold: float = 0.0, normalize_descriptors: bool = True,
                 sampling_mode: str = "nearest", border_margin: int | None = None) -> None:
        super().__init__()
        self._init_common(max_keypoints, block_size, patch_size, sigma, num_pairs, binarize, soft_binarize,
                          temperature, nms_radius, score_threshold, normalize_descriptors, sampling_mode, border_margin)
        self.matcher = SinkhornMatcherWithFilters(iterations=sinkhorn_iterations, epsilon=epsilon,
                                                  unused_score=unused_score, distance_type=distance_type,
                                                  ratio_threshold=ratio_threshold, dustbin_margin=dustbin_margin)

    @torch.no_grad()
    def forward(self, image1: torch.Tensor, image2: torch.Tensor):
        k1, k2, (probs, valid) = self._match(image1, image2)
        return k1, k2, probs, valid
