"""Matchers with the essential-matrix head -- mirrors of reference
pytorch_model/feature_detection/shi_tomasi_angle_sparse_bad_sinkhorn_essential_matrix.py:34-361 and
pytorch_model/feature_detection/akaze_sparse_bad_sinkhorn_essential_matrix.py:34-378."""
import torch
from torch import nn

from ... import _native as N
from ... import ops
from ..descriptor.bad import SparseBAD
from ..detector.akaze import AKAZE
from ..geometry.essential_matrix_estimator import EssentialMatrixEstimator
from ..matching.sinkhorn import SinkhornMatcher
from ..utils.keypoint_utils import detect_keypoints
from .shi_tomasi_angle import ShiTomasiWithAngle


class _EssentialHead(nn.Module):
    """Shared tail (:277-361): keypoints -> rotation-aware BAD -> Sinkhorn P -> E from the actual
    keypoint positions (K^-1 [x, y, 1], validity = keypoint score > 0).  forward returns
    (keypoints1, keypoints2, matching_probs, E); E is (3, 3) for a batch of one pair, as in the
    reference (which requires batch 1), and (B, 3, 3) otherwise."""

    def _init_tail(self, K, max_keypoints, num_pairs, binarize, soft_binarize, temperature, sinkhorn_iterations,
                   epsilon, unused_score, distance_type, nms_radius, score_threshold, normalize_descriptors,
                   sampling_mode, border_margin, top_k, n_iter, n_iter_manifold):
        self.max_keypoints = max_keypoints
        self.pair_launches = True                 # image1 / image2 share the front end's launches (False: one call per image)
        self.pair_launches_by_pointer = isinstance(self, ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix)
        self.nms_radius = nms_radius
        self.score_threshold = score_threshold
        self.top_k = top_k
        self.descriptor = SparseBAD(num_pairs=num_pairs, binarize=binarize, soft_binarize=soft_binarize,
                                    temperature=temperature, normalize_descriptors=normalize_descriptors,
                                    sampling_mode=sampling_mode)
        self.border_margin = self.descriptor.max_radius if border_margin is None else border_margin
        self.matcher = SinkhornMatcher(iterations=sinkhorn_iterations, epsilon=epsilon, unused_score=unused_score,
                                       distance_type=distance_type)
        # image_shape=(1, 1): the estimator's pixel grid is a placeholder, actual keypoints are used
        self.estimator = EssentialMatrixEstimator(K=K, image_shape=(1, 1), top_k=top_k, n_iter=n_iter,
                                                  n_iter_manifold=n_iter_manifold)
        K_f = K.float()
        self.register_buffer("K_inv", torch.linalg.inv(K_f.cpu()).to(K_f.device))

    def _detect(self, image):
        """-> (keypoints (B,K,2), keypoint scores (B,K), per-keypoint angles (B,K))"""
        raise NotImplementedError

    def _describe(self, image):
        packed = self.descriptor.mode == N.MI_BAD_HARD and self.matcher.distance_type == "l2"
        kp, ksc, theta = self._detect(image)
        d = self.descriptor.forward_bits(image, kp, theta) if packed else self.descriptor(image, kp, theta)
        return kp, ksc, d, packed

    def _normalised(self, kp: torch.Tensor) -> torch.Tensor:
        return ops.normalise_keypoints(kp, self.K_inv)                                        # (y,x) -> K^-1 [x, y, 1]

    def _solve(self, image1: torch.Tensor, image2: torch.Tensor, want_p: bool):
        """-> (k1, k2, P or None, E (B,3,3), solution or None).  Hard-binarised descriptors with the L2 cost (the export-
        CLI configuration) take the packed route: the head reads the uint16 dot products + the duals
        (`mi_essential_matrix_dots`), P is written only when the caller wants it and never read back."""
        from ..matching.sinkhorn import SinkhornSolution
        if image1.shape != image2.shape:
            raise RuntimeError(f"image shapes differ: {tuple(image1.shape)} vs {tuple(image2.shape)}")
        b = image1.shape[0]
        if self.pair_launches and self.pair_launches_by_pointer and image1.dtype == image2.dtype:
            # both images through every front-end kernel in ONE launch (ops.ImagePair: two base pointers, nothing is
            # concatenated): half the launches / graph nodes, one tail per stage instead of two
            kj, sj, dj, packed = self._describe(ops.ImagePair(image1, image2))
        elif self.pair_launches and b <= 8:
            # (front ends without two-pointer entry points: few pairs per call share the launches through one small copy)
            kj, sj, dj, packed = self._describe(torch.cat([image1, image2], dim=0))
        else:
            kj = None
            k1, s1, d1, packed = self._describe(image1)
            k2, s2, d2, _ = self._describe(image2)
            q1, q2, ok1, ok2 = self._normalised(k1.float()), self._normalised(k2.float()), s1 > 0, s2 > 0
        if kj is not None:                                   # the stacked keypoints: normalised and validity-tested once
            qj, okj = self._normalised(kj.float()), sj > 0
            k1, k2, d1, d2, q1, q2, ok1, ok2 = kj[:b], kj[b:], dj[:b], dj[b:], qj[:b], qj[b:], okj[:b], okj[b:]
        m = self.matcher
        if packed and m.use_dot_storage and ops.dots_supported(d1.shape[0], d1.shape[1], d2.shape[1], m.epsilon) \
                and self.estimator.top_k <= 8:
            probs, u, v, state = ops.sinkhorn_bits(d1, d2, self.descriptor.normalize_descriptors, m.epsilon, m.unused_score,
                                                   m.iterations, want_p=want_p, return_state=True)
            sol = SinkhornSolution("dots", state, d2.shape[1], state[3], m.epsilon, u, v)
            e = self.estimator.estimate_from_solution(sol, q1, q2, ok1, ok2)
            return k1, k2, probs, e, sol
        probs = m.forward_bits(d1, d2, self.descriptor.normalize_descriptors) if packed else m(d1, d2)
        return k1, k2, probs, self.estimator.estimate(probs, q1, q2, ok1, ok2), None

    @torch.no_grad()
    def forward(self, image1: torch.Tensor, image2: torch.Tensor):
        k1, k2, probs, e, _ = self._solve(image1, image2, want_p=True)
        return k1, k2, probs, (e[0] if e.shape[0] == 1 else e)

    @torch.no_grad()
    def essential(self, image1: torch.Tensor, image2: torch.Tensor):
        """forward() without its third output (extension): (keypoints1, keypoints2, E (B,3,3)); on the packed route the
        (B,K+1,K+1) matrix is never written."""
        k1, k2, _, e, _ = self._solve(image1, image2, want_p=False)
        return k1, k2, e

    @torch.no_grad()
    def match_and_essential(self, image1: torch.Tensor, image2: torch.Tensor, max_matches: int = 100,
                            threshold: float = 0.1):
        """What a visual-odometry host consumes per frame pair (reference sample/visual_odometry.py:520-613: E for the
        pose, mutual matches for the inlier test), all from the Sinkhorn solution (extension): (matched_kpts1,
        matched_kpts2, scores, valid, E (B,3,3)) = MutualNearestNeighborMatcher(forward()[2]) + forward()[3], without
        P on the packed route."""
        k1, k2, probs, e, sol = self._solve(image1, image2, want_p=False)
        if sol is not None and ops.mnn_duals_supported(k1.shape[0], k1.shape[1], k2.shape[1]):
            return (*sol.mutual_matches(k1, k2, max_matches, threshold), e)
        if probs is None:
            k1, k2, probs, e, _ = self._solve(image1, image2, want_p=True)
        return (*ops.mnn_extract(probs, k1, k2, max_matches, threshold), e)


class ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix(_EssentialHead):
    """Shi-Tomasi + angle front end (..._essential_matrix.py:98-182 constructor)."""

    def __init__(self, K: torch.Tensor, max_keypoints: int, block_size: int = 5, patch_size: int = 15,
                 sigma: float = 2.5, num_pairs: int = 256, binarize: bool = False, soft_binarize: bool = True,
                 temperature: float = 10.0, sinkhorn_iterations: int = 20, epsilon: float = 1.0,
                 unused_score: float = 1.0, distance_type: str = "l2", nms_radius: int = 3,
                 score_threshold: float = 0.0, normalize_descriptors: bool = True, sampling_mode: str = "nearest",
                 border_margin: int | None = None, top_k: int = 3, n_iter: int = 30, n_iter_manifold: int = 10) -> None:
        super().__init__()
        self.detector = ShiTomasiWithAngle(block_size=block_size, patch_size=patch_size, sigma=sigma)
        self._init_tail(K, max_keypoints, num_pairs, binarize, soft_binarize, temperature, sinkhorn_iterations, epsilon,
                        unused_score, distance_type, nms_radius, score_threshold, normalize_descriptors, sampling_mode,
                        border_margin, top_k, n_iter, n_iter_manifold)

    def _detect(self, image):
        scores = self.detector.shi_tomasi(image).squeeze(1)
        kp, ksc = detect_keypoints(scores, self.nms_radius, self.max_keypoints, self.score_threshold,
                                   self.border_margin)
        return kp, ksc, self.detector.angle_estimator.at_keypoints(image, kp)


class AKAZESparseBADSinkhornWithEssentialMatrix(_EssentialHead):
    """AKAZE front end (akaze_sparse_bad_sinkhorn_essential_matrix.py:107-199 constructor)."""

    def __init__(self, K: torch.Tensor, max_keypoints: int, num_scales: int = 3, diffusion_iterations: int = 3,
                 kappa: float = 0.05, threshold: float = 0.001, akaze_nms_size: int = 5,
                 orientation_patch_size: int = 15, orientation_sigma: float = 2.5, num_pairs: int = 256,
                 binarize: bool = False, soft_binarize: bool = True, temperature: float = 10.0,
                 sinkhorn_iterations: int = 20, epsilon: float = 1.0, unused_score: float = 1.0,
                 distance_type: str = "l2", nms_radius: int = 3, score_threshold: float = 0.0,
                 normalize_descriptors: bool = True, sampling_mode: str = "nearest",
                 border_margin: int | None = None, top_k: int = 3, n_iter: int = 30, n_iter_manifold: int = 10) -> None:
        super().__init__()
        self.detector = AKAZE(num_scales=num_scales, diffusion_iterations=diffusion_iterations, kappa=kappa,
                              threshold=threshold, nms_size=akaze_nms_size,
                              orientation_patch_size=orientation_patch_size, orientation_sigma=orientation_sigma)
        self._init_tail(K, max_keypoints, num_pairs, binarize, soft_binarize, temperature, sinkhorn_iterations, epsilon,
                        unused_score, distance_type, nms_radius, score_threshold, normalize_descriptors, sampling_mode,
                        border_margin, top_k, n_iter, n_iter_manifold)

    def _detect(self, image):
        scores, scale_scores, scale_images = self.detector.detect_select(image)
        kp, ksc = detect_keypoints(scores.squeeze(1), self.nms_radius, self.max_keypoints, self.score_threshold,
                                   self.border_margin)
        return kp, ksc, self.detector.orientation_at_keypoints(scale_scores, scale_images, kp)
