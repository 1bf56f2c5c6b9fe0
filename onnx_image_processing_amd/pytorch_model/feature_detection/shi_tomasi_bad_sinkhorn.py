"""ShiTomasiBADSinkhornMatcher (dense-descriptor variant) -- mirror of reference
pytorch_model/feature_detection/shi_tomasi_bad_sinkhorn.py:23-219."""
import torch
from torch import nn

from ... import _native as N
from ... import ops
from ..matching.sinkhorn import SinkhornMatcher
from ..utils.keypoint_utils import detect_keypoints
from .shi_tomasi_bad import ShiTomasiBADDetector


class ShiTomasiBADSinkhornMatcher(nn.Module):
    """forward(image1, image2) -> (keypoints1, keypoints2, matching_probs (B,K+1,K+1)).

    Reference order (:190-219): dense descriptor map, NMS/top-k WITHOUT border margin, descriptors
    sampled at the keypoints, invalid rows zeroed, optional L2 normalisation, Sinkhorn.  The dense
    (B,P,H,W) map (4.2 GB per 1080p image at P=512) is never built: the same responses are
    evaluated at the K keypoints only, exactly (at integer keypoints the reference's bilinear
    sampling returns the map value up to 1e-5)."""

    def __init__(self, max_keypoints: int, block_size: int = 3, sobel_size: int = 3, num_pairs: int = 256,
                 binarize: bool = False, soft_binarize: bool = True, temperature: float = 10.0,
                 sinkhorn_iterations: int = 20, epsilon: float = 1.0, unused_score: float = 1.0,
                 distance_type: str = "l2", nms_radius: int = 3, score_threshold: float = 0.0,
                 normalize_descriptors: bool = True) -> None:
        super().__init__()
        self.max_keypoints = max_keypoints
        self.pair_launches = True                 # image1 / image2 share the front end's launches (False: one call per image)
        self.nms_radius = nms_radius
        self.score_threshold = score_threshold
        self.normalize_descriptors = normalize_descriptors
        self.detector = ShiTomasiBADDetector(block_size=block_size, sobel_size=sobel_size, num_pairs=num_pairs,
                                             binarize=binarize, soft_binarize=soft_binarize, temperature=temperature)
        self.matcher = SinkhornMatcher(iterations=sinkhorn_iterations, epsilon=epsilon, unused_score=unused_score,
                                       distance_type=distance_type)

    def _plan(self, bad):
        """The fast-path plan of the descriptor's pair table (hard bits only), built once per device."""
        if bad.mode != N.MI_BAD_HARD:
            return None
        plan = getattr(self, "_bad_plan", None)
        if plan is None or plan.device != bad.pair_geom.device:
            plan = self._bad_plan = ops.bad_plan(bad.pair_geom, bad.pair_thr)
        return plan

    def _detect_describe(self, image1: torch.Tensor, image2: torch.Tensor):
        if image1.shape != image2.shape:
            raise RuntimeError(f"image shapes differ: {tuple(image1.shape)} vs {tuple(image2.shape)}")
        bad = self.detector.descriptor
        packed = bad.mode == N.MI_BAD_HARD and self.matcher.distance_type == "l2"
        if self.pair_launches and image1.dtype == image2.dtype:
            # both images through every front-end kernel in ONE launch (ops.ImagePair: two base pointers, nothing concatenated)
            b = image1.shape[0]
            image = ops.ImagePair(image1, image2)
            scores = self.detector.corner_detector(image).squeeze(1)
            kp, _ = detect_keypoints(scores, self.nms_radius, self.max_keypoints, self.score_threshold, 0)
            del scores
            d, bits = ops.sparse_bad(image, kp, bad.pair_geom, bad.pair_thr, bad.mode, bad.temperature,
                                     self.normalize_descriptors, want_desc=not packed, want_bits=packed,
                                     plan=self._plan(bad))
            out = bits if packed else d
            return [kp[:b], kp[b:]], [out[:b], out[b:]], packed
        kpts, descs = [], []
        for image in (image1, image2):
            scores = self.detector.corner_detector(image).squeeze(1)
            kp, _ = detect_keypoints(scores, self.nms_radius, self.max_keypoints, self.score_threshold, 0)
            del scores
            kpts.append(kp)
            d, bits = ops.sparse_bad(image, kp, bad.pair_geom, bad.pair_thr, bad.mode, bad.temperature,
                                     self.normalize_descriptors, want_desc=not packed, want_bits=packed,
                                     plan=self._plan(bad))
            descs.append(bits if packed else d)
        return kpts, descs, packed

    @torch.no_grad()
    def forward(self, image1: torch.Tensor, image2: torch.Tensor):
        kpts, descs, packed = self._detect_describe(image1, image2)
        if packed:
            probs = self.matcher.forward_bits(descs[0], descs[1], self.normalize_descriptors)
        else:
            probs = self.matcher(descs[0], descs[1])
        return kpts[0], kpts[1], probs

    @torch.no_grad()
    def match_solution(self, image1: torch.Tensor, image2: torch.Tensor):
        """forward() up to the Sinkhorn duals (extension, as on the sparse matcher): MatchExtractionWrapper takes the
        mutual matches straight from them, P is never written."""
        kpts, descs, packed = self._detect_describe(image1, image2)
        if packed:
            sol = self.matcher.solve_bits(descs[0], descs[1], self.normalize_descriptors)
        else:
            sol = self.matcher.solve(descs[0], descs[1])
        return kpts[0], kpts[1], sol
