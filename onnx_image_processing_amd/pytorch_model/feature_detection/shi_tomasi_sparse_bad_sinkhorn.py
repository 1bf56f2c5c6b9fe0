"""ShiTomasiSparseBADSinkhornMatcher -- mirror of reference
pytorch_model/feature_detection/shi_tomasi_sparse_bad_sinkhorn.py:27-182."""
import torch
from torch import nn

from ... import _native as N
from ... import ops
from ..descriptor.bad import SparseBAD
from ..detector.shi_tomasi import ShiTomasiScore
from ..matching.sinkhorn import SinkhornMatcher
from ..utils.keypoint_utils import detect_keypoints


class ShiTomasiSparseBADSinkhornMatcher(nn.Module):
    """forward(image1, image2 (B,1,H,W)) -> (keypoints1 (B,K,2), keypoints2 (B,K,2),
    matching_probs (B,K+1,K+1)).

    Sub-modules keep the reference's names (`corner_detector`, `descriptor`, `matcher`) and
    `border_margin=None` resolves to the descriptor's max radius (:120-124).  Composition order
    is the reference's (:156-180): scores -> NMS -> top-k -> sparse BAD -> Sinkhorn, both images
    of the batch going through each kernel in one launch.  With hard binarisation the
    descriptors stay packed bits from K4 to K5 (exact integer dot products) and the float
    descriptor tensors are never materialised.
    """

    def __init__(
        self,
        max_keypoints: int,
        block_size: int = 3,
        sobel_size: int = 3,
        num_pairs: int = 256,
        binarize: bool = False,
        soft_binarize: bool = True,
        temperature: float = 10.0,
        sinkhorn_iterations: int = 20,
        epsilon: float = 1.0,
        unused_score: float = 1.0,
        distance_type: str = "l2",
        nms_radius: int = 3,
        score_threshold: float = 0.0,
        normalize_descriptors: bool = True,
        sampling_mode: str = "nearest",
        border_margin: int | None = None,
    ) -> None:
        super().__init__()
        self.max_keypoints = max_keypoints
        self.pair_launches = True                 # image1 / image2 share the front end's launches (False: one call per image)
        self.nms_radius = nms_radius
        self.score_threshold = score_threshold
        self.corner_detector = ShiTomasiScore(block_size=block_size, sobel_size=sobel_size)
        self.descriptor = SparseBAD(
            num_pairs=num_pairs,
            binarize=binarize,
            soft_binarize=soft_binarize,
            temperature=temperature,
            normalize_descriptors=normalize_descriptors,
            sampling_mode=sampling_mode,
        )
        self.border_margin = self.descriptor.max_radius if border_margin is None else border_margin
        self.matcher = SinkhornMatcher(
            iterations=sinkhorn_iterations,
            epsilon=epsilon,
            unused_score=unused_score,
            distance_type=distance_type,
        )

    def _detect_describe(self, image1: torch.Tensor, image2: torch.Tensor):
        if image1.shape != image2.shape:
            raise RuntimeError(f"image shapes differ: {tuple(image1.shape)} vs {tuple(image2.shape)}")
        packed = self.descriptor.mode == N.MI_BAD_HARD and self.matcher.distance_type == "l2"
        b = image1.shape[0]
        if self.pair_launches and image1.dtype == image2.dtype:
            # both images through every front-end kernel in ONE launch (ops.ImagePair: two base pointers, nothing is
            # concatenated): half the launches / graph nodes and one tail per stage instead of two; the kernels treat
            # images independently, so the halves are what the separate calls give
            image = ops.ImagePair(image1, image2)
            scores = self.corner_detector(image).squeeze(1)
            kp, _ = detect_keypoints(scores, self.nms_radius, self.max_keypoints, self.score_threshold, self.border_margin)
            del scores
            d = self.descriptor.forward_bits(image, kp) if packed else self.descriptor(image, kp)
            return [kp[:b], kp[b:]], [d[:b], d[b:]], packed
        kpts, descs = [], []
        for image in (image1, image2):                                       # no concatenation: images stay in place
            scores = self.corner_detector(image).squeeze(1)
            kp, _ = detect_keypoints(scores, self.nms_radius, self.max_keypoints, self.score_threshold,
                                     self.border_margin)
            del scores
            kpts.append(kp)
            descs.append(self.descriptor.forward_bits(image, kp) if packed else self.descriptor(image, kp))
        return kpts, descs, packed

    @torch.no_grad()
    def forward(self, image1: torch.Tensor, image2: torch.Tensor):
        kpts, descs, packed = self._detect_describe(image1, image2)
        if packed:
            probs = self.matcher.forward_bits(descs[0], descs[1], self.descriptor.normalize_descriptors)
        else:
            probs = self.matcher(descs[0], descs[1])
        return kpts[0], kpts[1], probs

    @torch.no_grad()
    def match_solution(self, image1: torch.Tensor, image2: torch.Tensor):
        """forward() up to the Sinkhorn duals: (keypoints1, keypoints2, SinkhornSolution).  Used by
        MatchExtractionWrapper, which only needs the mutual matches, so P is never written."""
        kpts, descs, packed = self._detect_describe(image1, image2)
        if packed:
            sol = self.matcher.solve_bits(descs[0], descs[1], self.descriptor.normalize_descriptors)
        else:
            sol = self.matcher.solve(descs[0], descs[1])
        return kpts[0], kpts[1], sol
