"""AKAZESparseBADSinkhornMatcher -- mirror of reference
pytorch_model/feature_detection/akaze_sparse_bad_sinkhorn.py:24-196 (BASELINE config 4)."""
import torch
from torch import nn

from ... import _native as N
from ..descriptor.bad import SparseBAD
from ..detector.akaze import AKAZE
from ..matching.sinkhorn import SinkhornMatcher
from ..utils.keypoint_utils import detect_keypoints


class AKAZESparseBADSinkhornMatcher(nn.Module):
    """forward(image1, image2) -> (keypoints1 (B,K,2), keypoints2 (B,K,2), matching_probs
    (B,K+1,K+1)); sub-modules `detector` (AKAZE), `descriptor` (SparseBAD), `matcher`
    (SinkhornMatcher).  Composition (:148-196): AKAZE scores -> pipeline NMS(nms_radius) -> top-k
    with border margin -> rotation-aware sparse BAD with the AKAZE orientation -> Sinkhorn.  The
    reference samples its dense orientation map at the keypoints; here the same per-scale
    selection is evaluated only at those K points."""

    def __init__(self, max_keypoints: int, num_scales: int = 3, diffusion_iterations: int = 3, kappa: float = 0.05,
                 threshold: float = 0.001, akaze_nms_size: int = 5, orientation_patch_size: int = 15,
                 orientation_sigma: float = 2.5, num_pairs: int = 256, binarize: bool = False,
                 soft_binarize: bool = True, temperature: float = 10.0, sinkhorn_iterations: int = 20,
                 epsilon: float = 1.0, unused_score: float = 1.0, distance_type: str = "l2", nms_radius: int = 3,
                 score_threshold: float = 0.0, normalize_descriptors: bool = True, sampling_mode: str = "nearest",
                 border_margin: int | None = None) -> None:
        super().__init__()
        self.max_keypoints = max_keypoints
        self.nms_radius = nms_radius
        self.score_threshold = score_threshold
        self.detector = AKAZE(num_scales=num_scales, diffusion_iterations=diffusion_iterations, kappa=kappa,
                              threshold=threshold, nms_size=akaze_nms_size,
                              orientation_patch_size=orientation_patch_size, orientation_sigma=orientation_sigma)
        self.descriptor = SparseBAD(num_pairs=num_pairs, binarize=binarize, soft_binarize=soft_binarize,
                                    temperature=temperature, normalize_descriptors=normalize_descriptors,
                                    sampling_mode=sampling_mode)
        self.border_margin = self.descriptor.max_radius if border_margin is None else border_margin
        self.matcher = SinkhornMatcher(iterations=sinkhorn_iterations, epsilon=epsilon, unused_score=unused_score,
                                       distance_type=distance_type)

    def _detect_describe_pair(self, image1, image2):
        """Both images' keypoints and descriptors.  Detection -- AKAZE scales, NMS / top-k, orientation -- and description
        run on the two batches as ONE batch of 2B images (the first scale and the descriptor kernel read them where they
        lie: no concatenation)."""
        packed = self.descriptor.mode == N.MI_BAD_HARD and self.matcher.distance_type == "l2"
        b = image1.shape[0]
        scores, attain, scale_images = self.detector.detect_select(image1, image2)
        kp, _ = detect_keypoints(scores.squeeze(1), self.nms_radius, self.max_keypoints, self.score_threshold,
                                 self.border_margin)
        theta = self.detector.orientation_at_keypoints(attain, scale_images, kp)
        if image1.dtype == image2.dtype:
            from ... import ops
            pair = ops.ImagePair(image1, image2)
            d = self.descriptor.forward_bits(pair, kp, theta) if packed else self.descriptor(pair, kp, theta)
            return kp[:b], d[:b], kp[b:], d[b:], packed
        out = []
        for im, k, t in ((image1, kp[:b], theta[:b]), (image2, kp[b:], theta[b:])):
            k, t = k.contiguous(), t.contiguous()
            out.append((k, self.descriptor.forward_bits(im, k, t) if packed else self.descriptor(im, k, t)))
        return out[0][0], out[0][1], out[1][0], out[1][1], packed

    @torch.no_grad()
    def forward(self, image1: torch.Tensor, image2: torch.Tensor):
        if image1.shape != image2.shape:
            raise RuntimeError(f"image shapes differ: {tuple(image1.shape)} vs {tuple(image2.shape)}")
        k1, d1, k2, d2, packed = self._detect_describe_pair(image1, image2)
        if packed:
            probs = self.matcher.forward_bits(d1, d2, self.descriptor.normalize_descriptors)
        else:
            probs = self.matcher(d1, d2)
        return k1, k2, probs

    @torch.no_grad()
    def match_solution(self, image1: torch.Tensor, image2: torch.Tensor):
        """forward() up to the Sinkhorn duals (see MatchExtractionWrapper)."""
        if image1.shape != image2.shape:
            raise RuntimeError(f"image shapes differ: {tuple(image1.shape)} vs {tuple(image2.shape)}")
        k1, d1, k2, d2, packed = self._detect_describe_pair(image1, image2)
        if packed:
            return k1, k2, self.matcher.solve_bits(d1, d2, self.descriptor.normalize_descriptors)
        return k1, k2, self.matcher.solve(d1, d2)
