"""AKAZE detector -- mirror of reference pytorch_model/detector/akaze.py:27-453.

Same classes, constructor arguments, buffers (state_dict keys `diffusion_layers.{i}.sobel_xy`,
`diffusion_layers.{i}.sobel_xy_grouped`, `detector.hessian_kernels`,
`orientation_estimator.moment_kernels`) and forward signatures; the arithmetic runs in the K9 HIP
kernels (csrc/akaze.hip) and K8 (csrc/orient.hip).  The 3x3 weights are structural constants of
those kernels (the buffers exist for state_dict compatibility); the moment kernels are read from
the buffer.
"""
import torch
from torch import nn

from ... import ops


def _sobel_pair() -> torch.Tensor:
    sx = torch.tensor([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=torch.float32).view(1, 1, 3, 3) / 8.0
    sy = torch.tensor([[-1, -2, -1], [0, 0, 0], [1, 2, 1]], dtype=torch.float32).view(1, 1, 3, 3) / 8.0
    return torch.cat([sx, sy], dim=0)


class NonLinearDiffusion(nn.Module):
    """forward(image (N,1,H,W)) -> diffused image, `num_iterations` explicit Perona-Malik (g2) steps
    with dt = 0.25 (akaze.py:43-131)."""

    def __init__(self, num_iterations: int = 3, kappa: float = 0.05):
        super().__init__()
        self.num_iterations = num_iterations
        self.kappa = kappa
        self.register_buffer("sobel_xy", _sobel_pair())
        self.register_buffer("sobel_xy_grouped", _sobel_pair())
        self.dt = 0.25

    @torch.no_grad()
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        return ops.akaze_diffuse(image, self.num_iterations, self.kappa, self.dt)


class HessianDetector(nn.Module):
    """forward(image) -> score map: Hessian determinant at nms_size^2 local maxima above
    `threshold`, 0 elsewhere (akaze.py:146-254)."""

    def __init__(self, threshold: float = 0.001, nms_size: int = 5):
        super().__init__()
        self.threshold = threshold
        self.nms_size = nms_size
        kxx = torch.tensor([[1, -2, 1], [2, -4, 2], [1, -2, 1]], dtype=torch.float32).view(1, 1, 3, 3) / 16.0
        kyy = torch.tensor([[1, 2, 1], [-2, -4, -2], [1, 2, 1]], dtype=torch.float32).view(1, 1, 3, 3) / 16.0
        kxy = torch.tensor([[1, 0, -1], [0, 0, 0], [-1, 0, 1]], dtype=torch.float32).view(1, 1, 3, 3) / 4.0
        self.register_buffer("hessian_kernels", torch.cat([kxx, kyy, kxy], dim=0))

    @torch.no_grad()
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        return ops.akaze_hessian_scores(image, self.threshold, self.nms_size)


class OrientationEstimator(nn.Module):
    """forward(image) -> atan2(m01, m10) of Gaussian-weighted moments (akaze.py:256-314); the same
    arithmetic as orientation.AngleEstimator, so it runs on the same K8 kernels."""

    def __init__(self, patch_size: int = 15, sigma: float = 2.5):
        super().__init__()
        self.patch_size = patch_size
        self.sigma = sigma
        half = patch_size // 2
        c = torch.arange(-half, half + 1, dtype=torch.float32)
        y, x = torch.meshgrid(c, c, indexing="ij")
        gaussian = torch.exp(-(x ** 2 + y ** 2) / (2 * sigma ** 2))
        wx = (x * gaussian).view(1, 1, patch_size, patch_size)
        wy = (y * gaussian).view(1, 1, patch_size, patch_size)
        self.register_buffer("moment_kernels", torch.cat([wx, wy], dim=0))

    @torch.no_grad()
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        return ops.angle_map(image, self.moment_kernels, self.patch_size)

    @torch.no_grad()
    def at_keypoints(self, image: torch.Tensor, keypoints: torch.Tensor) -> torch.Tensor:
        return ops.angle_at_keypoints(image, keypoints, self.moment_kernels, self.patch_size)


class AKAZE(nn.Module):
    """forward(image (N,1,H,W)) -> (scores, orientations), both (N,1,H,W) (akaze.py:317-453):
    per scale i, L_i = diffusion_i(L_{i-1}); scores = max_i Hessian scores(L_i); orientations =
    mean of the orientation maps of the scales attaining that max.

    `detect(image)` (extension) returns (scores, scale_scores (S,N,1,H,W), scale_images) without
    the S dense orientation maps, `detect_select(image)` the same scores with the per-pixel set of
    scales that reach them instead of the stack, and `orientation_at_keypoints` evaluates the same
    selection at keypoints only -- what the matcher needs."""

    def __init__(self, num_scales: int = 3, diffusion_iterations: int = 3, kappa: float = 0.05,
                 threshold: float = 0.001, nms_size: int = 5, orientation_patch_size: int = 15,
                 orientation_sigma: float = 2.5):
        super().__init__()
        self.num_scales = num_scales
        self.diffusion_layers = nn.ModuleList(
            [NonLinearDiffusion(num_iterations=diffusion_iterations, kappa=kappa) for _ in range(num_scales)])
        self.detector = HessianDetector(threshold=threshold, nms_size=nms_size)
        self.orientation_estimator = OrientationEstimator(patch_size=orientation_patch_size, sigma=orientation_sigma)

    def _scale(self, i: int, cur: torch.Tensor, scores_out: torch.Tensor, image_out: torch.Tensor | None = None):
        layer = self.diffusion_layers[i]
        if layer.num_iterations > 0 and ops.akaze_kappa_fused(layer.kappa):
            # one launch per scale: diffusion steps + Hessian + NMS (streaming rolling window / LDS tile)
            cur, _ = ops.akaze_scale(cur, layer.num_iterations, layer.kappa, layer.dt, self.detector.threshold,
                                     self.detector.nms_size, scores_out=scores_out, image_out=image_out)
        else:                                            # kappa outside the fused kernels' verified range: IEEE per-step kernels
            cur = ops.akaze_diffuse(cur, layer.num_iterations, layer.kappa, layer.dt)
            ops.akaze_hessian_scores(cur, self.detector.threshold, self.detector.nms_size, out=scores_out)
            if image_out is not None:
                image_out.copy_(cur)
                cur = image_out
        return cur

    @torch.no_grad()
    def detect(self, image: torch.Tensor):
        img = ops._images(image, "image")
        n, _, h, w = img.shape
        scale_scores = torch.empty((self.num_scales, n, 1, h, w), dtype=torch.float32, device=img.device)
        scale_images = []
        cur = img
        for i in range(self.num_scales):
            cur = self._scale(i, cur, scale_scores[i])
            scale_images.append(cur)
        scores, _ = ops.akaze_combine(scale_scores, None)
        return scores, scale_scores, scale_images

    @torch.no_grad()
    def detect_select(self, image: torch.Tensor, image2: torch.Tensor | None = None):
        """detect() with the selection across scales folded into the last scale's launch (extension): returns
        (scores, attain (N,1,H,W) uint8 -- bit s: scale s reaches the maximum --, scale_images (S,N,1,H,W) stacked).
        Same scores as detect(); the stacked per-scale maps' last plane and the separate max-over-scales pass are never
        written.  image2 (a second batch of the same shape, a matcher's other image): both batches go through ONE
        launch per scale -- every output is for the 2N images, image's first."""
        img = ops._images(image, "image")
        n, _, h, w = img.shape
        img2 = ops._images(image2, "image2") if image2 is not None else None
        if img2 is not None and img2.shape != img.shape:
            raise RuntimeError(f"image shapes differ: {tuple(img.shape)} vs {tuple(img2.shape)}")
        last = self.diffusion_layers[-1]
        first = self.diffusion_layers[0]
        if self.num_scales > 8 or last.num_iterations <= 0 or not ops.akaze_kappa_fused(last.kappa) or \
                (img2 is not None and (first.num_iterations <= 0 or not ops.akaze_kappa_fused(first.kappa))):
            both = img if img2 is None else torch.cat([img, img2])
            scores, scale_scores, scale_images = self.detect(both)
            return scores, ops.akaze_attain(scale_scores, scores), torch.stack(scale_images)
        nn_ = n if img2 is None else 2 * n
        prev = torch.empty((max(self.num_scales - 1, 1), nn_, 1, h, w), dtype=torch.float32, device=img.device)
        scale_images = torch.empty((self.num_scales, nn_, 1, h, w), dtype=torch.float32, device=img.device)
        cur = img
        start = 0
        if img2 is not None and self.num_scales > 1:             # the first scale reads the two batches where they lie
            cur, _ = ops.akaze_scale_sets(img, img2, first.num_iterations, first.kappa, first.dt, self.detector.threshold,
                                          self.detector.nms_size, scores_out=prev[0], image_out=scale_images[0])
            start = 1
        elif img2 is not None:
            cur = torch.cat([img, img2])
        for i in range(start, self.num_scales - 1):
            cur = self._scale(i, cur, prev[i], scale_images[i])
        _, scores, attain = ops.akaze_scale_select(cur, last.num_iterations, last.kappa, last.dt, self.detector.threshold,
                                                   self.detector.nms_size, prev[:self.num_scales - 1] if self.num_scales > 1 else None,
                                                   image_out=scale_images[-1])
        return scores, attain, scale_images

    @torch.no_grad()
    def orientation_at_keypoints(self, scale_scores: torch.Tensor, scale_images, keypoints: torch.Tensor):
        """scale_scores: the stacked per-scale maps of detect() (float32) or the attain map of detect_select() (uint8;
        with the stacked scale images of detect_select() this is ONE launch that evaluates the moments only for the scales
        attain names)."""
        if scale_scores.dtype == torch.uint8 and torch.is_tensor(scale_images) and scale_images.is_contiguous():
            est = self.orientation_estimator
            return ops.akaze_orientation_select(scale_images, scale_scores, keypoints, est.moment_kernels, est.patch_size)
        theta = torch.stack([self.orientation_estimator.at_keypoints(im, keypoints) for im in scale_images])
        if scale_scores.dtype == torch.uint8:
            return ops.akaze_orientation_from_attain(scale_scores, theta.contiguous(), keypoints)
        return ops.akaze_orientation_at_keypoints(scale_scores, theta.contiguous(), keypoints)

    @torch.no_grad()
    def forward(self, image: torch.Tensor):
        _, scale_scores, scale_images = self.detect(image)
        scale_oris = torch.stack([self.orientation_estimator(im) for im in scale_images])
        return ops.akaze_combine(scale_scores, scale_oris.contiguous())
