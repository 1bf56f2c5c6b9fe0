"""AngleEstimator -- mirror of reference pytorch_model/orientation/angle_estimation.py:28-172."""
import torch
from torch import nn

from ... import ops


class AngleEstimator(nn.Module):
    """Dominant orientation from Gaussian-weighted intensity moments.

    forward(image (N,1,H,W)) -> angle map (N,1,H,W) in radians, atan2(m01, m10) with
    m10 = sum x*G*I, m01 = sum y*G*I over a patch_size^2 window (zero padding).  Constructor
    validation and the `moment_kernels` buffer follow angle_estimation.py:86-112; the K8 kernels
    read that buffer, so they multiply by the very weights the reference convolves with.
    `at_keypoints` is an extension: the angle only where the matcher needs it.
    """

    def __init__(self, patch_size: int = 15, sigma: float = 2.5):
        super().__init__()
        if patch_size % 2 == 0:
            raise ValueError(f"patch_size must be odd, got {patch_size}")
        if sigma <= 0:
            raise ValueError(f"sigma must be positive, got {sigma}")
        self.patch_size = patch_size
        self.sigma = sigma
        c = torch.arange(-(patch_size // 2), patch_size // 2 + 1, dtype=torch.float32)
        y, x = torch.meshgrid(c, c, indexing="ij")
        gaussian = torch.exp(-(x ** 2 + y ** 2) / (2 * sigma ** 2))
        self.register_buffer("moment_kernels", torch.stack([x * gaussian, y * gaussian]).unsqueeze(1))  # (2,1,ps,ps)

    @torch.no_grad()
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        return ops.angle_map(image, self.moment_kernels, self.patch_size)

    @torch.no_grad()
    def at_keypoints(self, image: torch.Tensor, keypoints: torch.Tensor) -> torch.Tensor:
        """(B,1,H,W), (B,K,2) -> (B,K): forward(image) sampled at the keypoints, without the map."""
        return ops.angle_at_keypoints(image, keypoints, self.moment_kernels, self.patch_size)
