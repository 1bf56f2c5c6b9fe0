"""DoGDetector / DoGDetectorWithScore -- mirrors of reference pytorch_model/detector/dog.py:8-204."""
import math

import torch
from torch import nn

from ... import ops


def create_gaussian_kernel(sigma: float, kernel_size: int) -> torch.Tensor:
    """Normalised 2-D Gaussian (1, 1, ks, ks) (dog.py:8-29)."""
    half = kernel_size // 2
    c = torch.arange(-half, half + 1, dtype=torch.float32)
    yy, xx = torch.meshgrid(c, c, indexing="ij")
    kernel = torch.exp(-(xx ** 2 + yy ** 2) / (2 * sigma ** 2))
    kernel = kernel / kernel.sum()
    return kernel.unsqueeze(0).unsqueeze(0)


class DoGDetector(nn.Module):
    """forward(image (N,1,H,W)) -> DoG responses (N, num_scales-1, H, W): differences of consecutive
    Gaussian blurs (sigmas sigma_base * sigma_ratio^i) of the replicate-padded image.  Arguments,
    validation and the `gaussian_kernels` buffer follow dog.py:54-98; the K11 kernel blurs with the
    1-D factors of those very kernels (their row sums -- the normalised 2-D Gaussian is their outer
    product)."""

    def __init__(self, num_scales: int = 5, sigma_base: float = 1.6, sigma_ratio: float = math.sqrt(2),
                 kernel_size: int = None) -> None:
        super().__init__()
        if num_scales < 2:
            raise ValueError(f"num_scales must be at least 2, got {num_scales}")
        self.num_scales = num_scales
        self.sigma_base = sigma_base
        self.sigma_ratio = sigma_ratio
        self.sigmas = [sigma_base * (sigma_ratio ** i) for i in range(num_scales)]
        if kernel_size is None:
            kernel_size = int(6 * self.sigmas[-1] + 1)
            if kernel_size % 2 == 0:
                kernel_size += 1
        if kernel_size % 2 == 0:
            raise ValueError(f"kernel_size must be odd, got {kernel_size}")
        self.kernel_size = kernel_size
        self.padding = kernel_size // 2
        kernels = torch.cat([create_gaussian_kernel(s, kernel_size) for s in self.sigmas], dim=0)
        self.register_buffer("gaussian_kernels", kernels)
        self.register_buffer("_weights_1d", kernels[:, 0].sum(dim=-1).contiguous(), persistent=False)

    def _check(self, image: torch.Tensor) -> None:
        if image.dim() == 4 and image.shape[1] != 1:
            raise ValueError(f"Input must be grayscale (1 channel), got {image.shape[1]} channels")

    @torch.no_grad()
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        self._check(image)
        return ops.dog_responses(image, self._weights_1d)[0]

    @torch.no_grad()
    def score_map(self, image: torch.Tensor) -> torch.Tensor:
        """max over scales of |DoG| without writing the per-scale maps (what DoGDetectorWithScore needs)."""
        self._check(image)
        return ops.dog_responses(image, self._weights_1d, want_maps=False, want_score=True)[1]


class DoGDetectorWithScore(nn.Module):
    """forward(image) -> (N,1,H,W): maximum absolute DoG response over the scales (dog.py:145-204)."""

    def __init__(self, num_scales: int = 5, sigma_base: float = 1.6, sigma_ratio: float = math.sqrt(2),
                 kernel_size: int = None) -> None:
        super().__init__()
        self.dog_detector = DoGDetector(num_scales=num_scales, sigma_base=sigma_base, sigma_ratio=sigma_ratio,
                                        kernel_size=kernel_size)

    @torch.no_grad()
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        return self.dog_detector.score_map(image)
