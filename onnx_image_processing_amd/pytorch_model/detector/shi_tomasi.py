"""ShiTomasiScore -- mirror of reference pytorch_model/detector/shi_tomasi.py:6-112."""
import torch
from torch import nn

from ... import ops


class ShiTomasiScore(nn.Module):
    """Minimum eigenvalue of the Sobel structure tensor for every pixel.

    forward(image (N,1,H,W)) -> score (N,1,H,W) float32, computed by the K1
    `mi_corner_response` kernel.  Constructor validation and the two constant buffers
    (`sobel_xy`, `sum_kernel_grouped`) follow shi_tomasi.py:34-64 so state_dict() round-trips;
    the kernel has the taps built in and does not read them.
    """

    def __init__(self, block_size: int = 3, sobel_size: int = 3) -> None:
        super().__init__()
        if sobel_size != 3:
            raise ValueError(f"sobel_size must be 3, got {sobel_size}")
        if block_size <= 0 or block_size % 2 == 0:
            raise ValueError(f"block_size must be a positive odd integer, got {block_size}")
        self.block_size = block_size
        self.sobel_size = sobel_size
        gx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]])
        self.register_buffer("sobel_xy", torch.stack([gx, gx.t()]).unsqueeze(1))            # (2,1,3,3)
        self.register_buffer("sum_kernel_grouped", torch.ones(3, 1, block_size, block_size))  # (3,1,bs,bs)

    @torch.no_grad()
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        return ops.corner_response(image, self.block_size)
