"""ShiTomasiBADDetector -- mirror of reference pytorch_model/feature_detection/shi_tomasi_bad.py:20-89."""
import torch
from torch import nn

from ..descriptor.bad import BADDescriptor
from ..detector.shi_tomasi import ShiTomasiScore


class ShiTomasiBADDetector(nn.Module):
    """forward(image) -> (scores (N,1,H,W), dense descriptors (N,num_pairs,H,W)); sub-modules
    `corner_detector`, `descriptor`."""

    def __init__(self, block_size: int = 3, sobel_size: int = 3, num_pairs: int = 256, binarize: bool = False,
                 soft_binarize: bool = True, temperature: float = 10.0) -> None:
        super().__init__()
        self.corner_detector = ShiTomasiScore(block_size=block_size, sobel_size=sobel_size)
        self.descriptor = BADDescriptor(num_pairs=num_pairs, binarize=binarize, soft_binarize=soft_binarize,
                                        temperature=temperature)

    @torch.no_grad()
    def forward(self, image: torch.Tensor):
        return self.corner_detector(image), self.descriptor(image)
