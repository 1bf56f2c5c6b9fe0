"""FASTScore -- mirror of reference pytorch_model/detector/fast.py:6-266."""
import torch
from torch import nn

from ... import ops


class FASTScore(nn.Module):
    """forward(image (N,1,H,W)) -> binary corner map (N,1,H,W): 1.0 where 9 contiguous pixels of the
    radius-3 Bresenham circle are all >= centre + threshold or all <= centre - threshold.
    Constructor arguments and the `circle_offsets` / `powers_of_2` buffers follow fast.py:33-63.

    `use_nms=True`: the reference keeps `score` where it equals its (2r+1)^2 max-pool and writes 0
    elsewhere (fast.py:240-266); on a {0,1} map a pixel that is not the window maximum is already 0,
    so that step is the identity and the score is returned as is."""

    def __init__(self, threshold: int = 20, use_nms: bool = False, nms_radius: int = 3) -> None:
        super().__init__()
        self.threshold = threshold
        self.use_nms = use_nms
        self.nms_radius = nms_radius
        self.register_buffer("circle_offsets", torch.tensor(
            [[0, -3], [1, -3], [2, -2], [3, -1], [3, 0], [3, 1], [2, 2], [1, 3],
             [0, 3], [-1, 3], [-2, 2], [-3, 1], [-3, 0], [-3, -1], [-2, -2], [-1, -3]], dtype=torch.long))
        self.register_buffer("powers_of_2",
                             torch.tensor([1 << i for i in range(16)], dtype=torch.int32).view(1, 1, 1, 16))

    @torch.no_grad()
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        return ops.fast_score(image, float(self.threshold))
