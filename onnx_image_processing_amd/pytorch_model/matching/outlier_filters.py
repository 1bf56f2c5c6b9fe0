"""Outlier filters on Sinkhorn results -- mirror of reference pytorch_model/matching/outlier_filters.py:11-116.

The reference's two functions are numpy post-processing of an exported model's output.  Here they accept BOTH forms:
a numpy array (the reference's own call pattern, sample/image_matching.py:49-118: the matrix is uploaded to the current
GPU, the mask comes back as a numpy bool array) or a GPU tensor (the matrix never has to leave the device; a bool tensor
comes back).  Either way the masks are computed by the K7 kernel behind `mi_match_filter_masks` (the arithmetic of
`SinkhornMatcherWithFilters`, masks only).  Same argument names, defaults, shapes and edge cases; fp32 like the
reference applied to the fp32 model output."""
import numpy as np
import torch

from ... import ops


def _as_device_tensor(P):
    """-> (tensor on the GPU, came_as_numpy)"""
    if isinstance(P, np.ndarray):
        if not torch.cuda.is_available():
            raise RuntimeError("outlier filters run on the GPU (there is no CPU path); no GPU is available")
        return torch.from_numpy(np.ascontiguousarray(P, dtype=np.float32)).cuda(), True
    return P, False


def _back(mask: torch.Tensor, as_numpy: bool):
    return mask.cpu().numpy() if as_numpy else mask


@torch.no_grad()
def probability_ratio_filter(P, ratio_threshold: float = 2.0):
    """P (K, K) core probabilities (dustbin excluded) -> bool (K,): best / (second best + 1e-8) >= ratio_threshold
    per row (outlier_filters.py:11-64; K < 2 accepts every row, :44-47)."""
    P, as_numpy = _as_device_tensor(P)
    return _back(_probability_ratio_filter(P, ratio_threshold), as_numpy)


def _probability_ratio_filter(P: torch.Tensor, ratio_threshold: float) -> torch.Tensor:
    if P.dim() != 2:
        raise RuntimeError(f"P must have shape (K, K), got {tuple(P.shape)}")
    k = P.shape[0]
    if k < 2:
        return torch.ones((k,), dtype=torch.bool, device=P.device)
    if ratio_threshold <= 0:                                   # every ratio of probabilities is >= a non-positive bound
        return torch.ones((k,), dtype=torch.bool, device=P.device)
    return ops.match_filter_masks(P.unsqueeze(0), False, ratio_threshold, -1.0)[0]


@torch.no_grad()
def dustbin_margin_filter(P, margin: float = 0.3):
    """P (K+1, K+1) full Sinkhorn matrix -> bool (K,): max_j P[i, :K] - P[i, K] >= margin (outlier_filters.py:67-116)."""
    P, as_numpy = _as_device_tensor(P)
    return _back(_dustbin_margin_filter(P, margin), as_numpy)


def _dustbin_margin_filter(P: torch.Tensor, margin: float) -> torch.Tensor:
    if P.dim() != 2 or P.shape[0] < 2 or P.shape[1] < 2:
        raise RuntimeError(f"P must have shape (K+1, K+1), got {tuple(P.shape)}")
    if margin < 0:
        # the kernel's convention "negative = disabled" does not apply to the reference function: shift instead
        core_best = ops.core_maxima(P.unsqueeze(0))[0][0]
        return (core_best - P[:-1, -1].float()) >= margin
    return ops.match_filter_masks(P.unsqueeze(0), True, -1.0, margin)[0]
