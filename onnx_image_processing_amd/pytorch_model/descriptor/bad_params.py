"""Learned BAD pair tables (data).  Values: reference descriptor/bad_params.py:4-1568, stored
as onnx_image_processing_amd/data/bad_tables.npz by tools/extract_bad_tables.py."""
import os

import numpy as np
import torch

_TABLES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "data", "bad_tables.npz")


def _get_bad_learned_params(num_pairs: int) -> tuple[torch.Tensor, torch.Tensor]:
    """(box_params (P,5) float32 rows (x1,x2,y1,y2,r), thresholds (P,) float32)."""
    if num_pairs not in (256, 512):
        raise ValueError(f"num_pairs must be 256 or 512 to use learned BAD patterns, got {num_pairs}")
    with np.load(_TABLES) as t:
        box = torch.from_numpy(t[f"box_{num_pairs}"].astype(np.float32))
        thr = torch.from_numpy(t[f"thr_{num_pairs}"].astype(np.float32))
    return box, thr
