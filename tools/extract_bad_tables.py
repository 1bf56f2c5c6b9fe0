#!/usr/bin/env python3
"""Extract the learned BAD pair tables (constants, i.e. data) from the reference.

Runs ONLY in the build container (needs /root/reference).  Writes
onnx_image_processing_amd/data/bad_tables.npz with, for P in (256, 512):
    box_P : int8  (P, 5)  rows (x1, x2, y1, y2, r) in the 32x32 patch frame
    thr_P : float32 (P,)  thresholds in 0..255 intensity units
Source of the numbers: reference pytorch_model/descriptor/bad_params.py:4-1568.
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
from pytorch_model.descriptor.bad_params import _get_bad_learned_params  # noqa: E402

out = {}
for p in (256, 512):
    box, thr = _get_bad_learned_params(p)
    box = box.numpy()
    assert np.all(box == np.round(box)) and box.min() >= 0 and box.max() <= 31
    out[f"box_{p}"] = box.astype(np.int8)
    out[f"thr_{p}"] = thr.numpy().astype(np.float32)
dst = os.path.join(os.path.dirname(__file__), "..", "onnx_image_processing_amd", "data", "bad_tables.npz")
np.savez_compressed(dst, **out)
print("wrote", os.path.abspath(dst), {k: v.shape for k, v in out.items()})
