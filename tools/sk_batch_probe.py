#!/usr/bin/env python3
"""Sinkhorn (cost + 20 iterations, duals only) per pair against the batch size: is the row kernel bound by where the dots
come from (L2 / Infinity Cache / HBM) or by its own instruction stream?  (development tool)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from onnx_image_processing_amd import _native as N  # noqa: E402
from onnx_image_processing_amd.pytorch_model.matching.sinkhorn import SinkhornMatcher  # noqa: E402

K = 512
rng = np.random.default_rng(5)
m = SinkhornMatcher(iterations=20, epsilon=0.05)
lib = N.use_debug_library()          # the mi_debug_* hooks live in lib/libmi355x_match_debug.so
for parts in (1, 2, 3, 4):
    lib.mi_debug_set(6, parts)
    for B in (16, 32, 64, 128, 224, 448, 896):
        bits1 = torch.from_numpy(rng.integers(0, 2 ** 31, (B, K, 16), dtype=np.int64).astype(np.int32)).cuda()
        bits2 = torch.from_numpy(rng.integers(0, 2 ** 31, (B, K, 16), dtype=np.int64).astype(np.int32)).cuda()
        m.solve_bits(bits1, bits2, True)
        torch.cuda.synchronize()
        reps = max(5, 2000 // B)
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(reps):
            m.solve_bits(bits1, bits2, True)
        e0.record()
        torch.cuda.synchronize()
        ms = s0.elapsed_time(e0) / reps
        print(f"parts {parts} batch {B:4d}: {ms:.3f} ms = {ms / B * 1e3:.2f} us per pair, dots {B * 0.533:.0f} MB", flush=True)
lib.mi_debug_set(6, 2)
