#!/usr/bin/env python3
"""Soak of the ticket-scheduled K1 (mi_corner_response_balanced): 400 launches per pixel type on 448 frames, every score
map compared with the static schedule's and the counter block checked for zero (development tool; the hand-off of a
ticket between the waves of a workgroup is timing dependent)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from onnx_image_processing_amd import _native as N, ops
from onnx_image_processing_amd.synth import synth_image
n, h, w = 448, 480, 640
base = np.stack([synth_image(1000 + i) for i in range(8)])[:, None]
img8 = torch.from_numpy(np.tile(base, (n // 8, 1, 1, 1))).cuda()
ctr = torch.zeros(ops.TILE_COUNTER_BYTES // 4, dtype=torch.int32, device="cuda")
bad = 0
for x, u8, static in ((img8.float(), 0, "mi_corner_response"), (img8, 1, "mi_corner_response_u8")):
    want = torch.empty((n, 1, h, w), dtype=torch.float32, device="cuda")
    N.call(static, x.data_ptr(), n, h, w, 3, want.data_ptr(), N.stream_ptr())
    got = torch.empty_like(want)
    for it in range(400):
        got.fill_(-1.0)
        N.call("mi_corner_response_balanced", x.data_ptr(), u8, n, h, w, 3, got.data_ptr(), ctr.data_ptr(), N.stream_ptr())
        if not torch.equal(got, want) or int(ctr.abs().sum()) != 0:
            bad += 1
    print("u8" if u8 else "f32", "400 launches of 448 images, mismatches:", bad, flush=True)
