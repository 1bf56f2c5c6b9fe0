#!/usr/bin/env python3
"""Phase times of the top-k kernel for one 640x480 image (development tool)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from onnx_image_processing_amd import _native as N, ops  # noqa: E402
from onnx_image_processing_amd.synth import synth_batch  # noqa: E402

lib = N.use_debug_library()          # the mi_debug_* hooks live in lib/libmi355x_match_debug.so
a, _ = synth_batch(1000, 2, 480, 640)
img = torch.from_numpy(a).cuda()
score = ops.corner_response(img, 3).squeeze(1)
buf = torch.zeros(8, dtype=torch.int64, device="cuda")
lib.mi_debug_topk_stamps(buf.data_ptr())
for _ in range(5):
    kp, sc = ops.nms_topk(score, 5, 512, 0.0, 7)
    torch.cuda.synchronize()
lib.mi_debug_topk_stamps(None)
t = buf.cpu().numpy().astype(np.int64)
names = ["count+slots", "gather", "radix select", "compaction", "sort", "epilogue"]
for nm, d in zip(names, np.diff(t[:7]) * 0.01):
    print(f"{nm:14s} {d:6.2f} us")
print("total", (t[6] - t[0]) * 0.01)
