#!/usr/bin/env python3
"""Detection + description of 448 frames in one go vs in chunks small enough to stay in the 256 MB Infinity Cache
(development tool): K1 -> K2 -> top-k -> K4 per chunk."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from onnx_image_processing_amd import ops  # noqa: E402
from onnx_image_processing_amd.pytorch_model.descriptor.bad import SparseBAD  # noqa: E402
from onnx_image_processing_amd.synth import synth_image  # noqa: E402

n, k = 448, 512
base = np.stack([synth_image(1000 + i) for i in range(8)])[:, None]
img8 = torch.from_numpy(np.tile(base, (n // 8, 1, 1, 1))).cuda()
bad = SparseBAD(num_pairs=512, binarize=True, soft_binarize=False).cuda()
for name, x in (("f32", img8.float()), ("u8", img8)):
    for chunk in (448, 224, 112, 64, 32, 448):
        def run():
            for c0 in range(0, n, chunk):
                xc = x[c0:c0 + chunk]
                kp, _ = ops.nms_topk(ops.corner_response(xc, 3)[:, 0], 5, k, 0.0, 7)
                bad.forward_bits(xc, kp)
        run()
        torch.cuda.synchronize()
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(10):
            run()
        e0.record()
        torch.cuda.synchronize()
        print(f"{name} chunk {chunk:3d}: {s0.elapsed_time(e0) / 10 * 1e3:.0f} us per {n} frames", flush=True)
