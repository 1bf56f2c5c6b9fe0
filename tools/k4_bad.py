#!/usr/bin/env python3
"""K4 (fast BAD kernel) on the bench workload: time per launch, fp32 and uint8 frames (development tool)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from onnx_image_processing_amd import _native as N, ops  # noqa: E402
from onnx_image_processing_amd.pytorch_model.descriptor.bad import SparseBAD  # noqa: E402
from onnx_image_processing_amd.synth import synth_image  # noqa: E402

n, k = int(sys.argv[1]) if len(sys.argv) > 1 else 448, 512
base = np.stack([synth_image(1000 + i) for i in range(8)])[:, None]
img8 = torch.from_numpy(np.tile(base, (n // 8, 1, 1, 1))).cuda()
img32 = img8.float()
kp, _ = ops.nms_topk(ops.corner_response(img32, 3)[:, 0], 5, k, 0.0, 7)
for pairs in (512, 256):
    bad = SparseBAD(num_pairs=pairs, binarize=True, soft_binarize=False).cuda()
    geom = np.ascontiguousarray(bad.pair_geom.cpu().numpy().astype(np.uint32))
    a, b = ctypes.c_int(0), ctypes.c_int(0)
    N.use_debug_library().mi_debug_bad_plan_passes(geom.ctypes.data, pairs, ctypes.byref(a), ctypes.byref(b))
    for name, x in (("f32", img32), ("u8", img8)):
        ref = bad.forward_bits(x, kp)
        torch.cuda.synchronize()
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(20):
            bad.forward_bits(x, kp)
        e0.record()
        torch.cuda.synchronize()
        print(f"pairs {pairs} {name}: {s0.elapsed_time(e0) / 20 * 1e3:.1f} us per {n} x {k} keypoints; gather passes {a.value} -> {b.value}")
