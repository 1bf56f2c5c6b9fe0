#!/usr/bin/env python3
"""mi_akaze_scale on the c4 bench's batch (128 images 640x480, 3 steps, NMS 5): time per launch (HIP events) and
bit-equality with the per-step kernels (development tool).  usage: akaze_bench.py [images] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from onnx_image_processing_amd import ops  # noqa: E402
from onnx_image_processing_amd.synth import synth_image  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
base = np.stack([synth_image(1000 + i) for i in range(8)])[:, None].astype(np.float32)
img = torch.from_numpy(np.tile(base, (n // 8, 1, 1, 1))).cuda()
want_l = ops.akaze_diffuse(img, 3, 0.05, 0.25)
want_s = ops.akaze_hessian_scores(want_l, 0.001, 5)
got_l, got_s = ops.akaze_scale(img, 3, 0.05, 0.25, 0.001, 5)
print("fused == per-step kernels:", bool(torch.equal(got_l, want_l)), bool(torch.equal(got_s, want_s)), flush=True)
for scale in (1, 2):       # later scales see smoother images: same time expected
    img2 = got_l.clone()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for _ in range(3):
        ops.akaze_scale(img2, 3, 0.05, 0.25, 0.001, 5)
    ev[0].record()
    for i in range(reps):
        ops.akaze_scale(img2, 3, 0.05, 0.25, 0.001, 5)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in zip(ev, ev[1:]))
    px = n * 480 * 640
    print(f"scale {scale}: mi_akaze_scale {n} images: median {ts[len(ts) // 2] * 1e3:.1f} us, min {ts[0] * 1e3:.1f} us "
          f"= {12.0 * px / (ts[len(ts) // 2] * 1e-3) / 1e12:.2f} TB/s at 12 B/px", flush=True)
    got_l, _ = ops.akaze_scale(img2, 3, 0.05, 0.25, 0.001, 5)
