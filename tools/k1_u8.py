#!/usr/bin/env python3
"""K1 on uint8 frames: time per launch and fraction of the HBM roofline at 5 B/px (development tool)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from onnx_image_processing_amd import _native as N, ops  # noqa: E402
from onnx_image_processing_amd.synth import synth_image  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 448
base = np.stack([synth_image(1000 + i) for i in range(8)])[:, None]
img8 = torch.from_numpy(np.tile(base, (n // 8, 1, 1, 1))).cuda()
img32 = img8.float()
lib = N.use_debug_library()          # the mi_debug_* hooks live in lib/libmi355x_match_debug.so
for rows in (4, 5, 8):
    lib.mi_debug_set(2, rows)
    for name, x, bpp in (("u8", img8, 5.0), ("f32", img32, 8.0)):
        ref = ops.corner_response(x, 3)
        assert torch.equal(ref, ops.corner_response(img32, 3))
        torch.cuda.synchronize()
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for i in range(40):
            ops.corner_response(x, 3)
        e0.record()
        torch.cuda.synchronize()
        ms = s0.elapsed_time(e0) / 40
        print(f"rows {rows} {name}: {ms * 1e3:.1f} us per {n} images, {bpp * n * 480 * 640 / ms / 1e6:.0f} GB/s = {bpp * n * 480 * 640 / ms / 1e6 / 8000:.3f} of 8 TB/s")
lib.mi_debug_set(2, 4)
# per-workgroup lifetimes of the persistent grid: are all workgroups resident together?
buf = torch.zeros(4 * 2048, dtype=torch.int64, device='cuda')
lib.mi_debug_clock_probe(buf.data_ptr())
for name, x in (("f32", img32), ("u8", img8)):
    for _ in range(3):
        buf.zero_()
        ops.corner_response(x, 3)
    torch.cuda.synchronize()
    c = buf.cpu().numpy().astype(np.int64).reshape(-1, 4)
    c = c[c[:, 1] > 0]
    t0 = c[:, 1].min()
    start, end = (c[:, 1] - t0) * 0.01, (c[:, 3] - t0) * 0.01
    mhz = ((c[:, 2] - c[:, 0]) / np.maximum(1, (c[:, 3] - c[:, 1]) * 0.01)).mean()
    life = end - start
    bx = np.arange(len(c))
    print("  lifetime by blockIdx % 8 (XCD):", np.array([life[bx % 8 == k].mean() for k in range(8)]).round(0))
    print("  lifetime by (blockIdx // 8) % 32 :", np.array([life[(bx // 8) % 32 == k].mean() for k in range(32)]).round(0))
    print("  lifetime by blockIdx quarter:", np.array([life[bx * 4 // len(c) == k].mean() for k in range(4)]).round(0))
    print(f"{name}: {len(c)} workgroups, kernel {end.max():.1f} us, clock {mhz:.0f} MHz; starts: {np.percentile(start, [0, 50, 75, 90, 100]).round(1)} "
          f"ends: {np.percentile(end, [0, 25, 50, 75, 100]).round(1)} lifetimes: {np.percentile(end - start, [0, 50, 100]).round(1)}")
lib.mi_debug_clock_probe(None)
