#!/usr/bin/env python3
"""Does the C2 step gain from running two half-batches on two HIP streams (one half's HBM-bound Sinkhorn under the
other half's LDS- / latency-bound front end)?  One stream x B pairs against two streams x B/2 pairs, same total work.
(development tool)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from onnx_image_processing_amd.pytorch_model.feature_detection import (MatchExtractionWrapper,  # noqa: E402
                                                                       ShiTomasiSparseBADSinkhornMatcher)
from onnx_image_processing_amd.synth import synth_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 448
STEPS = 200
dev = torch.device("cuda:0")
a, b = synth_batch(1000, B, bench.H, bench.W)
img1, img2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
model = MatchExtractionWrapper(ShiTomasiSparseBADSinkhornMatcher(max_keypoints=bench.K, **bench.CFG),
                               max_matches=bench.MNN["max_matches"], match_threshold=bench.MNN["threshold"]).to(dev)


def timed(fn, steps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


one = timed(lambda: model(img1, img2), STEPS)
print(f"one stream, {B} pairs per step: {one:.3f} ms = {B / one:.1f} k pairs/s", flush=True)
for parts in (2, 4):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    cut = [B * i // parts for i in range(parts + 1)]

    def step():
        for s, lo, hi in zip(streams, cut[:-1], cut[1:]):
            with torch.cuda.stream(s):
                model(img1[lo:hi], img2[lo:hi])
    ms = timed(step, STEPS)
    print(f"{parts} streams x {B // parts} pairs: {ms:.3f} ms = {B / ms:.1f} k pairs/s", flush=True)
