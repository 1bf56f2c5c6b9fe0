#!/usr/bin/env python3
"""Kernel timeline of ONE pair per call (development tool): run under rocprofv3 --kernel-trace, then
tools/latency_timeline.py prints per-kernel durations and the gaps between consecutive kernels of a call.

    rocprofv3 --kernel-trace -d DIR -o lat -- python3 tools/latency_trace.py [--pairs 1] [--graph]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from onnx_image_processing_amd.graph import GraphedModule  # noqa: E402
from onnx_image_processing_amd.pytorch_model.feature_detection import (MatchExtractionWrapper,  # noqa: E402
                                                                       ShiTomasiSparseBADSinkhornMatcher)
from onnx_image_processing_amd.synth import synth_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=1)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--graph", action="store_true")
ap.add_argument("--single-call", action="store_true")
args = ap.parse_args()
dev = "cuda:0"
cfg = dict(block_size=3, num_pairs=512, binarize=True, soft_binarize=False, sinkhorn_iterations=20, epsilon=0.05,
           unused_score=1.0, nms_radius=5)
model = MatchExtractionWrapper(ShiTomasiSparseBADSinkhornMatcher(max_keypoints=512, **cfg), 100, 0.1).to(dev)
a, b = synth_batch(1000, args.pairs, 480, 640)
i1, i2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
fn = (lambda: model.forward_single_call(i1, i2)) if args.single_call else (lambda: model(i1, i2))
if args.graph:
    g = GraphedModule(type("M", (), {"__call__": staticmethod(lambda x, y: fn())})() if args.single_call else model, i1, i2)
    fn = g.graph.replay
for _ in range(5):
    fn()
torch.cuda.synchronize()
for _ in range(args.iters):
    fn()
    torch.cuda.synchronize()
print("done")
