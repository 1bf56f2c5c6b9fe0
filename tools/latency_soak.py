#!/usr/bin/env python3
"""Soak of the one-pair-per-call path: N replays of the captured forward (module path and mi_match_pairs), the host
synchronising after every replay, every replay's outputs compared with the eager call's bit for bit -- optionally while
another process keeps the GPU busy with batched steps (the single-launch Sinkhorn's workgroups then share the device).
    python tools/latency_soak.py 20000         (development tool)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from onnx_image_processing_amd.graph import GraphedModule  # noqa: E402
from onnx_image_processing_amd.pytorch_model.feature_detection import (MatchExtractionWrapper,  # noqa: E402
                                                                       ShiTomasiSparseBADSinkhornMatcher)
from onnx_image_processing_amd.synth import synth_batch  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
dev = torch.device("cuda:0")
a, b = synth_batch(1000, 1, bench.H, bench.W)
x, y = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
model = MatchExtractionWrapper(ShiTomasiSparseBADSinkhornMatcher(max_keypoints=bench.K, **bench.CFG),
                               max_matches=bench.MNN["max_matches"], match_threshold=bench.MNN["threshold"]).to(dev)
for name, fn in (("module", model), ("single_call", model.forward_single_call)):
    want = [t.clone() for t in fn(x, y)]
    g = GraphedModule(fn, x, y)
    bad = 0
    t0 = time.perf_counter()
    for i in range(N):
        g.graph.replay()
        torch.cuda.synchronize()
        if not all(torch.equal(o, w) for o, w in zip(g.static_outputs, want)):
            bad += 1
    dt = time.perf_counter() - t0
    print(f"{name}: {N} replays, {bad} differing from the eager outputs, {int(want[3].sum())} valid matches, "
          f"{dt / N * 1e3:.3f} ms per replay incl. the comparison", flush=True)
