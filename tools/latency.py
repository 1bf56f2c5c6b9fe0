#!/usr/bin/env python3
"""Single-pair latency (the VO loop pattern): eager launches vs hipGraph replay (development tool).

    python tools/latency.py [--pairs 1] [--iters 200]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from onnx_image_processing_amd.graph import GraphedModule  # noqa: E402
from onnx_image_processing_amd.pytorch_model.feature_detection import (MatchExtractionWrapper,  # noqa: E402
                                                                       ShiTomasiSparseBADSinkhornMatcher)
from onnx_image_processing_amd.synth import synth_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    dev = "cuda:0"
    cfg = dict(block_size=3, num_pairs=512, binarize=True, soft_binarize=False, sinkhorn_iterations=20, epsilon=0.05,
               unused_score=1.0, nms_radius=5)
    model = MatchExtractionWrapper(ShiTomasiSparseBADSinkhornMatcher(max_keypoints=512, **cfg), 100, 0.1).to(dev)
    a, b = synth_batch(1000, args.pairs, 480, 640)
    i1, i2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)

    def timed(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.iters * 1e3, out

    eager_ms, eager_out = timed(lambda: model(i1, i2))
    graphed = GraphedModule(model, i1, i2)
    graph_ms, graph_out = timed(lambda: graphed(i1, i2))
    same = all(torch.equal(x, y) for x, y in zip(eager_out, graph_out))
    print(f"module path, pairs per call {args.pairs}: eager {eager_ms:.3f} ms, hipGraph replay {graph_ms:.3f} ms "
          f"({args.pairs / graph_ms * 1e3:.0f} pairs/s), outputs identical: {same}")

    class OneCall(torch.nn.Module):                       # the same forward as ONE C-ABI call (mi_match_pairs)
        def forward(self, x, y):
            return model.forward_single_call(x, y)
    one = OneCall()
    e1, o1 = timed(lambda: one(i1, i2))
    g1 = GraphedModule(one, i1, i2)
    e2, o2 = timed(lambda: g1(i1, i2))

    def synced(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.iters * 1e3
    s1, s2 = synced(lambda: one(i1, i2)), synced(g1.graph.replay)
    same = all(torch.equal(x, y) for x, y in zip(o1, eager_out)) and all(torch.equal(x, y) for x, y in zip(o2, eager_out))
    print(f"mi_match_pairs, pairs per call {args.pairs}: eager {e1:.3f} ms, hipGraph replay {e2:.3f} ms back to back; "
          f"{s1:.3f} / {s2:.3f} ms with a host sync per call; outputs identical to the module path: {same}")


if __name__ == "__main__":
    main()
