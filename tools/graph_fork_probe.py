#!/usr/bin/env python3
"""What does a hipGraph captured across a stream fork look like, and does replaying it survive a device with ONE
hardware queue (GPU_MAX_HW_QUEUES=1)?  Round 3 recorded a host segfault in hipGraphLaunch for exactly that case
(gpurun_out/hwq1.log); this tool separates the runtime from this library:

  stage "torch":  a fork/join capture made of torch ops only (kernel; side stream waits; kernel on each; join; kernel)
  stage "lib":    mi_sinkhorn_dots for 64 pairs captured with the fork pinned (schedule 0), with two helpers
                  (schedule 1) and unsplit (schedule 2); node / edge / fork counts of each capture are printed
                  (hipGraphGetNodes / hipGraphGetEdges: hipGraphDebugDotPrint writes no file on this stack)

Each stage prints a line before and after its first replay, so the last line printed names the replay that died.
    GPU_MAX_HW_QUEUES=1 python tools/graph_fork_probe.py torch|lib0|lib1|lib2 [outdir]      (development tool)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
stage = sys.argv[1]
outdir = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/graph_fork"
os.makedirs(outdir, exist_ok=True)
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), "stage", stage, flush=True)


def dump(graph, name):
    from onnx_image_processing_amd.graph import graph_topology
    print(f"{name}: {graph_topology(graph)}", flush=True)


if stage == "torch":
    x = torch.zeros(1 << 20, device="cuda")
    y = torch.zeros(1 << 20, device="cuda")
    main, helper = torch.cuda.Stream(), torch.cuda.Stream()
    main.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g, stream=main):
        x.add_(1.0)
        helper.wait_stream(main)                    # fork
        with torch.cuda.stream(helper):
            y.add_(2.0)
        x.mul_(3.0)
        main.wait_stream(helper)                    # join
        z = x + y
    dump(g, "torch_fork")
    for i in range(3):
        print("replay", i, "...", flush=True)
        g.replay()
        torch.cuda.synchronize()
        print("replay", i, "ok", float(z[0]), flush=True)
else:
    sched = int(stage[3:])
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from onnx_image_processing_amd import ops
    rng = np.random.default_rng(1)
    b1 = torch.from_numpy(rng.integers(0, 2 ** 31, size=(64, 96, 8)).astype(np.int32)).cuda()
    b2 = torch.from_numpy(rng.integers(0, 2 ** 31, size=(64, 96, 8)).astype(np.int32)).cuda()
    run = lambda: ops.sinkhorn_bits(b1, b2, True, 0.05, 1.0, 10, return_duals=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eager = [t.clone() for t in run()]
        ops.set_sinkhorn_schedule(sched)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g, stream=side):
        out = run()
    dump(g, f"lib_schedule{sched}")
    for i in range(3):
        print("replay", i, "...", flush=True)
        g.replay()
        torch.cuda.synchronize()
        print("replay", i, "ok", all(torch.equal(a, b) for a, b in zip(out, eager)), flush=True)
