"""HIP-graph replay of a matcher for latency-bound use (one pair per call, the VO loop pattern).

Every C-ABI entry point only enqueues kernels on the current stream -- no allocation, no host
synchronisation -- and the Python layer allocates through torch's caching allocator, so a whole
`forward()` (about 60 launches) can be captured once into a hipGraph and replayed with one launch
call.  At one 640x480 pair per call the eager path is bound by launch overhead; the replay is bound
by the kernels.  Shapes are fixed at capture time; inputs are copied into the captured buffers.
"""
from __future__ import annotations

import torch


class GraphedModule:
    """graphed = GraphedModule(model, image1, image2); out = graphed(image1, image2)

    `model` is any module of this package whose forward takes device tensors and returns a tensor or
    a tuple of tensors.  The returned tensors are the graph's own output buffers: they are
    overwritten by the next call (clone them to keep them)."""

    def __init__(self, model, *example_inputs: torch.Tensor, warmup: int = 2, before_capture=None, inside_capture=None,
                 debug: bool = False):
        """before_capture(): called on the capture stream after the warm-up calls and before the capture begins (e.g.
        ops.set_sinkhorn_schedule to pin the stream schedule the capture records); inside_capture(): called inside the
        capture after the model's forward (tests); debug: keep the captured hipGraph_t (graph_topology())."""
        if not example_inputs or not all(t.is_cuda for t in example_inputs):
            raise RuntimeError("GraphedModule needs example inputs on the GPU")
        self.model = model
        self.static_inputs = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # lazy state (BAD plan, ...) is built outside the capture
            for _ in range(max(1, warmup)):
                model(*self.static_inputs)
            if before_capture is not None:
                before_capture()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # debug: the hipGraph_t stays alive after the capture (graph_topology below reads its nodes and edges)
        self.graph = torch.cuda.CUDAGraph(keep_graph=True) if debug else torch.cuda.CUDAGraph()
        # capture on the stream the warm-up ran on: per-stream helper resources of the C ABI (the fork/join
        # streams of mi_sinkhorn_dots for >= 64 pairs) then already exist and nothing is created mid-capture
        with torch.cuda.graph(self.graph, stream=side):
            self.static_outputs = model(*self.static_inputs)
            if inside_capture is not None:
                inside_capture()

    def __call__(self, *inputs: torch.Tensor):
        if len(inputs) != len(self.static_inputs):
            raise RuntimeError(f"expected {len(self.static_inputs)} inputs, got {len(inputs)}")
        for dst, src in zip(self.static_inputs, inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise RuntimeError(f"input shape/dtype {tuple(src.shape)}/{src.dtype} differs from the captured "
                                   f"{tuple(dst.shape)}/{dst.dtype}")
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.static_outputs


def graph_topology(graph: "torch.cuda.CUDAGraph") -> dict:
    """Nodes and dependency edges of a captured graph (a CUDAGraph made with keep_graph=True), read through
    hipGraphGetNodes / hipGraphGetEdges / hipGraphNodeGetType: {"nodes", "edges", "forks" (nodes with more than one
    successor: a cross-stream fork inside the capture), "joins" (more than one predecessor), "types" {type id: count}}.
    hipGraphDebugDotPrint writes no file on this stack (ROCm 7.2), hence the direct query."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    g = ctypes.c_void_p(graph.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    if hip.hipGraphGetNodes(g, None, ctypes.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    nodes = (ctypes.c_void_p * max(1, n.value))()
    hip.hipGraphGetNodes(g, nodes, ctypes.byref(n))
    e = ctypes.c_size_t(0)
    if hip.hipGraphGetEdges(g, None, None, ctypes.byref(e)) != 0:
        raise RuntimeError("hipGraphGetEdges failed")
    src = (ctypes.c_void_p * max(1, e.value))()
    dst = (ctypes.c_void_p * max(1, e.value))()
    hip.hipGraphGetEdges(g, src, dst, ctypes.byref(e))
    succ, pred, types = {}, {}, {}
    for i in range(e.value):
        succ.setdefault(src[i], set()).add(dst[i])
        pred.setdefault(dst[i], set()).add(src[i])
    for i in range(n.value):
        t = ctypes.c_int(-1)
        hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t))
        types[t.value] = types.get(t.value, 0) + 1
    return {"nodes": n.value, "edges": e.value, "forks": sum(1 for v in succ.values() if len(v) > 1),
            "joins": sum(1 for v in pred.values() if len(v) > 1), "types": types}
