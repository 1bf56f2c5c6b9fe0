#!/usr/bin/env python3
"""Which calls made by ANOTHER host thread invalidate a stream capture that is open in global capture mode (torch's
default)?  Thread A opens a capture and holds it; thread B performs ONE candidate operation on its own stream; A closes
the capture and reports whether it survived.  (development tool; one line per candidate)"""
import ctypes
import os
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnx_image_processing_amd import ops

hip = ctypes.CDLL("libamdhip64.so")
dev = "cuda"
x = torch.zeros(1 << 16, device=dev)
sb = torch.cuda.Stream()
sa = torch.cuda.Stream()
rng = np.random.default_rng(1)
b1 = torch.from_numpy(rng.integers(0, 2 ** 31, size=(64, 96, 8)).astype(np.int32)).cuda()
b2 = torch.from_numpy(rng.integers(0, 2 ** 31, size=(64, 96, 8)).astype(np.int32)).cuda()
ev = ctypes.c_void_p()
hip.hipEventCreate(ctypes.byref(ev))
ev2 = ctypes.c_void_p()
hip.hipEventCreate(ctypes.byref(ev2))
with torch.cuda.stream(sb):
    for _ in range(3):
        ops.sinkhorn_bits(b1, b2, True, 0.05, 1.0, 10, return_duals=True)
    y = x + 1
    hip.hipEventRecord(ev, ctypes.c_void_p(sb.cuda_stream))
    hip.hipEventRecord(ev2, ctypes.c_void_p(sb.cuda_stream))
torch.cuda.synchronize()


def relaxed(fn):
    def run():
        mode = ctypes.c_int(2)                       # hipStreamCaptureModeRelaxed
        hip.hipThreadExchangeStreamCaptureMode(ctypes.byref(mode))
        try:
            return fn()
        finally:
            hip.hipThreadExchangeStreamCaptureMode(ctypes.byref(mode))
    return run


# preallocated buffers: torch.cuda.graph empties the allocator's cache on entry, and a fresh hipMalloc by torch while
# another thread captures in global mode is refused (that would be torch's violation, not the library's)
from onnx_image_processing_amd import _native as N
with torch.cuda.stream(sb):
    _, u0, v0, (dots, ri, ci, pitch, (work, _)) = ops.sinkhorn_bits(b1, b2, True, 0.05, 1.0, 10, want_p=False, return_state=True)
torch.cuda.synchronize()


def lib_call(pin=None, flags=0):
    def run():
        with torch.cuda.stream(sb):
            if pin is not None:
                ops.set_sinkhorn_schedule(pin)
            for _ in range(12):
                N.call("mi_sinkhorn_dots", dots.data_ptr(), ri.data_ptr(), ci.data_ptr(), 64, 96, 96, pitch, 0.05, 1.0, 1.0,
                       10, u0.data_ptr(), v0.data_ptr(), None, work.data_ptr(), work.numel() * 8, flags, N.stream_ptr())
            return ops.sinkhorn_schedule(64, 96, 96, 10)
    return run


def torch_kernel():
    with torch.cuda.stream(sb):
        x.add_(1.0)


def torch_alloc():
    with torch.cuda.stream(sb):
        return torch.empty(1 << 10, device=dev)


cands = {
    "nothing": lambda: None,
    "torch kernel on another stream": torch_kernel,
    "torch.empty (pooled)": torch_alloc,
    "hipStreamIsCapturing": lambda: hip.hipStreamIsCapturing(ctypes.c_void_p(sb.cuda_stream), ctypes.byref(ctypes.c_int())),
    "hipEventRecord": lambda: hip.hipEventRecord(ev, ctypes.c_void_p(sb.cuda_stream)),
    "hipEventRecord relaxed": relaxed(lambda: hip.hipEventRecord(ev, ctypes.c_void_p(sb.cuda_stream))),
    "hipEventQuery": lambda: hip.hipEventQuery(ev),
    "hipEventQuery relaxed": relaxed(lambda: hip.hipEventQuery(ev)),
    "hipEventElapsedTime relaxed": relaxed(lambda: hip.hipEventElapsedTime(ctypes.byref(ctypes.c_float()), ev, ev2)),
    "hipStreamWaitEvent": lambda: hip.hipStreamWaitEvent(ctypes.c_void_p(sb.cuda_stream), ev, 0),
    "hipGetLastError": lambda: hip.hipGetLastError(),
    "hipGetDevice": lambda: hip.hipGetDevice(ctypes.byref(ctypes.c_int())),
    "lib: 64 pairs, NO_FORK": lib_call(flags=2),
    "lib: 64 pairs, pinned unsplit": lib_call(pin=2),
    "lib: 64 pairs, pinned fork": lib_call(pin=0),
    "lib: 64 pairs, tuning": lib_call(pin=-1),
    "hipStreamSynchronize": lambda: hip.hipStreamSynchronize(ctypes.c_void_p(sb.cuda_stream)),
}
only = sys.argv[1:]
for name, fn in cands.items():
    if only and name not in only:
        continue
    torch.cuda.synchronize()
    rc = {}
    opened, done = threading.Event(), threading.Event()

    def b():
        opened.wait(30)
        try:
            rc["ret"] = fn()
        except Exception as e:      # noqa: BLE001
            rc["exc"] = repr(e)[:120]
        done.set()

    t = threading.Thread(target=b)
    t.start()
    g = torch.cuda.CUDAGraph()
    ok = True
    try:
        with torch.cuda.graph(g, stream=sa):
            z = x * 2.0
            opened.set()
            done.wait(60)
    except Exception as e:      # noqa: BLE001
        ok = False
        rc["capture"] = repr(e)[:100]
    opened.set()
    t.join(60)
    torch.cuda.synchronize()
    print(f"{'SURVIVED' if ok else 'INVALIDATED':12s} {name:34s} {rc}", flush=True)
