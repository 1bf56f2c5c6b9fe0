#!/usr/bin/env python3
"""A bare hipMemsetAsync captured into a hipGraph (torch.cuda.graph on ROCm 7.2 / torch 2.10), none of this repo's code
involved: which words of the range are zero after each replay when the host fills the buffer and synchronises the device
between replays?  Observed on MI355X: replay 0 zeroes the range; every later replay fills it with a 16-byte pattern that
is the argument block of the host's last fill_ kernel (element count, fill value, pointer).  That is why the library
clears its hand-off area and ticket counters with a kernel (csrc/common.h, mi_zero_async).
    python tools/graph_memset_probe.py 16656        (development tool)"""
import os
import sys

import numpy as np
import torch

size = int(sys.argv[1])
buf = torch.full((size // 4 + 64,), 0x55555555, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
torch.cuda.synchronize()
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=side):
    rc = hip.hipMemsetAsync(buf.data_ptr(), 0, size, torch.cuda.current_stream().cuda_stream)
    out = buf + 0          # a kernel node after the memset node
for it in range(4):
    buf.fill_(0x55555555)
    torch.cuda.synchronize()
    gr.replay()
    torch.cuda.synchronize()
    w = buf[: size // 4].cpu().numpy()
    nz = np.nonzero(w)[0]
    print("size", size, "replay", it, "rc", rc, "nonzero words in the memset range:", len(nz), (int(nz[0]), int(nz[-1]), [hex(int(x) & 0xffffffff) for x in w[nz[:4]]]) if len(nz) else "")
