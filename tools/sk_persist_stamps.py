#!/usr/bin/env python3
"""Phase times inside the single-launch Sinkhorn kernel (development tool; mi_debug_set key 8).
Stamps per iteration: 0 loop top, 1 row pass + in-workgroup reduction done, 2 granules gathered + column update done,
3 derived state done.  100 MHz clock."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from onnx_image_processing_amd import _native as N  # noqa: E402

lib = N.use_debug_library()          # the mi_debug_* hooks live in lib/libmi355x_match_debug.so
lib.mi_debug_set(8, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = m = 512
rng = np.random.default_rng(0)
b1 = torch.from_numpy(rng.integers(0, 2 ** 31, size=(batch, n, 16)).astype(np.int32)).cuda()
b2 = b1.clone()
pitch = 512
dots = torch.empty((batch, n, pitch), dtype=torch.int16, device="cuda")
ri = torch.empty((batch, n, 2), device="cuda")
ci = torch.empty((batch, m, 2), device="cuda")
N.call("mi_cost_dots_bits", b1.data_ptr(), b2.data_ptr(), batch, n, m, 512, 1, dots.data_ptr(), pitch, ri.data_ptr(), ci.data_ptr(), N.stream_ptr())
wbytes = int(lib.mi_sinkhorn_dots_workspace_bytes(batch, n, m))
work = torch.zeros((wbytes // 8,), dtype=torch.int64, device="cuda")
u = torch.empty((batch, n + 1), device="cuda")
v = torch.empty((batch, m + 1), device="cuda")
for _ in range(5):
    N.call("mi_sinkhorn_dots", dots.data_ptr(), ri.data_ptr(), ci.data_ptr(), batch, n, m, pitch, 0.05, 1.0, 1.0, 20, u.data_ptr(),
           v.data_ptr(), None, work.data_ptr(), wbytes, 0, N.stream_ptr())
    torch.cuda.synchronize()
w = work.cpu().numpy()
prof = w[-(4096 // 8):][:160].reshape(20, 8).astype(np.int64)
fail = int(w[-(4096 // 8) - 2]) & 0xFFFFFFFF
print("fail word", fail)
t = prof[:, [0, 1, 4, 5, 2, 3]]
d = np.diff(t, axis=1) * 0.01
print("per iteration us: row pass | publish | gather (col 0) | column math + barrier | derive ; polls")
for i in range(20):
    print(np.round(d[i], 2))
print("total us", (prof[-1, 2] - prof[0, 0]) * 0.01, "mean", np.round(d[1:19].mean(0), 2))
