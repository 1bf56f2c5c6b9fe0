"""Same-box A/B of a debug key (tools/scratch/ll_ab.py KEY V1,V2,...): ms per mi_cost_dots_bits + mi_sinkhorn_dots call
(20 iterations) for a few shapes, every value measured twice, alternating."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from onnx_image_processing_amd import _native as N, ops

def bench(batch, n, m, lib, key, reps=30):
    rng = np.random.default_rng(1)
    b1 = torch.from_numpy(rng.integers(0, 2 ** 32, size=(batch, n, 16), dtype=np.uint64).astype(np.uint32).view(np.int32)).cuda()
    b2 = torch.from_numpy(rng.integers(0, 2 ** 32, size=(batch, m, 16), dtype=np.uint64).astype(np.uint32).view(np.int32)).cuda()
    out = {}
    for val in VALUES + VALUES:
        assert lib.mi_debug_set(key, val) == 0
        for _ in range(12):
            ops.sinkhorn_bits(b1, b2, True, 0.05, 1.0, 20, return_duals=True, want_p=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.sinkhorn_bits(b1, b2, True, 0.05, 1.0, 20, return_duals=True, want_p=False)
        torch.cuda.synchronize()
        out.setdefault(val, []).append((time.perf_counter() - t0) / reps * 1e3)
    return out

with N.debug_library() as lib:
    key = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    VALUES = tuple(int(x) for x in sys.argv[2].split(',')) if len(sys.argv) > 2 else (1, 0)
    for shape in ((128, 512, 512), (448, 512, 512), (128, 1024, 1024), (64, 512, 512)):
        print(shape, {k: [round(x, 4) for x in v] for k, v in bench(*shape, lib, key).items()}, flush=True)
