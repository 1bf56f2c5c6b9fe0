"""One single-thread worker of bench.py's `cpu_baseline.host_parallel` leg -- TEST / MEASUREMENT INFRASTRUCTURE, not
product code (see oracle/torch_cpu.py for what is timed and how it is pinned to the reference).

Image pairs are independent, so what a host can do with the reference's CPU path is run one pair per core:
bench.py starts W of these processes (fresh CPU-only children, started by the bench parent as child processes -- no
exec of a process that has touched the GPU), each pinned to one torch thread.  Protocol: build the path, one warm-up
pair, print "ready", wait for "go" on stdin (so that all workers time the same wall-clock window), time `pairs` pairs
one per call (the reference harness's call pattern, sample/image_matching.py:313-328), print one JSON line.

    python oracle/cpu_worker.py <seed> <pairs> <height> <width> <max_keypoints>
"""
import json
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main() -> None:
    seed, pairs, h, w, k = (int(x) for x in sys.argv[1:6])
    torch.set_num_threads(1)
    from onnx_image_processing_amd.synth import synth_batch
    from oracle.torch_cpu import TorchCpuPath
    t = np.load(os.path.join(ROOT, "onnx_image_processing_amd", "data", "bad_tables.npz"))
    path = TorchCpuPath(t["box_512"], t["thr_512"], k, block_size=3, binarize=True, soft_binarize=False,
                        sinkhorn_iterations=20, epsilon=0.05, unused_score=1.0, nms_radius=5, score_threshold=0.0,
                        normalize_descriptors=True)
    a, b = synth_batch(seed, pairs, h, w)
    ta, tb = torch.from_numpy(a), torch.from_numpy(b)
    path.match(ta[:1], tb[:1], max_matches=100, threshold=0.1)          # warm-up
    print("ready", flush=True)
    if sys.stdin.readline().strip() != "go":
        raise SystemExit(2)
    t0 = time.perf_counter()
    nvalid = 0
    for i in range(pairs):
        out = path.match(ta[i:i + 1], tb[i:i + 1], max_matches=100, threshold=0.1)
        nvalid += int(np.asarray(out[3]).sum())
    print(json.dumps({"pairs": pairs, "seconds": time.perf_counter() - t0, "valid_matches": nvalid}), flush=True)


if __name__ == "__main__":
    main()
