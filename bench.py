#!/usr/bin/env python3
"""Benchmark of the hot path: image-pairs/sec (640x480, K=512) -- BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs-per-gpu B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the whole path (Shi-Tomasi -> NMS/top-k -> sparse BAD -> cost ->
Sinkhorn -> mutual-NN match extraction; the reference's MatchExtractionWrapper form) over a batch of B synthetic pairs per GPU that is
already resident in HBM, followed (N > 1) by the RCCL gather of the match records to rank 0.
Configuration = BASELINE.json configs[1] hyper-parameters (the export-CLI values, SURVEY.md
§2.2) with K=512; pairs are independent, so ranks hold different pairs (weak scaling).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

H, W, K, NUM_PAIRS = 480, 640, 512, 512
CFG = dict(block_size=3, num_pairs=NUM_PAIRS, binarize=True, soft_binarize=False, sinkhorn_iterations=20,
           epsilon=0.05, unused_score=1.0, distance_type="l2", nms_radius=5, score_threshold=0.0,
           normalize_descriptors=True, sampling_mode="nearest")
MNN = dict(max_matches=100, threshold=0.1)
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def pmc_traffic(kernel_prefix: str, pairs_per_gpu: int):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC passes
    (profiles/*_bench_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this
    script at the recorded pairs per GPU, reads doubled per the gfx950 note).  None when no profile of this batch
    size exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        prof = json.load(open(files[-1]))
        if int(prof.get("pairs_per_gpu", 256)) != pairs_per_gpu:
            return None, None
        kernels = prof["kernels"]
        for name, row in kernels.items():
            if name.startswith(kernel_prefix):
                return row["total_MB"] * 1e6, os.path.relpath(files[-1], ROOT)
    except Exception:
        pass
    return None, None


def cpu_baseline(pairs: int, gpu_records=None) -> dict:
    """The oracle (numpy port of the reference algorithm) on this host's cores, same workload.  gpu_records: the
    (pairs, max_matches, 6) match records the GPU path produced for the same pairs (rank 0's first `pairs` pairs);
    when given, the oracle's outputs are compared with them after the clock stops ("parity")."""
    from oracle import numpy_oracle as O
    from onnx_image_processing_amd.synth import synth_batch
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    t = np.load(os.path.join(ROOT, "onnx_image_processing_amd", "data", "bad_tables.npz"))
    kw = {k: v for k, v in CFG.items() if k not in ("num_pairs", "sampling_mode")}
    a, b = synth_batch(1000, pairs, H, W)
    O.match_pair(a[:1], b[:1], t["box_512"], t["thr_512"], K, **kw)          # warm-up
    results = []
    t0 = time.perf_counter()
    for i in range(pairs):
        k1, k2, p = O.match_pair(a[i:i + 1], b[i:i + 1], t["box_512"], t["thr_512"], K, **kw)
        results.append(O.mnn_extract(p, k1, k2, **MNN))
    dt = time.perf_counter() - t0
    out = {"value": pairs / dt, "unit": "image-pairs/sec", "cores": int(threads), "kind": "port",
           "sample": f"{pairs} pairs 640x480 K=512 (seeds 1000..{999 + pairs}), oracle/numpy_oracle.py, one pair "
                     f"at a time; BLAS matmul uses {threads} threads, the rest is single-threaded numpy"}
    if gpu_records is not None:
        # match-set parity of the very pairs just timed: same matched coordinates, same validity, scores within 1e-4
        # A pair with more than max_matches mutual matches keeps the max_matches best: two scores closer than the 1e-4
        # bound that straddle that cut may legitimately swap ("cut ties"); anything else is a real difference.
        same, cut, other, worst, nmatch = 0, 0, 0, 0.0, 0
        for i, (mk1, mk2, sc, valid, _) in enumerate(results):
            g = gpu_records[i]
            gv = g[:, 5] > 0.5
            want = {(*mk1[0, j], *mk2[0, j]): float(sc[0, j]) for j in np.nonzero(valid[0])[0]}
            got = {tuple(g[j, 0:4]): float(g[j, 4]) for j in np.nonzero(gv)[0]}
            nmatch += len(want)
            worst = max([worst] + [abs(want[k] - got[k]) for k in set(want) & set(got)])
            if set(want) == set(got):
                same += 1
                continue
            full = len(want) == MNN["max_matches"] and len(got) == MNN["max_matches"]
            lo = min(min(want.values()), min(got.values()))
            odd = [k for k in set(want) ^ set(got) if abs({**want, **got}[k] - lo) > 1e-4]
            if full and not odd:
                cut += 1
            else:
                other += 1
        out["parity"] = {"pairs_checked": pairs, "pairs_with_identical_match_set": same,
                         "pairs_differing_only_by_ties_at_the_max_matches_cut": cut, "pairs_differing_otherwise": other,
                         "matches_checked": nmatch, "max_abs_score_diff": worst, "bound": 1e-4}
    return out


def side_workload(args, rank, world, dev) -> None:
    """BASELINE configs[2]/[3] through the same modules: an informational JSON line (value and per-call
    times only), not the bench contract's line."""
    from onnx_image_processing_amd import _native, distributed as D
    from onnx_image_processing_amd.pytorch_model.feature_detection import (AKAZESparseBADSinkhornMatcher,
                                                                           MatchExtractionWrapper,
                                                                           ShiTomasiSparseBADSinkhornMatcher)
    from onnx_image_processing_amd.synth import synth_batch
    B = args.pairs_per_gpu
    if args.workload == "c3":
        h, w, k = 1080, 1920, 1024
        base = ShiTomasiSparseBADSinkhornMatcher(max_keypoints=k, **CFG)
        what = "Shi-Tomasi sparse pipeline, 1920x1080, K=1024 (BASELINE configs[2])"
    else:
        h, w, k = H, W, K
        # export_akaze_sparse_bad_sinkhorn.py defaults: 256 pairs, no binarisation, NMS radius 3
        cfg = dict(num_pairs=256, binarize=False, sinkhorn_iterations=20, epsilon=0.05, unused_score=1.0,
                   distance_type="l2", nms_radius=3, score_threshold=0.0, normalize_descriptors=True,
                   sampling_mode="nearest")
        base = AKAZESparseBADSinkhornMatcher(max_keypoints=k, **cfg)
        what = ("AKAZE(3 scales x 3 steps) + oriented sparse BAD(256, raw) + Sinkhorn(20, eps 0.05), 640x480, K=512 "
                "(BASELINE configs[3], AKAZE export-CLI values)")
    begin, _ = D.shard_range(B * world, rank, world)
    a, b = synth_batch(1000 + begin, B, h, w)
    img1, img2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    del a, b
    model = MatchExtractionWrapper(base, max_matches=MNN["max_matches"], match_threshold=MNN["threshold"]).to(dev)
    model.fuse_extraction = not args.two_step

    def step():
        return D.gather_records(D.pack_records(*model(img1, img2)), dst=0)

    for _ in range(args.warmup):
        out = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    _native.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed_ms = D.barrier_max_ms((time.perf_counter() - t0) * 1e3, dev)
    per_call = _native.timings_ms()
    _native.enable_timing(False)
    if rank == 0:
        ms = elapsed_ms / args.steps
        print(json.dumps({
            "metric": f"image-pairs/sec ({w}x{h}, K={k})", "value": B * world / (ms * 1e-3), "unit": "image-pairs/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": what, "pairs_per_gpu_per_step": B,
                       "mean_valid_matches_per_pair": float(out[..., 5].sum().item()) / (B * world)},
            "kernels": {kk: {"ms_per_step": float(np.sum(v)) / args.steps, "calls_per_step": len(v) / args.steps}
                        for kk, v in per_call.items()}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs-per-gpu", type=int, default=448,
                    help="pairs resident per GPU and processed per step (448: the Sinkhorn row kernel's workgroups of "
                         "each half-batch fill whole rounds of the 256 CUs, and top-k runs two workgroups per CU)")
    ap.add_argument("--cpu-pairs", type=int, default=192, help="oracle sample size for cpu_baseline (0 = skip)")
    ap.add_argument("--single-call", action="store_true",
                    help="run the step as ONE C-ABI call (mi_match_pairs); informational: no per-stage timers, so the "
                         "line carries no roofline object")
    ap.add_argument("--two-step", action="store_true",
                    help="materialise P and run the extractor on it (default: matches straight from the duals)")
    ap.add_argument("--workload", choices=["c2", "c3", "c4"], default="c2",
                    help="c2 (default, the metric's configuration); c3 = 1080x1920 K=1024; c4 = AKAZE front end "
                         "(informational lines: no roofline/cpu_baseline)")
    args = ap.parse_args()

    from onnx_image_processing_amd import _native, distributed as D
    from onnx_image_processing_amd.pytorch_model.feature_detection import (MatchExtractionWrapper,
                                                                           ShiTomasiSparseBADSinkhornMatcher)
    from onnx_image_processing_amd.synth import synth_batch

    rank, world, local = D.init()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if not os.path.exists(_native.LIB_PATH) and local == 0:      # a checkout without the (git-ignored) build product
        from onnx_image_processing_amd.build import build
        build(verbose=False)
    if world > 1:
        dist.barrier()
    _native.load()

    B = args.pairs_per_gpu
    if args.workload != "c2":
        return side_workload(args, rank, world, dev)
    begin, _ = D.shard_range(B * world, rank, world)            # this rank's pairs in the global order
    a, b = synth_batch(1000 + begin, B, H, W)
    img1, img2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)   # resident in HBM before timing
    del a, b
    # the reference's deployment form of "matcher + match extraction" (match_extraction_wrapper.py:82-113)
    model = MatchExtractionWrapper(ShiTomasiSparseBADSinkhornMatcher(max_keypoints=K, **CFG),
                                   max_matches=MNN["max_matches"], match_threshold=MNN["threshold"]).to(dev)
    model.fuse_extraction = not args.two_step

    def step():
        rec = D.pack_records(*(model.forward_single_call(img1, img2) if args.single_call else model(img1, img2)))
        return D.gather_records(rec, dst=0)

    for _ in range(args.warmup):
        out = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events around the roofline kernel's calls only (two per step) inside the timed region; the per-stage
    # table below comes from extra steps after it, so its 24 events per step do not sit in the measurement
    _native.enable_timing(True, only=None if args.single_call else {"mi_corner_response"})
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed_ms = (time.perf_counter() - t0) * 1e3
    per_call = _native.timings_ms()
    _native.enable_timing(False)
    elapsed_ms = D.barrier_max_ms(elapsed_ms, dev)
    stage_steps = 3
    if not args.single_call:                                     # per-stage times (informational), outside the timed region
        _native.enable_timing(True)
        for _ in range(stage_steps):
            out = step()
        stages = _native.timings_ms()
        _native.enable_timing(False)
    else:
        stages = {}

    if rank == 0:
        ms_per_step = elapsed_ms / args.steps
        pairs_per_step = B * world
        kernels = {k: {"ms_per_step": float(np.sum(v)) / stage_steps, "calls_per_step": len(v) / stage_steps}
                   for k, v in stages.items()}
        # K1 corner response: 8 algorithmic bytes per pixel (4 read + 4 written); one launch covers
        # one image of every pair of this rank, two launches per step
        k1_bytes = 8.0 * B * H * W
        if args.single_call:                       # informational line: the step is one C-ABI call, no per-stage events
            print(json.dumps({"metric": "image-pairs/sec (640x480, K=512)", "value": pairs_per_step / (ms_per_step * 1e-3),
                              "unit": "image-pairs/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": "as the default line, issued as one mi_match_pairs call per step",
                                         "pairs_per_gpu_per_step": B}, "kernels": kernels}), flush=True)
            if world > 1:
                dist.barrier()
                dist.destroy_process_group()
            return
        k1_ms = float(np.mean(per_call["mi_corner_response"]))
        achieved = k1_bytes / (k1_ms * 1e-3) / 1e9
        nvalid = float(out[..., 5].sum().item()) / pairs_per_step
        traffic, traffic_src = pmc_traffic("corner_stream_kernel", B)
        line = {
            "metric": "image-pairs/sec (640x480, K=512)",
            "value": pairs_per_step / (ms_per_step * 1e-3),
            "unit": "image-pairs/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "Shi-Tomasi(3) + NMS(r=5)/top-k + sparse BAD(512, hard) + Sinkhorn(20, eps 0.05) "
                                   "+ MNN(100, 0.1), 640x480 gray pairs, K=512 (BASELINE configs[1])",
                       "pairs_per_gpu_per_step": B, "global_pairs_per_step": pairs_per_step,
                       "height": H, "width": W, "max_keypoints": K, "parallelism": f"pair-sharded x{world}",
                       "mean_valid_matches_per_pair": nvalid},
            "roofline": {"kernel": "corner_stream_kernel<3,4> (mi_corner_response)", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "bytes_per_launch": k1_bytes,
                         "ms_per_launch": k1_ms},
            "kernels": kernels,
            # informational: the other bandwidth-type stages by their algorithmic bytes
            "roofline_other": {
                "mi_nms_candidates": {"bytes_per_call": 4.0 * B * H * W, "unit": "GB/s",
                                      "achieved": 4.0 * B * H * W / (float(np.mean(stages["mi_nms_candidates"])) * 1e-3) / 1e9},
                "mi_sinkhorn_dots (per iteration, 2 B/element)": {
                    "bytes_per_call": 2.0 * B * K * K * CFG["sinkhorn_iterations"], "unit": "GB/s",
                    "achieved": 2.0 * B * K * K * CFG["sinkhorn_iterations"]
                    / (float(np.mean(stages["mi_sinkhorn_dots"])) * 1e-3) / 1e9} if "mi_sinkhorn_dots" in stages else None,
            },
        }
        if world == 1 and args.cpu_pairs > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_pairs, out[:args.cpu_pairs].cpu().numpy()
                                                if args.cpu_pairs <= B else None)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
