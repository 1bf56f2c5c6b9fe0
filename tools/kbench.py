#!/usr/bin/env python3
"""Per-kernel micro-benchmark through the C ABI (development tool, GPU box only).

    python tools/kbench.py [corner nms topk bad cost sinkhorn mnn] [--images 256] [--iters 20]

Prints average milliseconds per call (HIP events on the launch stream) and the achieved
algorithmic GB/s where a byte count is defined.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from onnx_image_processing_amd import _native as N, ops  # noqa: E402
from onnx_image_processing_amd.pytorch_model.descriptor.bad import SparseBAD  # noqa: E402
from onnx_image_processing_amd.synth import synth_batch  # noqa: E402

H, W, K = 480, 640, 512


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=["corner", "nms", "topk", "bad", "cost", "sinkhorn", "mnn"])
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--radius", type=int, default=5)
    ap.add_argument("--block", type=int, default=3)
    args = ap.parse_args()
    dev = "cuda:0"
    n = args.images
    a, _ = synth_batch(1000, 8, H, W)
    img = torch.from_numpy(a).to(dev).repeat((n + 7) // 8, 1, 1, 1)[:n].contiguous()
    px = n * H * W
    res = {}
    score = ops.corner_response(img, 3)
    if "corner" in args.which and args.block != 3:
        N.use_debug_library().mi_debug_set(1, 1)
        ref5 = ops.corner_response(img, args.block)
        for impl, rows, name in ((1, 8, "corner(tile)"), (0, 8, "corner(stream R=8)"), (0, 5, "corner(stream R=5)"),
                                 (0, 4, "corner(stream R=4)")):
            N.use_debug_library().mi_debug_set(1, impl)
            N.use_debug_library().mi_debug_set(2, rows)
            alt = ops.corner_response(img, args.block)
            assert torch.equal(alt, ref5), name
            ms = timeit(lambda: ops.corner_response(img, args.block), args.iters)
            res[f"block {args.block} {name}"] = (ms, 8.0 * px / ms / 1e6)
        N.use_debug_library().mi_debug_set(1, 0)
        N.use_debug_library().mi_debug_set(2, 4)
    if "corner" in args.which and args.block == 3:
        for impl, rows, name in ((1, 8, "corner(tile)"), (0, 8, "corner(stream R=8)"), (0, 5, "corner(stream R=5)"),
                                 (0, 4, "corner(stream R=4)")):
            N.use_debug_library().mi_debug_set(1, impl)
            N.use_debug_library().mi_debug_set(2, rows)
            alt = ops.corner_response(img, 3)
            assert torch.equal(alt, score), name
            ms = timeit(lambda: ops.corner_response(img, 3), args.iters)
            res[name] = (ms, 8.0 * px / ms / 1e6)
    sc = score.squeeze(1)
    if "nms" in args.which:
        cand, count, seg, cap = ops._candidate_buffers(n, H, W, dev)

        def f():
            N.call("mi_nms_candidates", sc.data_ptr(), n, H, W, args.radius, 0.0, 7, cand.data_ptr(), count.data_ptr(),
                   N.stream_ptr())
        ms = timeit(f, args.iters)
        res["nms_candidates"] = (ms, 4.0 * px / ms / 1e6)
        ms = timeit(lambda: ops.nms_mask(sc, args.radius), args.iters)
        res["nms_mask"] = (ms, 8.0 * px / ms / 1e6)
        print("candidates per image:", float(count.float().sum(1).mean()))
    kp, _ = ops.nms_topk(sc, args.radius, K, 0.0, 7)
    if "topk" in args.which:
        cand, count, seg, cap = ops._candidate_buffers(n, H, W, dev)
        N.call("mi_nms_candidates", sc.data_ptr(), n, H, W, args.radius, 0.0, 7, cand.data_ptr(), count.data_ptr(),
               N.stream_ptr())
        ms = timeit(lambda: ops._topk_from_candidates(cand, count, seg, cap, n, H, W, K), args.iters)
        res["topk"] = (ms, 0.0)
    bad = SparseBAD(512, binarize=True, soft_binarize=False).to(dev)
    if "bad" in args.which:
        ms = timeit(lambda: bad.forward_bits(img, kp), args.iters)
        res["bad_bits"] = (ms, 0.0)
        ms = timeit(lambda: bad(img, kp), args.iters)
        res["bad_f32"] = (ms, 4.0 * n * K * 512 / ms / 1e6)
    if "badori" in args.which:
        # rotation-aware BAD on per-keypoint angles (the VO / AKAZE matchers' form): unrolled bits kernel (hook 13 = 0)
        # against the generic kernel (1), same bits
        g = torch.Generator(device="cpu").manual_seed(5)
        ang = ((torch.rand((n, K), generator=g) * 2 - 1) * 3.14159).to(dev)
        for bad_o, tag in ((bad, "512"), (SparseBAD(256, binarize=True, soft_binarize=False).to(dev), "256")):
            ref_bits = None
            for impl, name in ((1, "generic"), (0, "bits kernel")):
                N.use_debug_library().mi_debug_set(13, impl)
                got = bad_o.forward_bits(img, kp, ang)
                assert ref_bits is None or torch.equal(got, ref_bits), name
                ref_bits = got
                ms = timeit(lambda: bad_o.forward_bits(img, kp, ang), args.iters)
                res[f"bad_oriented_bits {tag} ({name})"] = (ms, 0.0)
        N.use_debug_library().mi_debug_set(13, 0)
        for kw, tag in ((dict(binarize=False), "256 raw"), (dict(binarize=True, soft_binarize=True), "256 soft"),
                        (dict(binarize=True, soft_binarize=False), "256 hard f32")):
            bad_d = SparseBAD(256, **kw).to(dev)
            ref_d = None
            for impl, name in ((1, "generic"), (0, "fast kernel")):
                N.use_debug_library().mi_debug_set(13, impl)
                got = bad_d(img, kp, ang)
                assert ref_d is None or torch.equal(got, ref_d), (tag, name, float((got - ref_d).abs().max()))
                ref_d = got
                res[f"bad_oriented_desc {tag} ({name})"] = (timeit(lambda: bad_d(img, kp, ang), args.iters), 0.0)
        N.use_debug_library().mi_debug_set(13, 0)
        fr = img.clone()
        fr[::2, :, 100:200, 100:300] += 0.25                        # half the images: windows that are not uint8-valued
        for impl, name in ((1, "generic"), (0, "bits kernel")):
            N.use_debug_library().mi_debug_set(13, impl)
            got = bad.forward_bits(fr, kp, ang)
            assert impl == 1 or torch.equal(got, ref_bits), "fp64 rest"
            ref_bits = got
            res[f"bad_oriented_bits 512, part fp64 ({name})"] = (timeit(lambda: bad.forward_bits(fr, kp, ang), args.iters), 0.0)
        N.use_debug_library().mi_debug_set(13, 0)
    bits = bad.forward_bits(img, kp)
    b2 = torch.roll(bits, 1, 0)
    if "cost" in args.which:
        ms = timeit(lambda: ops.cost_logscores_bits(bits, b2, True, 0.05), args.iters)
        res["cost_bits"] = (ms, 4.0 * n * K * K / ms / 1e6)
        d = bad(img[:32], kp[:32])
        ms = timeit(lambda: ops.cost_logscores_f32(d, torch.roll(d, 1, 0), 0, 0.05), args.iters)
        res["cost_f32(32 pairs)"] = (ms, 2.0 * 32 * K * K * 512 / ms / 1e9)
    z, pitch = ops.cost_logscores_bits(bits, b2, True, 0.05)
    if "sinkhorn" in args.which:
        for mode, name in ((1, "sinkhorn_fused log-partials"), (2, "sinkhorn_fused prob-partials v1"),
                           (0, "sinkhorn_fused prob-partials lean")):
            N.use_debug_library().mi_debug_set(4, mode)
            ms = timeit(lambda: ops.sinkhorn(z, K, pitch, -20.0, 20), args.iters)
            res[name] = (ms, (21.0 * 4 * n * K * K + 4.0 * n * (K + 1) ** 2) / ms / 1e6)
        ms = timeit(lambda: ops.sinkhorn(z, K, pitch, -20.0, 20, use_workspace=False), args.iters)
        res["sinkhorn_2pass(20 it)"] = (ms, (41.0 * 4 * n * K * K + 4.0 * n * (K + 1) ** 2) / ms / 1e6)
        ref_p = None
        for mode, name in ((1, "one stream"), (2, "2 parts"), (3, "3 parts"), (4, "4 parts")):
            N.use_debug_library().mi_debug_set(6, mode)
            pp_ = ops.sinkhorn_bits(bits, b2, True, 0.05, 1.0, 20)
            assert os.environ.get("KBENCH_NOCHECK") or ref_p is None or torch.equal(pp_, ref_p), name
            ref_p = pp_
            ms = timeit(lambda: ops.sinkhorn_bits(bits, b2, True, 0.05, 1.0, 20), args.iters)
            res[f"cost+sinkhorn dots(20 it, {name})"] = (ms, (20.0 * 2 * n * K * K + 4.0 * n * (K + 1) ** 2) / ms / 1e6)
        N.use_debug_library().mi_debug_set(6, 2)
    p = ops.sinkhorn(z, K, pitch, -20.0, 20)
    if "mnn" in args.which:
        ms = timeit(lambda: ops.mnn_extract(p, kp, torch.roll(kp, 1, 0), 100, 0.1), args.iters)
        res["mnn"] = (ms, 2 * 4.0 * n * (K + 1) ** 2 / ms / 1e6)
    if "acc" in args.which:
        # accuracy margin of the Sinkhorn path vs the fp64 oracle (bound: 1e-4 * max(1, |P|))
        from oracle import numpy_oracle as O
        rng = np.random.default_rng(1)
        b1 = rng.random((2, 512, 512)) < 0.4
        b2 = rng.random((2, 512, 512)) < 0.4
        b2[:, :256] = b1[:, :256]

        def desc(bb):
            f = bb.astype(np.float32)
            return f / np.maximum(np.sqrt(f.sum(-1, keepdims=True, dtype=np.float32)), np.float32(1e-12))
        ref = O.sinkhorn_match(desc(b1).astype(np.float64), desc(b2).astype(np.float64), 20, 0.05, 1.0, dtype=np.float64)
        tb1 = torch.from_numpy(O.pack_bits(b1).view(np.int32)).to(dev)
        tb2 = torch.from_numpy(O.pack_bits(b2).view(np.int32)).to(dev)
        zz, pp = ops.cost_logscores_bits(tb1, tb2, True, 0.05)
        for name, ws in (("fused", True), ("two-pass", False), ("dots", None)):
            got = (ops.sinkhorn_bits(tb1, tb2, True, 0.05, 1.0, 20) if ws is None else
                   ops.sinkhorn(zz, 512, pp, -20.0, 20, use_workspace=ws)).cpu().numpy().astype(np.float64)
            ratio = np.abs(got - ref) / (1e-4 * np.maximum(1.0, np.abs(ref)))
            print(f"accuracy {name}: worst |dP|/bound = {ratio.max():.4f}, max |dP| core = {np.abs(got - ref)[:, :512, :512].max():.3e}")
    for k, (ms, gbs) in res.items():
        print(f"{k:26s} {ms:8.3f} ms   {gbs:9.1f} GB/s (or GFLOP/s)")


if __name__ == "__main__":
    main()
