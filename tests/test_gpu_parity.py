"""GPU parity: the HIP path (through the C ABI, via the nn.Module mirrors) against the oracle
and against the reference-generated golden fixtures.  Run with `-m gpu` on an MI355X.

Bars (north_star): keypoint indices and BAD bit strings bit-exact; Shi-Tomasi scores bit-exact
for uint8-valued input / block 3; Sinkhorn P within 1e-4 (relative on the dustbin row/column).
"""
import os

import numpy as np
import pytest
import torch

from helpers import (ALLOW, bad_tables, bits_mismatch, cfg_of, check_match_sets, load_golden, match_dict, p_close, permute_p,
                     tie_canonical_perm, unpack_bits)
from gpu_common import DEV, ROOT, _images, _status_word, gpu, mods  # noqa: F401  (mods: the module fixture)
from onnx_image_processing_amd.synth import synth_batch, synth_image
from oracle import numpy_oracle as O

pytestmark = pytest.mark.gpu



# ------------------------------------------------------------------ K1
@pytest.mark.parametrize("shape,bs", [((2, 480, 640), 3), ((3, 96, 128), 3), ((1, 37, 64), 3), ((2, 61, 83), 3),
                                      ((1, 8, 8), 3), ((1, 200, 136), 3)])
def test_corner_bitexact_uint8_bs3(mods, shape, bs):
    n, h, w = shape
    img = np.stack([synth_image(500 + i, h, w) for i in range(n)])[:, None].astype(np.float32)
    got = mods["ShiTomasiScore"](bs)(gpu(img)).cpu().numpy()
    assert np.array_equal(got, O.shi_tomasi_score(img, bs))


def test_corner_c1_golden(mods):
    g = load_golden("c1_shi_tomasi")
    img = synth_image(int(g["seed"]))[None, None].astype(np.float32)
    got = mods["ShiTomasiScore"](3)(gpu(img)).cpu().numpy()
    ref, root = O.shi_tomasi_score(img, 3, return_sqrt_term=True)
    assert np.array_equal(got, ref)
    diff = np.abs(got[0, 0].astype(np.float64) - g["score3"])       # reference run: MKL sqrt is 1 ulp off on <1 %
    assert np.all(diff <= np.spacing(root[0, 0]) + np.spacing(ref[0, 0])) and (diff > 0).mean() < 0.02


@pytest.mark.parametrize("bs,shape", [(5, (2, 96, 128)), (7, (1, 64, 128)), (5, (1, 61, 83)), (9, (1, 40, 52))])
def test_corner_other_blocks(mods, bs, shape):
    n, h, w = shape
    img = np.stack([synth_image(600 + i, h, w) for i in range(n)])[:, None].astype(np.float32)
    got = mods["ShiTomasiScore"](bs)(gpu(img)).cpu().numpy()
    ref = O.shi_tomasi_score(img, bs)
    # sums may exceed 2^24 for bs >= 5: order-dependent rounding, relative to the trace scale
    assert np.abs(got - ref).max() <= 2e-6 * max(float(ref.max()), 1.0) + 1.0


def test_corner_float_image_tolerance(mods):
    g = load_golden("c1_shi_tomasi")
    for bs, key in ((3, "float_score3"), (7, "float_score7")):
        got = mods["ShiTomasiScore"](bs)(gpu(g["float_img"])).cpu().numpy()
        assert np.abs(got - g[key]).max() <= 4e-6 * float(g[key].max()) + 1.0


# ------------------------------------------------------------------ K2 / K3
def test_nms_mask_golden_and_oracle(mods):
    g = load_golden("nms_topk")
    s = g["plateau_scores"]
    for r in (1, 2, 3, 5):
        m = mods["apply_nms_maxpool"](gpu(s), r).cpu().numpy()
        assert np.array_equal(np.packbits(m.astype(bool)), g[f"mask_r{r}"])
    img = synth_image(77, 200, 264)[None, None].astype(np.float32)
    sc = O.shi_tomasi_score(img, 3)[:, 0]
    for r in (0, 3, 5, 9):
        assert np.array_equal(mods["apply_nms_maxpool"](gpu(sc), r).cpu().numpy(), O.nms_mask(sc, r))


@pytest.mark.parametrize("radius", [1, 2, 3, 4, 5, 6, 7, 8])
def test_nms_fast_kernel_every_radius(mods, radius):
    """The one-plane fast kernel (w % 4 == 0, r <= 8), mask and candidate modes, on maps with negative values,
    plateaus, ragged tile edges and several images; against the oracle."""
    rng = np.random.default_rng(radius)
    for (n, h, w) in ((2, 75, 132), (1, 32, 128), (3, 97, 260), (1, 5, 8)):
        sc = rng.standard_normal((n, h, w)).astype(np.float32)
        sc[:, : h // 2] = np.round(sc[:, : h // 2] * 2) / 2                     # exact ties in the upper half
        got = mods["apply_nms_maxpool"](gpu(sc), radius).cpu().numpy()
        ref = O.nms_mask(sc, radius)
        assert np.array_equal(got, ref), (radius, n, h, w)
        k = min(64, h * w)
        for thr, margin in ((0.0, 0), (0.25, 3)):
            kp_ref, sc_ref, _ = O.select_topk_keypoints(sc, ref, k, thr, margin)
            kp, ks = mods["detect_keypoints"](gpu(sc), radius, k, thr, margin)
            assert np.array_equal(kp.cpu().numpy(), kp_ref) and np.array_equal(ks.cpu().numpy(), sc_ref), (radius, h, w, thr)


def test_topk_golden_tie_free(mods):
    g = load_golden("nms_topk")
    s = gpu(g["uniq_scores"])
    for i in range(4):
        r, k, thr, margin = g[f"t{i}_args"]
        mask = mods["apply_nms_maxpool"](s, int(r))
        kp, sc = mods["select_topk_keypoints"](s, mask, int(k), float(thr), int(margin))
        kp2, sc2 = mods["detect_keypoints"](s, int(r), int(k), float(thr), int(margin))
        assert torch.equal(kp, kp2) and torch.equal(sc, sc2)                     # fused == two-call form
        kp, sc = kp.cpu().numpy(), sc.cpu().numpy()
        assert np.array_equal(sc, g[f"t{i}_scores"])
        valid = sc > 0
        assert np.array_equal(kp[valid], g[f"t{i}_kpts"][valid]) and np.all(kp[~valid] == -1)


def test_topk_plateaus_vs_oracle(mods):
    """Massive exact ties: every pixel of a flat-ish map is a candidate (> 4096 -> radix-select path)."""
    g = load_golden("nms_topk")
    s = g["plateau_scores"]
    big = np.tile(s, (1, 4, 4))[:, :150, :200].copy()                           # 30000 px, 12 distinct values
    for arr, r, k in ((s, 2, 200), (big, 1, 512), (big, 0, 4096), (big + 1.0, 0, 100)):
        mask = O.nms_mask(arr, r)
        kp_ref, sc_ref, _ = O.select_topk_keypoints(arr, mask, k, 0.0, 0)
        kp, sc = mods["detect_keypoints"](gpu(arr), r, k, 0.0, 0)
        assert np.array_equal(kp.cpu().numpy(), kp_ref) and np.array_equal(sc.cpu().numpy(), sc_ref)


def test_topk_big_kernel_equals_small_kernel_and_oracle(mods):
    """Large images run top-k with the scores of up to 32,768 candidates resident in LDS (topk_kernel<BIG>: score select in
    LDS, one gather, merge-rank sort) instead of eight passes over global memory.  Both kernels (test hook key 10) must
    return the same keypoints and scores, and the oracle's, on: a 1080p corner map (~22 k candidates; k = 1024, 512,
    100 and 2000 > 1024), a quantised map whose k-th place falls inside a run of EQUAL scores (the tie is broken by the
    linear index, so the bin's keys go through the key select), a map with more than 4,096 keys tied at the k-th place
    (falls back to the global-memory select), and a sparse map with fewer candidates than k."""
    from onnx_image_processing_amd import _native as N, ops
    a, _ = synth_batch(4100, 2, 1080, 1920)
    sc = ops.corner_response(gpu(a), 3).squeeze(1)
    quant = (sc / sc.amax() * 40.0).round()                     # 41 distinct values: long runs of equal scores
    coarse = (sc > 0).float() * 3.0 + (sc > sc.mean()).float()  # 3 values: tens of thousands tied at the k-th place
    sparse = torch.zeros_like(sc)
    sparse[:, 100:900:40, 100:1800:50] = sc[:, 100:900:40, 100:1800:50] + 1.0      # 20 x 34 = 680 candidates per image
    cases = ((sc, 5, (1024, 512, 100, 2000)), (quant, 5, (1024, 300)), (coarse, 3, (1024,)), (sparse, 5, (1024, 64)))
    for scores, r, ks in cases:
        ncand = int((ops.nms_mask(scores, r) * (scores > 0)).sum(dim=(1, 2)).max())
        for k in ks:
            with N.debug_library() as lib:
                assert lib.mi_debug_set(10, 0) == 0
                kp0, s0 = ops.nms_topk(scores, r, k, 0.0, 7)
                assert lib.mi_debug_set(10, 1) == 0
                kp1, s1 = ops.nms_topk(scores, r, k, 0.0, 7)
            kp2, s2 = ops.nms_topk(scores, r, k, 0.0, 7)        # the product library (big kernel by segment count)
            assert torch.equal(kp0, kp1) and torch.equal(s0, s1), (ncand, r, k)
            assert torch.equal(kp1, kp2) and torch.equal(s1, s2), (ncand, r, k)
        arr = scores[:1].cpu().numpy()
        kp_ref, sc_ref, _ = O.select_topk_keypoints(arr, O.nms_mask(arr, r), ks[0], 0.0, 7)
        kp, s_ = ops.nms_topk(scores[:1], r, ks[0], 0.0, 7)
        assert np.array_equal(kp.cpu().numpy(), kp_ref) and np.array_equal(s_.cpu().numpy(), sc_ref), (ncand, r)
    # the small images' kernel forced onto the big path's code and vice versa: 640x480 through the big kernel
    b, _ = synth_batch(3100, 2, 480, 640)
    sc = ops.corner_response(gpu(b), 3).squeeze(1)
    with N.debug_library() as lib:
        lib.mi_debug_set(10, 1)
        kp1, s1 = ops.nms_topk(sc, 5, 512, 0.0, 7)
    kp2, s2 = ops.nms_topk(sc, 5, 512, 0.0, 7)
    assert torch.equal(kp1, kp2) and torch.equal(s1, s2)


def test_topk_select_path_equals_full_sort(mods):
    """k << candidates: the radix-select + sort-k path and the full bitonic sort return the same keypoints and
    scores (640x480: ~3300 candidates per image, k = 512 and k = 100; a plateau map with massive ties too)."""
    from onnx_image_processing_amd import _native as N, ops
    a, _ = synth_batch(3100, 3, 480, 640)
    sc = ops.corner_response(gpu(a), 3).squeeze(1)
    g = load_golden("nms_topk")
    plateau = gpu(np.tile(g["plateau_scores"], (1, 4, 4))[:, :150, :200].copy())
    with N.debug_library() as lib:                 # the kernel-variant hooks exist only in the debug build of the library
        for scores, r, ks in ((sc, 5, (512, 100, 7)), (plateau, 1, (512, 64))):
            for k in ks:
                assert lib.mi_debug_set(9, 0) == 0
                kp0, s0 = ops.nms_topk(scores, r, k, 0.0, 0)
                assert lib.mi_debug_set(9, 1) == 0
                kp1, s1 = ops.nms_topk(scores, r, k, 0.0, 0)
                assert torch.equal(kp0, kp1) and torch.equal(s0, s1), (r, k)
    for scores, r, k in ((sc, 5, 512), (plateau, 1, 64)):       # and the product library gives the same again
        with N.debug_library():
            kp1, s1 = ops.nms_topk(scores, r, k, 0.0, 0)
        kp2, s2 = ops.nms_topk(scores, r, k, 0.0, 0)
        assert torch.equal(kp1, kp2) and torch.equal(s1, s2)


def test_topk_errors_and_empty(mods):
    s = torch.zeros(1, 16, 16, device=DEV)
    kp, sc = mods["detect_keypoints"](s, 2, 10, 0.0, 0)
    assert torch.all(kp == -1) and torch.all(sc == 0)
    with pytest.raises(RuntimeError):
        mods["detect_keypoints"](s, 2, 300, 0.0, 0)                              # k > H*W, as torch.topk


# ------------------------------------------------------------------ K4
@pytest.mark.parametrize("num_pairs", [256, 512])
def test_sparse_bad_modes_vs_oracle(mods, num_pairs):
    box, thr = bad_tables(num_pairs)
    a, _ = synth_batch(900, 2, 120, 160)
    rng = np.random.default_rng(3)
    kp = np.stack([rng.integers(0, 120, (2, 90)), rng.integers(0, 160, (2, 90))], -1).astype(np.float32)
    kp[0, 5] = (-1, -1)                                                          # invalid keypoint
    kp[1, :4] = [(0, 0), (119, 159), (0, 159), (119, 0)]                         # corners: clamped boxes
    for kw in (dict(binarize=False, normalize_descriptors=False), dict(binarize=False),
               dict(binarize=True, soft_binarize=True, temperature=4.0),
               dict(binarize=True, soft_binarize=False), dict(binarize=True, soft_binarize=False, normalize_descriptors=False)):
        mod = mods["SparseBAD"](num_pairs=num_pairs, **kw).to(DEV)
        got = mod(gpu(a), gpu(kp)).cpu().numpy()
        ref, aux = O.sparse_bad(a, kp, box, thr, return_aux=True, **kw)
        if kw.get("binarize") and not kw.get("soft_binarize", True):
            assert np.array_equal(got, ref)                                      # bit-exact incl. fp32 normalisation
            bits = mod.forward_bits(gpu(a), gpu(kp)).cpu().numpy().view(np.uint32)
            assert np.array_equal(bits, O.pack_bits(aux["bits"]))
        else:
            np.testing.assert_allclose(got, ref, rtol=0, atol=3e-5 if kw.get("normalize_descriptors", True) else 2e-4)
        assert np.all(got[0, 5] == 0)


def test_sparse_bad_fast_path_equals_general(mods):
    """The int32 fast path (interior integer keypoints on uint8 patches) and the general fp64 path give
    identical bits and descriptors; non-integer pixels or fractional keypoints fall back per keypoint."""
    a, _ = synth_batch(910, 2, 240, 320)
    a[1, 0, 100:140, 100:180] += 0.5                                            # non-integer patch region
    rng = np.random.default_rng(4)
    kp = np.stack([rng.integers(0, 240, (2, 300)), rng.integers(0, 320, (2, 300))], -1).astype(np.float32)
    kp[0, :6] = [(15, 15), (14, 15), (225, 305), (226, 305), (15, 306), (-1, -1)]   # eligibility edges
    kp[0, 6] = (100.5, 50.0)                                                    # fractional keypoint
    box, thr = bad_tables(512)
    for kw in (dict(normalize_descriptors=True), dict(normalize_descriptors=False)):
        fast = mods["SparseBAD"](512, binarize=True, soft_binarize=False, **kw).to(DEV)
        slow = mods["SparseBAD"](512, binarize=True, soft_binarize=False, **kw).to(DEV)
        slow.use_fast_path = False
        assert torch.equal(fast(gpu(a), gpu(kp)), slow(gpu(a), gpu(kp)))
        bf = fast.forward_bits(gpu(a), gpu(kp))
        assert torch.equal(bf, slow.forward_bits(gpu(a), gpu(kp)))
        _, aux = O.sparse_bad(a, kp, box, thr, binarize=True, soft_binarize=False, return_aux=True, **kw)
        assert np.array_equal(bf.cpu().numpy().view(np.uint32), O.pack_bits(aux["bits"]))


def test_sparse_bad_plan_with_bad_geometry_is_not_used(mods):
    """A pair table with a box that leaves the 32x32 patch fails the plan's geometry check: the fast kernel
    must hand every keypoint to the general kernel (same output as without a plan)."""
    from onnx_image_processing_amd import ops
    a, _ = synth_batch(911, 1, 120, 160)
    rng = np.random.default_rng(5)
    kp = np.stack([rng.integers(0, 120, (1, 64)), rng.integers(0, 160, (1, 64))], -1).astype(np.float32)
    m = mods["SparseBAD"](512, binarize=True, soft_binarize=False).to(DEV)
    geom = m.pair_geom.clone()
    q = int(geom[3].item()) & 0xFFFFFFFF
    q = (q & ~(31 | (15 << 20))) | 31 | (7 << 20)                # pair 3: x offset +15, radius 7 -> box past the patch edge
    geom[3] = q if q < 2 ** 31 else q - 2 ** 32
    plan = ops.bad_plan(geom, m.pair_thr)
    _, with_plan = ops.sparse_bad(gpu(a), gpu(kp), geom, m.pair_thr, m.mode, m.temperature, True, want_desc=False,
                                  want_bits=True, plan=plan)
    _, without = ops.sparse_bad(gpu(a), gpu(kp), geom, m.pair_thr, m.mode, m.temperature, True, want_desc=False,
                                want_bits=True, plan=None)
    assert torch.equal(with_plan, without)


# ------------------------------------------------------------------ K5 / K6
def test_sinkhorn_unit_golden(mods):
    g = load_golden("sinkhorn_unit")
    for i in range(4):
        kw = cfg_of(g, f"s{i}_cfg")
        p = mods["SinkhornMatcher"](**kw)(gpu(g["d1"]), gpu(g["d2"])).cpu().numpy()
        ok, worst = p_close(p, g[f"s{i}_P"], atol=3e-5)
        assert ok, (i, kw, worst)


@pytest.mark.parametrize("n,m,d", [(512, 512, 512), (130, 67, 256), (1, 1, 32), (300, 513, 64), (1024, 1024, 512)])
def test_sinkhorn_bits_vs_float_vs_oracle(mods, n, m, d):
    rng = np.random.default_rng(n + m)
    b1 = rng.random((2, n, d)) < 0.4
    b2 = rng.random((2, m, d)) < 0.4
    b2[:, : min(n, m) // 2] = b1[:, : min(n, m) // 2]                            # true matches
    b1[0, 0] = False                                                             # an all-zero (invalid) descriptor
    for normalized, eps, unused in ((True, 0.05, 1.0), (False, 16.0, 60.0)):
        def desc(bits):
            f = bits.astype(np.float32)
            if normalized:
                nrm = np.sqrt(f.sum(-1, keepdims=True, dtype=np.float32))
                f = f / np.maximum(nrm, np.float32(1e-12))
            return f
        d1, d2 = desc(b1), desc(b2)
        ref = O.sinkhorn_match(d1.astype(np.float64), d2.astype(np.float64), 20, eps, unused, dtype=np.float64)
        mt = mods["SinkhornMatcher"](iterations=20, epsilon=eps, unused_score=unused)
        pb = mt.forward_bits(gpu(O.pack_bits(b1).view(np.int32)), gpu(O.pack_bits(b2).view(np.int32)), normalized)
        pf = mt(gpu(d1), gpu(d2))
        mt.use_dot_storage = True                     # uint16 dot-product storage (m <= 1024), same answer
        pd = mt.forward_bits(gpu(O.pack_bits(b1).view(np.int32)), gpu(O.pack_bits(b2).view(np.int32)), normalized)
        for name, p in (("bits", pb), ("f32", pf), ("dots", pd)):
            ok, worst = p_close(p.cpu().numpy(), ref)
            assert ok, (name, normalized, worst)


@pytest.mark.parametrize("b,n,m,bits", [(2, 512, 512, 512), (3, 300, 77, 256), (1, 40, 1000, 512), (2, 130, 130, 96)])
def test_packed_dot_products_are_exact_popcounts(mods, b, n, m, bits):
    """mi_cost_dots_bits: the uint16 dot products of packed hard-bit descriptors are popcount(a & b) EXACTLY -- on the FP4
    MFMA (256 / 512 bits: a bit becomes the nibble 1.0 / 0.0, fp32 sums <= 4096 are exact; default), on the int8 MFMA
    (debug key 14 = 1; also what other lengths run) and against numpy, for all-ones / all-zeros / random descriptors and
    shapes that are not multiples of the tile; the per-descriptor (scale, squared norm) pairs agree too, and the fp32
    log-scores of mi_cost_logscores_bits are the same floats from both."""
    from onnx_image_processing_amd import _native as N, ops
    rng = np.random.default_rng(bits + n)
    w = bits // 32
    b1 = rng.integers(0, 2 ** 32, size=(b, n, w), dtype=np.uint64).astype(np.uint32)
    b2 = rng.integers(0, 2 ** 32, size=(b, m, w), dtype=np.uint64).astype(np.uint32)
    b1[0, 0], b1[0, 1], b2[0, 0], b2[0, 1] = 0xFFFFFFFF, 0, 0xFFFFFFFF, 0
    b1[0, 2], b2[0, 2] = 0xAAAAAAAA, 0x55555555
    b2[0, 3] = b1[0, 3]
    t1, t2 = gpu(b1.view(np.int32)), gpu(b2.view(np.int32))
    want = np.zeros((b, n, m), np.int64)
    for k in range(w):
        x = b1[:, :, None, k] & b2[:, None, :, k]
        want += np.unpackbits(np.ascontiguousarray(x).view(np.uint8).reshape(b, n, m, 4), axis=-1).sum(-1, dtype=np.int64)

    def dots():
        _, _, _, (d, ri, ci, pitch, _) = ops.sinkhorn_bits(t1, t2, True, 0.05, 1.0, 1, return_state=True)
        return d[:, :, :m].clone(), ri.clone(), ci.clone()
    got = dots()
    assert np.array_equal(got[0].cpu().numpy().astype(np.int64), want)
    z = ops.cost_logscores_bits(t1, t2, True, 0.05)[0][:, :, :m].clone()
    with N.debug_library() as lib:
        assert lib.mi_debug_set(14, 1) == 0
        alt = dots()
        z_alt = ops.cost_logscores_bits(t1, t2, True, 0.05)[0][:, :, :m].clone()
    for x, y in zip(got, alt):
        assert torch.equal(x, y)
    assert torch.equal(z, z_alt)


@pytest.mark.parametrize("eps,unused", [(0.035, 1.0), (0.05, 0.2), (0.2, 1.0), (1.0, 3.0), (0.03, 1.0)])
def test_sinkhorn_dots_bounded_shift_vs_row_maximum(mods, eps, unused):
    """The bounded-shift row pass (one analytic shift per pair; taken when 2*sqnorm_bound/eps*log2(e) < 90) and
    the per-row-maximum kernel (sqnorm_bound = 0) against the fp64 oracle and against each other, from an
    easy regime to the edge of the fast path's range (eps = 0.035; eps = 0.03 is past it: both calls take
    the row-maximum kernel).  Includes empty descriptors and duplicated ones (cos = 1)."""
    from onnx_image_processing_amd import _native as N, ops
    rng = np.random.default_rng(int(eps * 1000))
    n, m, d = 200, 333, 256
    b1 = rng.random((2, n, d)) < 0.3
    b2 = rng.random((2, m, d)) < 0.3
    b2[:, :100] = b1[:, :100]
    b1[0, 5] = False
    b2[1, 7] = False

    def desc(bits):
        f = bits.astype(np.float32)
        return f / np.maximum(np.sqrt(f.sum(-1, keepdims=True, dtype=np.float32)), np.float32(1e-12))
    ref = O.sinkhorn_match(desc(b1).astype(np.float64), desc(b2).astype(np.float64), 20, eps, unused, dtype=np.float64)
    t1, t2 = gpu(O.pack_bits(b1).view(np.int32)), gpu(O.pack_bits(b2).view(np.int32))
    p_fast = ops.sinkhorn_bits(t1, t2, True, eps, unused, 20)
    ok, worst = p_close(p_fast.cpu().numpy(), ref)
    assert ok, worst
    # the same call with the bound withheld: per-row maxima
    _, _, _, (dots, ri, ci, pitch, _) = ops.sinkhorn_bits(t1, t2, True, eps, unused, 20, return_state=True)
    wbytes = int(N.load().mi_sinkhorn_dots_workspace_bytes(2, n, m))
    work = torch.empty((wbytes + 7) // 8, dtype=torch.int64, device=DEV)
    u = torch.empty((2, n + 1), device=DEV)
    v = torch.empty((2, m + 1), device=DEV)
    p_slow = torch.empty((2, n + 1, m + 1), device=DEV)
    N.call("mi_sinkhorn_dots", dots.data_ptr(), ri.data_ptr(), ci.data_ptr(), 2, n, m, pitch, float(eps), float(unused),
           0.0, 20, u.data_ptr(), v.data_ptr(), p_slow.data_ptr(), work.data_ptr(), wbytes, 0, N.stream_ptr())
    ok, worst = p_close(p_slow.cpu().numpy(), ref)
    assert ok, worst
    assert float(((p_fast - p_slow).abs() / p_slow.abs().clamp(min=1.0)).max()) < 2e-5      # relative on the dustbin entries


@pytest.mark.parametrize("batch,n,m,normalized,iters", [(1, 512, 512, True, 20), (8, 512, 512, True, 20), (3, 300, 470, True, 7),
                                                         (2, 512, 33, True, 1), (1, 31, 512, False, 20), (5, 512, 511, False, 3),
                                                         (1, 1, 1, True, 4)])
def test_sinkhorn_single_launch_equals_multi_launch(mods, batch, n, m, normalized, iters):
    """mi_sinkhorn_dots for <= 8 pairs runs ONE persistent launch (bands exchange column sums as tagged granules);
    its duals and P must equal the 41-launch form's bit for bit, for both row-pass variants (bounded shift for unit
    descriptors, per-row maxima for raw bit vectors), ragged shapes and repeated calls on one workspace."""
    from onnx_image_processing_amd import _native as N, ops
    rng = np.random.default_rng(batch * 1000 + n + m)
    words = 16
    b1 = rng.integers(0, 2 ** 32, size=(batch, n, words), dtype=np.uint64).astype(np.uint32)
    b2 = rng.integers(0, 2 ** 32, size=(batch, m, words), dtype=np.uint64).astype(np.uint32)
    k = min(n, m) // 2
    b2[:, :k] = b1[:, :k]                                        # true matches
    t1, t2 = gpu(b1.view(np.int32)), gpu(b2.view(np.int32))
    eps, unused = (0.05, 1.0) if normalized else (24.0, 300.0)
    try:                                                         # the caller's flag selects the form (product library)
        ops.set_solver_flags(ops.MI_SOLVER_MULTI_LAUNCH)
        want = [t.clone() for t in ops.sinkhorn_bits(t1, t2, normalized, eps, unused, iters, return_duals=True)]
        ops.set_solver_flags(ops.MI_SOLVER_DEFAULT)
        for _ in range(3):                                       # the same shapes again: allocator reuse, stale granules
            *got, state = ops.sinkhorn_bits(t1, t2, normalized, eps, unused, iters, return_state=True)
            for x, y in zip(got, want):
                assert torch.equal(x, y)
            assert _status_word(state) == 0
    finally:
        ops.set_solver_flags(ops.MI_SOLVER_DEFAULT)
    ref = O.sinkhorn_match(*[unpack_bits(b, 512).astype(np.float64) / (np.sqrt(unpack_bits(b, 512).sum(-1, keepdims=True)) if normalized else 1.0)
                             for b in (b1, b2)], iters, eps, unused, "l2", dtype=np.float64)
    ok, worst = p_close(got[0].cpu().numpy(), ref)
    assert ok, worst


@pytest.mark.parametrize("n,m", [(512, 512), (97, 301), (5, 3), (700, 1000), (64, 1500)])
def test_sinkhorn_fused_equals_two_pass(mods, n, m):
    """The band-fused iteration (Z read once) and the two-pass form agree to fp32 rounding, and both
    match the fp64 oracle; m > 1024 exercises the generic two-pass kernels."""
    from onnx_image_processing_amd import ops
    rng = np.random.default_rng(n * 7 + m)
    cost = (rng.random((2, n, m)) * 2.0).astype(np.float32)
    eps, unused = 0.05, 1.0
    pitch = (m + 3) // 4 * 4
    z = np.full((2, n, pitch), np.nan, np.float32)               # poison the row padding
    z[:, :, :m] = -cost / np.float32(eps)
    ref = O.sinkhorn_from_cost(cost.astype(np.float64), 20, eps, unused, dtype=np.float64)
    pa = ops.sinkhorn(gpu(z), m, pitch, -unused / eps, 20, use_workspace=True).cpu().numpy()
    pb = ops.sinkhorn(gpu(z), m, pitch, -unused / eps, 20, use_workspace=False).cpu().numpy()
    for name, p in (("fused", pa), ("two-pass", pb)):
        ok, worst = p_close(p, ref)
        assert ok, (name, worst)
    assert np.abs(pa - pb).max() <= 2e-5 * max(1.0, float(np.abs(pb).max()))


def test_sinkhorn_with_scores(mods):
    g = load_golden("sinkhorn_unit")
    p, s0, s1 = mods["SinkhornMatcherWithScores"](iterations=5, epsilon=0.1, unused_score=0.7)(gpu(g["d1"]), gpu(g["d2"]))
    assert torch.equal(s0, p[:, :40, :56].amax(-1)) and torch.equal(s1, p[:, :40, :56].amax(-2))


# ------------------------------------------------------------------ K7
def test_filters_golden_and_known_answers(mods):
    from onnx_image_processing_amd.pytorch_model.matching import SinkhornMatcherWithFilters
    from onnx_image_processing_amd import ops
    from test_oracle_golden import KNOWN_RATIO, _augment
    for core, thr, expect in KNOWN_RATIO:                      # reference test_vectorized_filter.py vectors
        _, valid = ops.match_filters(gpu(_augment(core)), thr, -1.0)
        assert valid[0].cpu().tolist() == expect
    g = load_golden("filters_unit")
    for i in range(5):
        kw = cfg_of(g, f"f{i}_cfg")
        pf, valid = SinkhornMatcherWithFilters(**kw)(gpu(g["d1"]), gpu(g["d2"]))
        assert valid.dtype == torch.bool and np.array_equal(valid.cpu().numpy(), g[f"f{i}_valid"])
        ok, worst = p_close(pf.cpu().numpy(), g[f"f{i}_P"], atol=3e-5)
        assert ok, (i, worst)
    # random P with exact duplicates (top-2 multiplicity) and M == 1
    rng = np.random.default_rng(9)
    p = rng.random((3, 70, 131)).astype(np.float32)
    p[0, 5, 7] = p[0, 5, 90] = 2.0                               # best appears twice -> ratio 1
    for rt, dm in ((1.5, None), (None, 0.2), (1.2, 0.0)):
        ref_p, ref_v = O.match_filters(p, rt, dm)
        got_p, got_v = ops.match_filters(gpu(p.copy()), -1.0 if rt is None else rt, -1.0 if dm is None else dm)
        assert np.array_equal(got_v.cpu().numpy(), ref_v) and np.array_equal(got_p.cpu().numpy(), ref_p)
    p1 = rng.random((1, 9, 2)).astype(np.float32)
    ref_p, ref_v = O.match_filters(p1, 2.0, None)
    got_p, got_v = ops.match_filters(gpu(p1.copy()), 2.0, -1.0)
    assert np.array_equal(got_v.cpu().numpy(), ref_v) and np.array_equal(got_p.cpu().numpy(), ref_p)


def test_outlier_filters_module(mods):
    """matching/outlier_filters.py mirror (device tensors in, bool tensor out) against the reference's known answers
    and the oracle, including a negative margin, K = 1 and exact duplicates."""
    from onnx_image_processing_amd.pytorch_model.matching.outlier_filters import (dustbin_margin_filter,
                                                                                  probability_ratio_filter)
    from test_oracle_golden import KNOWN_DUSTBIN, KNOWN_RATIO
    for core, thr, expect in KNOWN_RATIO:
        got = probability_ratio_filter(gpu(core.astype(np.float32)), thr)
        assert got.dtype == torch.bool and got.cpu().tolist() == expect
    for full, margin, expect in KNOWN_DUSTBIN:
        assert dustbin_margin_filter(gpu(full.astype(np.float32)), margin).cpu().tolist() == expect
    rng = np.random.default_rng(12)
    p = (rng.random((200, 200)) ** 6).astype(np.float32)
    p[7, 3] = p[7, 150] = 1.5
    for thr in (1.0, 1.5, 3.0, 0.0):
        assert np.array_equal(probability_ratio_filter(gpu(p), thr).cpu().numpy(), O.probability_ratio_filter(p, thr))
    full = (rng.random((131, 131)) ** 3).astype(np.float32)
    for margin in (0.3, 0.0, -0.2):
        assert np.array_equal(dustbin_margin_filter(gpu(full), margin).cpu().numpy(), O.dustbin_margin_filter(full, margin))
    with pytest.raises(RuntimeError):
        probability_ratio_filter(gpu(p[None]), 2.0)
    # the reference's own call pattern (sample/image_matching.py:49-118): numpy in (float64 too), numpy bool out
    for core, thr, expect in KNOWN_RATIO:
        got = probability_ratio_filter(np.asarray(core, dtype=np.float64), thr)
        assert isinstance(got, np.ndarray) and got.dtype == np.bool_ and got.tolist() == expect
    for full_, margin, expect in KNOWN_DUSTBIN:
        got = dustbin_margin_filter(np.asarray(full_), margin)
        assert isinstance(got, np.ndarray) and got.tolist() == expect
    assert np.array_equal(dustbin_margin_filter(full, 0.3), O.dustbin_margin_filter(full, 0.3))


def test_mnn_vs_oracle(mods):
    rng = np.random.default_rng(5)
    for n, m, mx, thr in ((512, 512, 100, 0.1), (40, 56, 100, 0.01), (300, 77, 64, 0.0)):
        p = rng.random((2, n + 1, m + 1)).astype(np.float32) ** 8
        p[0, 3] = p[0, 4]                                                         # duplicate rows -> argmax ties
        k1 = rng.integers(0, 400, (2, n, 2)).astype(np.float32)
        k2 = rng.integers(0, 400, (2, m, 2)).astype(np.float32)
        ref = O.mnn_extract(p, k1, k2, mx, thr)
        got = mods["MutualNearestNeighborMatcher"](mx, thr)(gpu(p), gpu(k1), gpu(k2))
        for a, b in zip(got, ref[:4]):
            assert np.array_equal(a.cpu().numpy(), b)


# ------------------------------------------------------------------ composite


PIPELINES = ["c2_pair_480x640_k512", "c2_pair_noise_seed1001", "small_default_120x160_k64",
             "small_hamming_96x128_k48", "small_soft_l1_96x128_k32", "ragged_120x160_k96"]


@pytest.mark.parametrize("name", PIPELINES)
def test_pipeline_vs_oracle_and_golden(mods, name):
    g = load_golden(name)
    cfg = cfg_of(g)
    a, b = _images(g)
    model = mods["ShiTomasiSparseBADSinkhornMatcher"](max_keypoints=int(g["k"]), **cfg).to(DEV)
    k1, k2, p = [t.cpu().numpy() for t in model(gpu(a), gpu(b))]
    # --- vs the oracle on the same inputs: keypoints bit-exact, P within tolerance
    box, thr = bad_tables(cfg.get("num_pairs", 256))
    kw = {k: v for k, v in cfg.items() if k not in ("num_pairs", "sampling_mode")}
    o1, o2, op, aux = O.match_pair(a, b, box, thr, int(g["k"]), return_aux=True, **kw)
    assert np.array_equal(k1, o1) and np.array_equal(k2, o2)
    ok, worst = p_close(p, op)
    assert ok, f"P vs oracle: worst ratio {worst:.3g}"
    # --- vs the reference's own output (tie groups canonicalised)
    w = int(g["w"])
    perms = [tie_canonical_perm(g["kpts" + t][0], g["kscores" + t][0], w) for t in "12"]
    assert np.array_equal(k1[0], g["kpts1"][0][perms[0]]) and np.array_equal(k2[0], g["kpts2"][0][perms[1]])
    if "P" in g.files:
        ok, worst = p_close(p[0], permute_p(g["P"][0], perms[0], perms[1]))
        assert ok, f"P vs reference: worst ratio {worst:.3g}"
    if "mk1" in g.files:
        mcfg = cfg_of(g, "mnn_cfg")
        wrap = mods["MatchExtractionWrapper"](model, max_matches=mcfg["max_matches"], match_threshold=mcfg["threshold"])
        mk1, mk2, sc, valid = [t.cpu().numpy() for t in wrap(gpu(a), gpu(b))]
        assert np.array_equal(valid, g["mvalid"])
        gm = {(tuple(x), tuple(y)) for x, y, v in zip(g["mk1"][0], g["mk2"][0], g["mvalid"][0]) if v}
        hm = {(tuple(x), tuple(y)) for x, y, v in zip(mk1[0], mk2[0], valid[0]) if v}
        assert gm == hm                                                           # match-set parity


def test_pipeline_bits_bitexact_full_size(mods):
    """BAD bit strings of the north-star pair: bit-exact vs the oracle, and vs the reference except
    where the reference's own fp32 box means put a response within 5e-4 of its threshold."""
    g = load_golden("c2_pair_480x640_k512")
    cfg = cfg_of(g)
    a, b = _images(g)
    model = mods["ShiTomasiSparseBADSinkhornMatcher"](max_keypoints=512, **cfg).to(DEV)
    box, thr = bad_tables(512)
    for tag, im in (("1", a), ("2", b)):
        s = model.corner_detector(gpu(im)).squeeze(1)
        kp, _ = mods["detect_keypoints"](s, 5, 512, 0.0, 7)
        bits = model.descriptor.forward_bits(gpu(im), kp).cpu().numpy().view(np.uint32)
        _, aux = O.sparse_bad(im, kp.cpu().numpy(), box, thr, binarize=True, soft_binarize=False, return_aux=True)
        assert np.array_equal(bits, O.pack_bits(aux["bits"]))
        perm = tie_canonical_perm(g["kpts" + tag][0], g["kscores" + tag][0], 640)
        diff = np.argwhere(unpack_bits(g["bits" + tag][0][perm], 512) != unpack_bits(bits[0], 512))
        assert len(diff) == 0, (tag, len(diff))          # measured on this fixture: 0 of 262 144 bits
        for kk, pp in diff:
            assert abs(aux["centered"][0, kk, pp]) < 5e-4


def test_batch_consistency_and_determinism(mods):
    """A batch equals its pairs run one by one; two runs are bit-identical (atomics only decide
    the append order of candidates, never the result)."""
    a, b = synth_batch(3000, 3, 120, 160)
    cfg = dict(num_pairs=512, binarize=True, soft_binarize=False, epsilon=0.05, nms_radius=5)
    model = mods["ShiTomasiSparseBADSinkhornMatcher"](max_keypoints=128, **cfg).to(DEV)
    out1 = model(gpu(a), gpu(b))
    out2 = model(gpu(a), gpu(b))
    for x, y in zip(out1, out2):
        assert torch.equal(x, y)
    for i in range(3):
        single = model(gpu(a[i:i + 1]), gpu(b[i:i + 1]))
        for x, y in zip(out1, single):
            assert torch.equal(x[i:i + 1], y)


def test_full_size_batch_properties(mods):
    """BASELINE full size (640x480, K=512), batch of 8: size-independent properties of the output."""
    a, b = synth_batch(4000, 8, 480, 640)
    model = mods["ShiTomasiSparseBADSinkhornMatcher"](
        max_keypoints=512, num_pairs=512, binarize=True, soft_binarize=False, epsilon=0.05, nms_radius=5).to(DEV)
    k1, k2, p = model(gpu(a), gpu(b))
    assert p.shape == (8, 513, 513) and torch.isfinite(p).all()
    # after the final column normalisation column sums are exact marginals; row sums nearly
    np.testing.assert_allclose(p[:, :, :512].sum(1).cpu().numpy(), 1.0, atol=2e-4)
    np.testing.assert_allclose(p[:, :, 512].sum(1).cpu().numpy(), 512.0, rtol=2e-4)
    # 20 iterations at epsilon 0.05 have not converged: row sums are only close to 1 (reference too)
    np.testing.assert_allclose(p[:, :512].sum(2).cpu().numpy(), 1.0, atol=0.15)
    # keypoints: inside the border, sorted by score, distinct
    kk = k1.cpu().numpy()
    assert kk[..., 0].min() >= 7 and kk[..., 0].max() < 473 and kk[..., 1].min() >= 7 and kk[..., 1].max() < 633
    for i in range(8):
        assert len({tuple(x) for x in kk[i]}) == 512
    # image 2 is image 1 shifted by (3,5): mutual matches must recover that shift
    mk1, mk2, sc, valid = mods["MutualNearestNeighborMatcher"](100, 0.1)(p, k1, k2)
    d = (mk2 - mk1)[valid]
    frac = ((d[:, 0] == 3) & (d[:, 1] == 5)).float().mean().item()
    assert valid.float().mean().item() > 0.9 and frac > 0.95


# ------------------------------------------------------------------ K8 orientation + oriented K4
def _ang_diff(a, b):
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    return np.minimum(d, 2 * np.pi - d)


def test_angle_map_and_keypoint_angles(mods):
    from onnx_image_processing_amd.pytorch_model.orientation import AngleEstimator
    g = load_golden("angle_pipeline")
    a, _ = synth_batch(int(g["seed"]), 1, int(g["h"]), int(g["w"]))
    for ps, sg, key in ((15, 2.5, "angle_map"), (9, 1.5, "angle_map_p9")):
        est = AngleEstimator(ps, sg).to(DEV)
        got = est(gpu(a)).cpu().numpy()
        assert _ang_diff(got, O.angle_map(a, ps, sg)).max() < 3e-4          # vs oracle (fp64 accumulation)
        assert _ang_diff(got, g[key]).max() < 3e-4                           # vs the reference's map
        rng = np.random.default_rng(ps)
        kp = np.stack([rng.integers(0, 120, (1, 50)), rng.integers(0, 160, (1, 50))], -1).astype(np.float32)
        kp[0, 0] = (-1, -1)
        th = est.at_keypoints(gpu(a), gpu(kp)).cpu().numpy()
        assert _ang_diff(th, O.sample_nearest(got, kp)).max() < 3e-4         # = the map sampled at the keypoints
    with pytest.raises(ValueError):
        AngleEstimator(14)
    with pytest.raises(ValueError):
        AngleEstimator(15, -1.0)


def test_oriented_bad_vs_oracle(mods):
    box, thr = bad_tables(512)
    a, _ = synth_batch(920, 2, 120, 160)
    rng = np.random.default_rng(6)
    kp = np.stack([rng.integers(0, 120, (2, 80)), rng.integers(0, 160, (2, 80))], -1).astype(np.float32)
    kp[0, 3] = (-1, -1)
    kp[1, :4] = [(0, 0), (119, 159), (3, 150), (110, 2)]
    theta = (rng.random((2, 80)).astype(np.float32) * 2 - 1) * np.float32(np.pi)
    theta[0, :8] = [0.0, np.pi / 2, -np.pi / 2, np.pi, 0.3, -2.0, 1e-3, 3.0]
    amap = np.zeros((2, 1, 120, 160), np.float32)                              # dense-map form of the same angles
    for bi in range(2):
        for j in range(80):
            if kp[bi, j, 0] >= 0:
                amap[bi, 0, int(kp[bi, j, 0]), int(kp[bi, j, 1])] = theta[bi, j]
    amap[0, 0, 0, 0] = theta[0, 3]                                             # invalid keypoint samples (0,0)
    for kw in (dict(binarize=True, soft_binarize=False), dict(binarize=False, normalize_descriptors=False),
               dict(binarize=True, soft_binarize=True)):
        mod = mods["SparseBAD"](512, **kw).to(DEV)
        for ori in (theta, amap):
            # dense-map form: the angle is whatever the map holds at the (clamped) keypoint pixel
            th = theta if ori is theta else O.sample_nearest(amap, kp)
            ref, aux = O.sparse_bad_oriented(a, kp, th, box, thr, return_aux=True, **kw)
            got = mod(gpu(a), gpu(kp), gpu(ori)).cpu().numpy()
            if kw.get("binarize") and not kw.get("soft_binarize", True):
                # cosf/sinf on the GPU vs numpy can move a centre that sits within rounding of x.5
                bits_mismatch(got != 0, aux["bits"], ALLOW["gpu_oriented_vs_oracle"], "oriented hard bits vs oracle")
            else:
                bad = np.abs(got - ref) > (3e-5 if kw.get("normalize_descriptors", True) else 2e-4)
                assert bad.mean() < 5e-4
    mod = mods["SparseBAD"](512, binarize=True, soft_binarize=False).to(DEV)
    bits = mod.forward_bits(gpu(a), gpu(kp), gpu(theta)).cpu().numpy().view(np.uint32)
    d = mod(gpu(a), gpu(kp), gpu(theta)).cpu().numpy()
    assert np.array_equal(unpack_bits(bits, 512), d != 0)


def test_oriented_bad_small_window_equals_full_window(mods):
    """Rotation-aware BAD with the table's reach stated (SparseBAD.max_reach = 22.22 px: a 48 x 48 window per keypoint)
    against the generic 60 x 60 window: identical descriptors and bits for every mode, on keypoints all over the image
    (corners, borders, sub-pixel positions, an invalid one) and angles that put the far pairs on the window's diagonal
    (odd multiples of 45 degrees), for both tables and for float (non-integral) images, which take the fp64 table."""
    from onnx_image_processing_amd import _native as N, ops
    rng = np.random.default_rng(31)
    a, _ = synth_batch(930, 2, 120, 160)
    imgs = [gpu(a), gpu(a + np.float32(0.37))]
    kp = np.stack([rng.integers(0, 120, (2, 96)), rng.integers(0, 160, (2, 96))], -1).astype(np.float32)
    kp[0, :6] = [(0, 0), (119, 159), (0, 159), (119, 0), (60.5, 80.25), (-1, -1)]
    kp[1, :4] = [(23, 23), (24, 135), (96, 24), (95.75, 136.5)]
    theta = (rng.random((2, 96)).astype(np.float32) * 2 - 1) * np.float32(np.pi)
    theta[:, 8:16] = np.float32(np.pi / 4) * np.array([1, 3, 5, 7, -1, -3, -5, -7], np.float32)
    for pairs in (256, 512):
        m = mods["SparseBAD"](pairs, binarize=True, soft_binarize=False).to(DEV)
        assert 22.0 < m.max_reach < 22.5
        for img in imgs:
            for mode, want_bits in ((N.MI_BAD_HARD, True), (N.MI_BAD_SOFT, False), (N.MI_BAD_RAW, False)):
                outs = []
                for reach in (m.max_reach, 0.0):
                    outs.append(ops.sparse_bad_oriented(img, gpu(kp), gpu(theta), m.pair_geom, m.pair_thr, mode, 10.0, True,
                                                        want_desc=True, want_bits=want_bits, max_reach=reach))
                assert torch.equal(outs[0][0], outs[1][0]), (pairs, mode)
                if want_bits:
                    assert torch.equal(outs[0][1], outs[1][1]), (pairs, mode)
    with pytest.raises(RuntimeError):
        ops.sparse_bad_oriented(imgs[0], gpu(kp), gpu(theta), m.pair_geom, m.pair_thr, N.MI_BAD_HARD, 10.0, True, max_reach=-1.0)


def test_pair_entries_equal_two_calls(mods):
    """mi_corner_response_pair / mi_sparse_bad_pair / mi_angle_at_keypoints_pair / mi_sparse_bad_oriented_pair (image1 and
    image2 of a matcher behind ONE launch, two base pointers, ops.ImagePair) against one call per image batch: identical
    outputs for float32 and uint8 frames, blocks 3 and 5, bits and float descriptors, a batch large enough for K1's
    ticket schedule -- and the matchers built on them against their one-call-per-image form (pair_launches = False)."""
    from onnx_image_processing_amd import _native as N, ops
    from onnx_image_processing_amd.pytorch_model.feature_detection import (ShiTomasiAngleSparseBADSinkhornMatcher,
                                                                           ShiTomasiBADSinkhornMatcher)
    from onnx_image_processing_amd.pytorch_model.orientation.angle_estimation import AngleEstimator
    for n, h, w in ((3, 120, 160), (70, 128, 256)):
        a, b = synth_batch(7000 + n, n, h, w)
        for conv in (lambda x: gpu(x), lambda x: gpu(x.astype(np.uint8))):
            ia, ib = conv(a), conv(b)
            pair = ops.ImagePair(ia, ib)
            for bs in (3, 5):
                want = torch.cat([ops.corner_response(ia, bs), ops.corner_response(ib, bs)])
                assert torch.equal(ops.corner_response(pair, bs), want), (n, bs, ia.dtype)
            kp, _ = ops.nms_topk(want.squeeze(1), 3, 64, 0.0, 7)
            kp[0, :3] = torch.tensor([[-1.0, -1.0], [0.0, 0.0], [h - 1.0, w - 1.0]], device=DEV)
            m = mods["SparseBAD"](512, binarize=True, soft_binarize=False).to(DEV)
            for packed in (True, False):
                f = (lambda im, k: m.forward_bits(im, k)) if packed else (lambda im, k: m(im, k))
                assert torch.equal(f(pair, kp), torch.cat([f(ia, kp[:n]), f(ib, kp[n:])])), (n, packed, ia.dtype)
            est = AngleEstimator(15, 2.5).to(DEV)
            theta = est.at_keypoints(pair, kp)
            assert torch.equal(theta, torch.cat([est.at_keypoints(ia, kp[:n]), est.at_keypoints(ib, kp[n:])]))
            for kw in (dict(binarize=True, soft_binarize=False), dict(binarize=False)):
                mo = mods["SparseBAD"](256, **kw).to(DEV)
                got = mo(pair, kp, theta)
                assert torch.equal(got, torch.cat([mo(ia, kp[:n], theta[:n]), mo(ib, kp[n:], theta[n:])])), (n, kw)
            with pytest.raises(RuntimeError):
                mo(pair, kp, torch.zeros((2 * n, 1, h, w), device=DEV))              # a dense angle map: one batch only
    with pytest.raises(RuntimeError):
        ops.ImagePair(gpu(a), gpu(b[:1]))
    # the matchers
    a, b = synth_batch(7100, 2, 120, 160)
    cfg = dict(max_keypoints=96, num_pairs=512, binarize=True, soft_binarize=False, epsilon=0.05, nms_radius=3)
    for cls in (mods["ShiTomasiSparseBADSinkhornMatcher"], ShiTomasiAngleSparseBADSinkhornMatcher, ShiTomasiBADSinkhornMatcher):
        model = cls(**cfg).to(DEV)
        assert model.pair_launches if hasattr(model, "pair_launches") else True
        got = model(gpu(a), gpu(b))
        model.pair_launches = False
        for x, y in zip(got, model(gpu(a), gpu(b))):
            assert torch.equal(x, y), cls.__name__


def test_oriented_bad_fast_kernels_equal_the_generic_kernel(mods):
    """The matchers' rotation-aware BAD kernel (all loads up front, straight-line pair phase, fp32 threshold test with
    the product's exact residual; packed bits or float descriptors) against the generic kernel (debug key 13 = 1):
    identical bits / floats for 256 and 512 pairs, every mode, both windows (48 with the table's reach stated, 60
    without), per-keypoint angles and a dense angle map, keypoints on corners and borders, sub-pixel and invalid ones,
    an image with NON-uint8-valued regions (those windows go to the fp64 rest kernel) and one that has none, a -0.0
    pixel, and a threshold table with huge, infinite and NaN entries (the whole launch then takes the fp64 path)."""
    from onnx_image_processing_amd import _native as N, ops
    rng = np.random.default_rng(77)
    a, _ = synth_batch(940, 3, 150, 200)
    b = a.copy()
    b[0, 0, 40:90, 50:120] += np.float32(0.5)
    b[1, 0, 10, 10] = np.float32(-0.0)
    b[2, 0, 100:, :] *= np.float32(1.001)
    kn = 160
    kp = np.stack([rng.integers(0, 150, (3, kn)), rng.integers(0, 200, (3, kn))], -1).astype(np.float32)
    kp[0, :8] = [(0, 0), (149, 199), (0, 199), (149, 0), (60.5, 80.25), (-1, -1), (23, 23), (24, 176)]
    kp[1, :4] = [(126, 24), (125.75, 175.5), (10, 10), (11, 11)]
    theta = (rng.random((3, kn)).astype(np.float32) * 2 - 1) * np.float32(np.pi)
    theta[:, 8:16] = np.float32(np.pi / 4) * np.array([1, 3, 5, 7, -1, -3, -5, -7], np.float32)
    theta[2, 20] = np.float32(1e6)                                            # sine / cosine with the big-argument reduction
    amap = (rng.random((3, 1, 150, 200)).astype(np.float32) * 2 - 1) * np.float32(np.pi)
    for pairs in (256, 512):
        m = mods["SparseBAD"](pairs, binarize=True, soft_binarize=False).to(DEV)
        wild = m.pair_thr.clone()
        wild[3], wild[70], wild[130], wild[200] = 3e38, float("inf"), float("nan"), -2e37
        for img in (gpu(a), gpu(b)):
            for ori in (gpu(theta), gpu(amap)):
                for thr in (m.pair_thr, wild):
                    for reach in (m.max_reach, 0.0):
                        for mode, bits_only in ((N.MI_BAD_HARD, True), (N.MI_BAD_HARD, False), (N.MI_BAD_SOFT, False),
                                                (N.MI_BAD_RAW, False)):
                            for norm in (True, False):
                                def run():
                                    return ops.sparse_bad_oriented(img, gpu(kp), ori, m.pair_geom, thr, mode, 10.0, norm,
                                                                   want_desc=not bits_only, want_bits=bits_only,
                                                                   max_reach=reach)
                                fast = run()
                                with N.debug_library() as lib:
                                    lib.mi_debug_set(13, 1)
                                    want = run()
                                what = (pairs, mode, bits_only, norm, reach, thr is wild)
                                if bits_only:
                                    assert torch.equal(fast[1], want[1]), what
                                else:                                          # NaN thresholds: NaN descriptors, same places
                                    assert torch.equal(torch.nan_to_num(fast[0], nan=7.0), torch.nan_to_num(want[0], nan=7.0)), what


@pytest.mark.parametrize("name", ["hard", "soft"])
def test_angle_pipeline_vs_oracle_and_golden(mods, name):
    from onnx_image_processing_amd.pytorch_model.feature_detection import ShiTomasiAngleSparseBADSinkhornMatcher
    g = load_golden("angle_pipeline")
    a, b = synth_batch(int(g["seed"]), 1, int(g["h"]), int(g["w"]))
    cfg = cfg_of(g, name + "_cfg")
    model = ShiTomasiAngleSparseBADSinkhornMatcher(**cfg).to(DEV)
    k1, k2, p = [t.cpu().numpy() for t in model(gpu(a), gpu(b))]
    box, thr = bad_tables(cfg["num_pairs"])
    kw = {k: v for k, v in cfg.items() if k not in ("num_pairs", "max_keypoints")}
    o1, o2, op = O.match_pair_angle(a, b, box, thr, cfg["max_keypoints"], **kw)
    # block 5 scores are tolerance-only (sums above 2^24): the keypoint SETS must agree, order may not
    if cfg["block_size"] == 3:
        assert np.array_equal(k1, o1) and np.array_equal(k2, o2)
    assert {tuple(x) for x in k1[0]} == {tuple(x) for x in o1[0]} == {tuple(x) for x in g[name + "_k1"][0]}
    assert {tuple(x) for x in k2[0]} == {tuple(x) for x in o2[0]} == {tuple(x) for x in g[name + "_k2"][0]}
    if np.array_equal(k1, o1) and np.array_equal(k2, o2):
        ok, worst = p_close(p, op, atol=2e-4)
        assert ok, worst


def test_filters_model_reference_smoke_config(mods):
    """reference test_filters_pytorch.py:9-57 (shapes/dtypes, filters on and off), plus values vs oracle."""
    from onnx_image_processing_amd.pytorch_model.feature_detection import (
        ShiTomasiAngleSparseBADDetector, ShiTomasiAngleSparseBADSinkhornMatcherWithFilters)
    g = load_golden("angle_pipeline")
    a, b = synth_batch(int(g["filt_seed"]), 1, 240, 320)
    cfg = cfg_of(g, "filt_cfg")
    model = ShiTomasiAngleSparseBADSinkhornMatcherWithFilters(**cfg).to(DEV)
    k1, k2, p, valid = model(gpu(a), gpu(b))
    assert k1.shape == (1, 128, 2) and p.shape == (1, 129, 129) and valid.shape == (1, 128) and valid.dtype == torch.bool
    assert {tuple(x) for x in k1[0].cpu().numpy()} == {tuple(x) for x in g["filt_k1"][0]}
    off = ShiTomasiAngleSparseBADSinkhornMatcherWithFilters(max_keypoints=128, ratio_threshold=None, dustbin_margin=None,
                                                            sinkhorn_iterations=10, num_pairs=256).to(DEV)
    assert bool(off(gpu(a), gpu(b))[3].all())
    if np.array_equal(k1.cpu().numpy(), g["filt_k1"]) and np.array_equal(k2.cpu().numpy(), g["filt_k2"]):
        assert (valid.cpu().numpy() == g["filt_valid"]).mean() >= 0.99
    det = ShiTomasiAngleSparseBADDetector(max_keypoints=40, num_pairs=256, binarize=True, soft_binarize=False).to(DEV)
    a0, _ = synth_batch(int(g["seed"]), 1, int(g["h"]), int(g["w"]))
    dk, ds, dd = det(gpu(a0))
    assert {tuple(x) for x in dk[0].cpu().numpy()} == {tuple(x) for x in g["det_k"][0]}
    order = {tuple(x): i for i, x in enumerate(g["det_k"][0])}
    idx = [order[tuple(x)] for x in dk[0].cpu().numpy()]
    bits_mismatch(dd[0].cpu().numpy() != 0, g["det_d"][0][idx] != 0, ALLOW["gpu_angle_detector_vs_reference"],
                  "angle detector hard bits vs reference")


# ------------------------------------------------------------------ dense BAD variant (config 3 semantics)
def test_dense_bad_map_and_gathers(mods):
    from onnx_image_processing_amd.pytorch_model.descriptor.bad import (
        BADDescriptor, extract_descriptors_at_keypoints, extract_descriptors_at_keypoints_subpixel)
    g = load_golden("dense_bad")
    small = synth_image(int(g["seed"]), 20, 28)[None, None].astype(np.float32)
    got = BADDescriptor(256).to(DEV)(gpu(small)).cpu().numpy()
    np.testing.assert_allclose(got, g["raw256"], rtol=0, atol=1e-4)
    hard = BADDescriptor(512, binarize=True, soft_binarize=False).to(DEV)(gpu(small)).cpu().numpy()
    assert np.array_equal(np.packbits(hard != 0), g["hard512"])
    soft = BADDescriptor(256, binarize=True, soft_binarize=True, temperature=3.0).to(DEV)(gpu(small)).cpu().numpy()
    np.testing.assert_allclose(soft[:, ::16], g["soft256"], rtol=0, atol=5e-5)
    # a larger, non-tile-aligned image against the oracle, and map == sparse descriptors at those pixels
    img = np.stack([synth_image(3300 + i, 45, 70) for i in range(2)])[:, None].astype(np.float32)
    box, thr = bad_tables(256)
    dense = BADDescriptor(256, binarize=True, soft_binarize=False).to(DEV)
    dmap = dense(gpu(img))
    assert np.array_equal(dmap.cpu().numpy(), O.bad_dense(img, box, thr, binarize=True, soft_binarize=False))
    rng = np.random.default_rng(8)
    kp = np.stack([rng.integers(0, 45, (2, 60)), rng.integers(0, 70, (2, 60))], -1).astype(np.float32)
    at = dense.at_keypoints(gpu(img), gpu(kp))
    assert torch.equal(at, extract_descriptors_at_keypoints(dmap, gpu(kp)))
    assert np.array_equal(extract_descriptors_at_keypoints(gpu(g["gather_map"]), gpu(g["gather_ki"])).cpu().numpy(),
                          g["gather_nearest"])
    np.testing.assert_allclose(
        extract_descriptors_at_keypoints_subpixel(gpu(g["gather_map"]), gpu(g["gather_kf"])).cpu().numpy(),
        g["gather_bilinear"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("tag", ["m", "s"])
def test_dense_variant_matcher(mods, tag):
    from onnx_image_processing_amd.pytorch_model.feature_detection import ShiTomasiBADSinkhornMatcher
    g = load_golden("dense_bad")
    a, b = synth_batch(int(g["m_seed"]), 1, 120, 160)
    cfg = cfg_of(g, tag + "_cfg")
    k1, k2, p = [t.cpu().numpy() for t in ShiTomasiBADSinkhornMatcher(**cfg).to(DEV)(gpu(a), gpu(b))]
    assert np.array_equal(k1, g[tag + "_k1"]) and np.array_equal(k2, g[tag + "_k2"])       # no border margin here
    ok, worst = p_close(p, g[tag + "_P"])
    assert ok, worst


def test_config3_1080p_k1024_batch(mods):
    """BASELINE config 3 sizes: 1080x1920, K=1024, batch of pairs, both the sparse pipeline and the
    dense-variant semantics; size-independent properties + agreement between the two on interior keypoints."""
    from onnx_image_processing_amd.pytorch_model.feature_detection import ShiTomasiBADSinkhornMatcher
    a, b = synth_batch(4100, 2, 1080, 1920)
    cfg = dict(num_pairs=512, binarize=True, soft_binarize=False, epsilon=0.05, nms_radius=5)
    sparse = mods["ShiTomasiSparseBADSinkhornMatcher"](max_keypoints=1024, **cfg).to(DEV)
    dense = ShiTomasiBADSinkhornMatcher(max_keypoints=1024, **cfg).to(DEV)
    for model, margin in ((sparse, 7), (dense, 0)):
        k1, k2, p = model(gpu(a), gpu(b))
        assert p.shape == (2, 1025, 1025) and torch.isfinite(p).all()
        np.testing.assert_allclose(p[:, :, :1024].sum(1).cpu().numpy(), 1.0, atol=3e-4)
        kk = k1.cpu().numpy()
        assert kk[..., 0].min() >= margin and kk[..., 0].max() < 1080 - margin
        mk1, mk2, sc, valid = mods["MutualNearestNeighborMatcher"](100, 0.1)(p, k1, k2)
        d = (mk2 - mk1)[valid]
        assert valid.float().mean().item() > 0.9 and ((d[:, 0] == 3) & (d[:, 1] == 5)).float().mean().item() > 0.95


# ------------------------------------------------------------------ AKAZE (config 4)
def test_akaze_detector_vs_oracle_and_golden(mods):
    from onnx_image_processing_amd.pytorch_model.detector import AKAZE
    g = load_golden("akaze_pipeline")
    img = synth_image(int(g["seed"]), int(g["h"]), int(g["w"]))[None, None].astype(np.float32)
    for tag, x in (("u8", img), ("unit", img / np.float32(255.0))):
        m = AKAZE().to(DEV)
        d1 = m.diffusion_layers[0](gpu(x))
        assert np.array_equal(d1.cpu().numpy(), g[tag + "_diffused1"])         # 3x3 stencils: bit-exact
        assert np.array_equal(m.detector(d1).cpu().numpy(), g[tag + "_scores1"])
        sc, ori = [t.cpu().numpy() for t in m(gpu(x))]
        osc, oori = O.akaze(x)
        assert np.array_equal(sc, osc)
        np.testing.assert_allclose(sc, g[tag + "_scores"], rtol=1e-6, atol=0)
        assert _ang_diff(ori, oori).max() < 3e-4 and _ang_diff(ori, g[tag + "_orientations"]).max() < 3e-4
    alt = AKAZE(num_scales=2, diffusion_iterations=2, kappa=0.2, threshold=0.0005, nms_size=3,
                orientation_patch_size=9, orientation_sigma=1.5).to(DEV)
    sc, ori = [t.cpu().numpy() for t in alt(gpu(img / np.float32(255.0)))]
    assert np.array_equal(sc, g["alt_scores"]) and _ang_diff(ori, g["alt_orientations"]).max() < 3e-4
    # batch > 1, sizes that are not tile multiples, one-scale / one-iteration corner cases
    odd = np.stack([synth_image(3400 + i, 37, 53) for i in range(3)])[:, None].astype(np.float32)
    for kw in (dict(num_scales=1, diffusion_iterations=1), dict(num_scales=4, diffusion_iterations=2, nms_size=7)):
        m = AKAZE(**kw).to(DEV)
        sc, ori = [t.cpu().numpy() for t in m(gpu(odd))]
        osc, oori = O.akaze(odd, kw["num_scales"], kw["diffusion_iterations"], nms_size=kw.get("nms_size", 5))
        assert np.array_equal(sc, osc) and _ang_diff(ori, oori).max() < 3e-4
        # the keypoint-only orientation path == the combined map sampled at the keypoints
        s2, ss, ims = m.detect(gpu(odd))
        assert np.array_equal(s2.cpu().numpy(), sc)
        rng = np.random.default_rng(3)
        kp = np.stack([rng.integers(0, 37, (3, 40)), rng.integers(0, 53, (3, 40))], -1).astype(np.float32)
        kp[0, 0] = (-1, -1)
        th = m.orientation_at_keypoints(ss, ims, gpu(kp)).cpu().numpy()
        assert _ang_diff(th, O.sample_nearest(ori, kp)).max() < 3e-4
    assert set(AKAZE().state_dict()) == {
        "diffusion_layers.0.sobel_xy", "diffusion_layers.0.sobel_xy_grouped", "diffusion_layers.1.sobel_xy",
        "diffusion_layers.1.sobel_xy_grouped", "diffusion_layers.2.sobel_xy", "diffusion_layers.2.sobel_xy_grouped",
        "detector.hessian_kernels", "orientation_estimator.moment_kernels"}


@pytest.mark.parametrize("key,div", [("soft_u8", 1.0), ("hard_u8", 1.0), ("soft_unit", 255.0)])
def test_akaze_pipeline_vs_oracle_and_golden(mods, key, div):
    from onnx_image_processing_amd.pytorch_model.feature_detection import AKAZESparseBADSinkhornMatcher
    g = load_golden("akaze_pipeline")
    a, b = synth_batch(int(g["pair_seed"]), 1, 120, 160)
    a, b = a / np.float32(div), b / np.float32(div)
    cfg = cfg_of(g, key + "_cfg")
    model = AKAZESparseBADSinkhornMatcher(**cfg).to(DEV)
    k1, k2, p = [t.cpu().numpy() for t in model(gpu(a), gpu(b))]
    assert p.shape == (1, cfg["max_keypoints"] + 1, cfg["max_keypoints"] + 1)
    box, thr = bad_tables(cfg.get("num_pairs", 256))
    kw = {k: v for k, v in cfg.items() if k not in ("num_pairs", "max_keypoints")}
    o1, o2, op = O.match_pair_akaze(a, b, box, thr, cfg["max_keypoints"], **kw)
    assert np.array_equal(k1, o1) and np.array_equal(k2, o2)                    # scores are bit-exact
    assert np.array_equal(k1, g[key + "_k1"]) and np.array_equal(k2, g[key + "_k2"])
    # angles are tolerance-only: a rotated box centre within rounding of x.5 may move one pixel
    atol = 5e-3 if key == "hard_u8" else 5e-4
    ok, worst = p_close(p, op, atol=atol)
    assert ok, worst
    ok, worst = p_close(p, g[key + "_P"], atol=atol)
    assert ok, worst


@pytest.mark.parametrize("shape", [(2, 37, 53), (1, 96, 128), (3, 70, 200), (1, 480, 640), (2, 131, 258), (1, 33, 2)])
def test_akaze_fused_scale_equals_step_kernels(mods, shape):
    """mi_akaze_scale -- the streaming rolling-window kernel (csrc/akaze_stream.hip: even widths, nms_size 3 / 5) and the
    LDS-tile kernel (odd widths, nms_size 7, or debug key 12) -- against the per-step kernels they replace, bit for bit:
    every fused (iterations, nms_size) instance, the unfused fall-back, image sizes that are not tile / strip multiples
    (several strips, several row chunks, a 2-pixel-wide image), unit-range and uint8-range images."""
    from onnx_image_processing_amd import _native as N, ops
    n, h, w = shape
    img = np.stack([synth_image(3500 + i, h, w) for i in range(n)])[:, None].astype(np.float32)
    for scale, kappa, thr in ((1.0, 0.05, 0.001), (1.0 / 255.0, 0.05, 1e-5), (1.0, 7.5, 3.0)):
        x = gpu(img * np.float32(scale))
        for iters, nms in ((1, 3), (2, 5), (3, 5), (3, 7), (3, 3), (4, 5), (2, 9), (1, 5), (2, 3)):
            if (h, w) == (480, 640) and (iters, nms) not in ((3, 5), (4, 5)):
                continue
            assert bool(N.load().mi_akaze_scale_fused(iters, nms)) == (iters <= 3 and nms <= 7)
            want_l = ops.akaze_diffuse(x, iters, kappa, 0.25)
            want_s = ops.akaze_hessian_scores(want_l, thr, nms)
            got_l, got_s = ops.akaze_scale(x, iters, kappa, 0.25, thr, nms)
            assert torch.equal(got_l, want_l) and torch.equal(got_s, want_s), (scale, iters, nms)
            assert int((got_s > 0).sum()) > 0 or scale != 1.0 or min(h, w) < 16
            with N.debug_library() as lib:            # the LDS-tile kernel where the streaming one runs by default
                lib.mi_debug_set(12, 1)
                tile_l, tile_s = ops.akaze_scale(x, iters, kappa, 0.25, thr, nms)
            assert torch.equal(tile_l, want_l) and torch.equal(tile_s, want_s), (scale, iters, nms, "tile")


@pytest.mark.parametrize("shape,scales", [((2, 120, 160), 3), ((1, 37, 53), 2), ((3, 70, 200), 4), ((1, 96, 128), 1)])
def test_akaze_select_in_the_last_scale_equals_combine(mods, shape, scales):
    """AKAZE.detect_select (mi_akaze_scale_select: the max over scales and the set of scales reaching it, written by the
    last scale's launch) against detect() + mi_akaze_combine on the stacked maps: same scores bit for bit, attain = the
    equality bits, and the keypoint orientations computed from either are identical -- streaming form, general-
    parameter form (odd width) and the LDS-tile form behind the debug hook."""
    from onnx_image_processing_amd import _native as N
    from onnx_image_processing_amd.pytorch_model.detector import AKAZE
    n, h, w = shape
    img = gpu(np.stack([synth_image(3600 + i, h, w) for i in range(n)])[:, None].astype(np.float32))
    m = AKAZE(num_scales=scales, diffusion_iterations=2 if scales == 4 else 3).to(DEV)
    scores, ss, ims = m.detect(img)
    rng = np.random.default_rng(5)
    kp = np.stack([rng.integers(0, h, (n, 64)), rng.integers(0, w, (n, 64))], -1).astype(np.float32)
    top = torch.topk(scores.flatten(1), 32).indices.cpu().numpy()
    kp[:, :32, 0], kp[:, :32, 1] = top // w, top % w                     # real maxima among the query points
    kp[0, 63] = (-1, -1)
    want_theta = m.orientation_at_keypoints(ss, ims, gpu(kp))
    want_attain = sum(((ss[s] == scores).to(torch.int32) << s) for s in range(scales)).to(torch.uint8)

    def check():
        s2, attain, ims2 = m.detect_select(img)
        assert torch.equal(s2, scores) and attain.dtype == torch.uint8 and torch.equal(attain, want_attain)
        assert tuple(ims2.shape) == (scales,) + tuple(img.shape) and all(torch.equal(a, b) for a, b in zip(ims, ims2))
        # one launch (moments only for the attaining scales) == per-scale moments + selection, either selection source
        assert torch.equal(m.orientation_at_keypoints(attain, ims2, gpu(kp)), want_theta)
        assert torch.equal(m.orientation_at_keypoints(attain, list(ims2), gpu(kp)), want_theta)

    def check_pair():                                    # two batches behind one launch per scale == each batch on its own
        img_b = torch.flip(img, dims=[3]).contiguous()
        sa, aa, ia = m.detect_select(img)
        sb, ab, ib = m.detect_select(img_b)
        s2, a2, i2 = m.detect_select(img, img_b)
        assert torch.equal(s2, torch.cat([sa, sb])) and torch.equal(a2, torch.cat([aa, ab]))
        assert torch.equal(i2, torch.cat([ia, ib], dim=1))

    check()
    check_pair()
    with N.debug_library() as lib:
        lib.mi_debug_set(12, 1)
        check()
        check_pair()


def test_akaze_fast_division_is_exact(mods):
    """The fused AKAZE kernel's arithmetic helpers (csrc/akaze_math.h) against the IEEE operators, EXHAUSTIVELY over the
    operand ranges of the diffusion step: ak_sqrt on [1e-8, 2^24] and the fp32-pipe form the kernel uses
    (ak_sqrt_fp<1>: v_rsq_f32 + one exact-residual correction; K1's sqrt_rn is the same sequence) on [1e-10, 2^126], x / kappa through the precomputed reciprocal
    (Markstein's 3-instruction form) for several kappa on [2^-30, 2^24], 1 / d on [1, 2^40), and the general division.
    ~1.8e9 evaluations; zero differing bit patterns allowed."""
    import struct
    from onnx_image_processing_amd import _native as N

    def bits(x):
        return struct.unpack("<I", struct.pack("<f", x))[0]

    cases = [(0, 0.05, 1e-8, 2.0 ** 24), (4, 0.05, 1e-10, 2.0 ** 126), (2, 0.05, 1.0, 2.0 ** 40)]
    cases += [(1, kappa, 2.0 ** -30, 2.0 ** 24) for kappa in (0.05, 0.03, 0.1, 0.7, 1.0, 3.0, 1e-3)]
    cases += [(3, 0.05, 2.0 ** -30, 2.0 ** 24)]
    with N.debug_library() as lib:
        for which, kappa, lo, hi in cases:
            bad = torch.zeros(1, dtype=torch.int64, device=DEV)
            first = torch.full((1,), -1, dtype=torch.int32, device=DEV)
            N.check(lib.mi_debug_akaze_math_check(which, kappa, bits(lo), bits(hi), bad.data_ptr(), first.data_ptr(),
                                                  N.stream_ptr()), "mi_debug_akaze_math_check")
            assert int(bad.item()) == 0, (which, kappa, int(bad.item()), hex(int(first.item()) & 0xFFFFFFFF))


def test_akaze_c4_480x640_k512_golden(mods):
    """BASELINE configs[3] at its own size, AKAZE export-CLI values (what `bench.py --workload c4` times): keypoints
    exact, P within 1e-4 of the recorded reference output, same MNN match set."""
    from onnx_image_processing_amd.pytorch_model.feature_detection import AKAZESparseBADSinkhornMatcher
    g = load_golden("akaze_c4_480x640_k512")
    cfg = cfg_of(g)
    a, b = synth_batch(int(g["seed"]), 1, int(g["h"]), int(g["w"]))
    model = AKAZESparseBADSinkhornMatcher(max_keypoints=int(g["k"]), **cfg).to(DEV)
    k1, k2, p = model(gpu(a), gpu(b))
    assert np.array_equal(k1.cpu().numpy(), g["k1"]) and np.array_equal(k2.cpu().numpy(), g["k2"])
    ok, worst = p_close(p.cpu().numpy(), g["P"])                  # 1e-4 (VERDICT r1 asked 5e-4; measured 1.3e-6)
    assert ok, worst
    if os.environ.get("MI_REPORT"):
        print(f"[akaze c4] worst |dP| / 1e-4 = {worst:.3g}")
    mcfg = cfg_of(g, "mnn_cfg")
    mk1, mk2, sc, valid = [t.cpu().numpy() for t in mods["MutualNearestNeighborMatcher"](mcfg["max_matches"], mcfg["threshold"])(p, k1, k2)]
    # the same match set; the only difference allowed is a swap of scores closer than 1e-4 at the max_matches cut,
    # and that is checked match by match (helpers.check_match_sets), not by a count
    kind = check_match_sets(match_dict(mk1[0], mk2[0], sc[0], valid[0]),
                            match_dict(g["mk1"][0], g["mk2"][0], g["mscores"][0], g["mvalid"][0]), mcfg["max_matches"])
    if os.environ.get("MI_REPORT"):
        print(f"[akaze c4] match set vs reference: {kind}")
    # batch of 3 identical pairs == the single pair (the fused per-scale kernels tile every image the same way)
    kb1, kb2, pb = model(gpu(np.repeat(a, 3, 0)), gpu(np.repeat(b, 3, 0)))
    for i in range(3):
        assert torch.equal(kb1[i], k1[0]) and torch.equal(kb2[i], k2[0]) and torch.equal(pb[i], p[0])


def test_c3_pair_1080p_k1024_golden(mods):
    """BASELINE configs[2] size against the recorded reference run (not only properties): keypoints after tie
    canonicalisation, packed bits, P through maxima / argmaxima / dustbins / marginals / sample rows, MNN matches."""
    from test_oracle_golden import check_c3_against_fixture
    g = load_golden("c3_pair_1080x1920_k1024")
    a, b = synth_batch(int(g["seed"]), 1, int(g["h"]), int(g["w"]))
    cfg = cfg_of(g)
    model = mods["ShiTomasiSparseBADSinkhornMatcher"](max_keypoints=int(g["k"]), **cfg).to(DEV)
    k1, k2, p = model(gpu(a), gpu(b))
    bits = [model.descriptor.forward_bits(gpu(im), kp).cpu().numpy() for im, kp in ((a, k1), (b, k2))]
    check_c3_against_fixture(g, k1.cpu().numpy(), k2.cpu().numpy(), p.cpu().numpy(), bits[0], bits[1])
    mcfg = cfg_of(g, "mnn_cfg")
    wrap = mods["MatchExtractionWrapper"](model, max_matches=mcfg["max_matches"], match_threshold=mcfg["threshold"])
    mk1, mk2, sc, valid = [t.cpu().numpy() for t in wrap(gpu(a), gpu(b))]
    assert np.array_equal(valid, g["mvalid"])
    want = {(tuple(x), tuple(y)) for x, y, v in zip(g["mk1"][0], g["mk2"][0], g["mvalid"][0]) if v}
    assert {(tuple(x), tuple(y)) for x, y, v in zip(mk1[0], mk2[0], valid[0]) if v} == want


def test_akaze_argument_checks(mods):
    from onnx_image_processing_amd import ops
    x = gpu(np.zeros((1, 1, 16, 16), np.float32))
    with pytest.raises(RuntimeError):
        ops.akaze_hessian_scores(x, 0.001, 4)            # even NMS window
    with pytest.raises(RuntimeError):
        ops.akaze_diffuse(x, 1, 0.0)                     # kappa must be positive
    with pytest.raises(RuntimeError):
        ops.akaze_diffuse(x.cpu(), 1, 0.05)              # no CPU path


# ------------------------------------------------------------------ matches straight from the duals
@pytest.mark.parametrize("n,m", [(512, 512), (40, 56), (300, 77), (33, 1000), (700, 520), (1024, 1024), (96, 1024)])
def test_mnn_from_duals_equals_two_step(mods, n, m):
    """mi_mnn_from_duals[_dots] == mi_sinkhorn(P) + mi_mnn_extract, bit for bit (incl. indices)."""
    from onnx_image_processing_amd import ops
    rng = np.random.default_rng(n * 7 + m)
    words = 8
    b1 = torch.from_numpy(rng.integers(-2**31, 2**31, (2, n, words), dtype=np.int64).astype(np.int32)).to(DEV)
    b2 = torch.from_numpy(rng.integers(-2**31, 2**31, (2, m, words), dtype=np.int64).astype(np.int32)).to(DEV)
    b2[0, : min(n, m) // 2] = b1[0, : min(n, m) // 2]                           # true matches + exact ties
    k1 = gpu(rng.integers(0, 400, (2, n, 2)).astype(np.float32))
    k2 = gpu(rng.integers(0, 400, (2, m, 2)).astype(np.float32))
    nvalid = 0
    for eps in (0.05, 1.0):
        # fp32-Z form
        z, pitch = ops.cost_logscores_bits(b1, b2, True, eps)
        p, u, v = ops.sinkhorn(z, m, pitch, -1.0 / eps, 7, return_duals=True)
        two = ops.mnn_extract(p, k1, k2, 50, 0.05, return_indices=True)
        one = ops.mnn_from_duals(z, m, pitch, u, v, k1, k2, 50, 0.05, return_indices=True)
        for a, c in zip(one, two):
            assert torch.equal(a, c)
        # uint16 dot-product form
        p, u, v, state = ops.sinkhorn_bits(b1, b2, True, eps, 1.0, 7, return_state=True)
        two = ops.mnn_extract(p, k1, k2, 50, 0.05, return_indices=True)
        one = ops.mnn_from_duals_dots(state, m, eps, u, v, k1, k2, 50, 0.05, return_indices=True)
        for a, c in zip(one, two):
            assert torch.equal(a, c)
        nvalid += int(two[3].sum())
    assert nvalid > 0


@pytest.mark.parametrize("name", ["c2_pair_480x640_k512", "small_soft_l1_96x128_k32", "ragged_120x160_k96"])
def test_wrapper_fused_equals_two_step(mods, name):
    g = load_golden(name)
    cfg = cfg_of(g)
    a, b = _images(g)
    model = mods["ShiTomasiSparseBADSinkhornMatcher"](max_keypoints=int(g["k"]), **cfg).to(DEV)
    wrap = mods["MatchExtractionWrapper"](model, max_matches=100, match_threshold=0.1)
    fused = wrap(gpu(a), gpu(b))
    wrap.fuse_extraction = False
    plain = wrap(gpu(a), gpu(b))
    for x, y in zip(fused, plain):
        assert torch.equal(x, y)
    # and with the fp32-Z Sinkhorn form instead of the dot-product form
    model.matcher.use_dot_storage = False
    wrap.fuse_extraction = True
    fused_z = wrap(gpu(a), gpu(b))
    wrap.fuse_extraction = False
    for x, y in zip(fused_z, wrap(gpu(a), gpu(b))):
        assert torch.equal(x, y)
    assert torch.equal(fused_z[3], fused[3])                                     # same match validity either way


# ------------------------------------------------------------------ essential-matrix head
def _e_close(e, ref, tol=1e-4):
    return np.abs(np.asarray(e, np.float64) - np.asarray(ref, np.float64)).max() <= tol * max(1.0, np.abs(ref).max())


def test_essential_matrix_estimator_vs_oracle_and_golden(mods):
    from onnx_image_processing_amd.pytorch_model.geometry import EssentialMatrixEstimator
    g = load_golden("essential_matrix")
    kg = torch.from_numpy(g["grid_K"])
    est = EssentialMatrixEstimator(K=kg, image_shape=(32, 32)).to(DEV)
    est5 = EssentialMatrixEstimator(K=kg, image_shape=(32, 32), top_k=5, n_iter=12, n_iter_manifold=4).to(DEV)
    assert set(est.state_dict()) == {"K", "K_inv", "pixel_coords", "pixel_coords_n"}
    for i in range(3):
        p = g[f"grid{i}_P"]
        e = est(gpu(p)).cpu().numpy()
        assert e.shape == (3, 3)
        assert _e_close(e, g[f"grid{i}_E"]) and _e_close(e, O.essential_matrix_grid(p, g["grid_K"]))
        assert _e_close(est5(gpu(p)).cpu().numpy(), g[f"grid{i}_E5"])
    # batched (extension) == one by one; ties at the k-th value and an all-zero row / column
    rng = np.random.default_rng(12)
    p = (rng.random((3, 130, 97)).astype(np.float32)) ** 4
    p[0, 5, :] = 0.0
    p[0, :, 7] = 0.0
    p[1, 9, 3:9] = 0.5                                                       # six equal values in one row
    eb = est(gpu(p)).cpu().numpy()
    for b in range(3):
        assert _e_close(eb[b], O.essential_matrix_grid(p[b], g["grid_K"]))
        assert np.array_equal(eb[b], est(gpu(p[b])).cpu().numpy())          # deterministic
    with pytest.raises(RuntimeError):
        est(gpu(np.zeros((1100, 20), np.float32)))                           # more features than grid points / N > 1024
    with pytest.raises(RuntimeError):
        EssentialMatrixEstimator(K=kg, image_shape=(32, 32), top_k=9).to(DEV)(gpu(p[0]))


def test_essential_matrix_banded_form_equals_dense_form(mods):
    """mi_essential_matrix with a workspace (one pass over P spread over the chip + one workgroup per pair on the sparse
    weights) against the single-launch dense form and the oracle: random matrices with a few confident entries per row,
    top_k 1..4, n != m, m > 512 (two column chunks per lane), validity masks, an all-zero row and column, exact ties inside
    the top k (six equal values in a row), and massive ties that overflow a row's candidate list and a column's
    contribution list (the pair then goes through the dense front inside the sparse kernel)."""
    from onnx_image_processing_amd import ops
    rng = np.random.default_rng(77)

    def pts(bn, cnt):
        return gpu((rng.random((bn, cnt, 2)).astype(np.float32) - 0.5) * 1.2)

    def run(pm, top_k, v1=None, v2=None):
        bn, n1, m1 = pm.shape
        q1, q2 = pts(bn, n1 - 1), pts(bn, m1 - 1)
        outs = [ops.essential_matrix(gpu(pm), q1, q2, v1, v2, top_k, 30, 10, banded=f).cpu().numpy() for f in (True, False)]
        for bi in range(bn):
            assert _e_close(outs[0][bi], outs[1][bi], tol=2e-4), (pm.shape, top_k, bi, np.abs(outs[0][bi] - outs[1][bi]).max())
        again = ops.essential_matrix(gpu(pm), q1, q2, v1, v2, top_k, 30, 10, banded=True).cpu().numpy()
        assert np.array_equal(again, outs[0])                                  # deterministic
        return outs[0]

    for shape in ((3, 513, 513), (2, 201, 141), (2, 65, 97), (2, 300, 1025), (1, 1025, 1025), (1, 34, 20)):
        pm = rng.random(shape).astype(np.float32) ** 6
        for top_k in (1, 2, 3, 4):
            run(pm, top_k)
    pm = rng.random((3, 257, 301)).astype(np.float32) ** 4
    pm[0, 5, :] = 0.0
    pm[0, :, 7] = 0.0
    pm[1, 9, 3:9] = 0.5                                                           # six equal values in one row
    pm[2, 20:40, 11] = 0.75                                                       # twenty equal values in one column
    run(pm, 3)
    v1 = gpu(rng.random((3, 256)) > 0.2)
    v2 = gpu(rng.random((3, 300)) > 0.2)
    run(pm, 3, v1, v2)
    ties = np.full((2, 129, 129), 0.3, np.float32)                                # every entry tied: candidate lists overflow
    ties[1] = rng.random((129, 129)).astype(np.float32) ** 5                      # (pair 1 stays on the sparse path)
    run(ties, 3)
    col_over = rng.random((1, 200, 90)).astype(np.float32) * 0.005                # nothing above 0.01 ...
    col_over[0, :40, 17] = np.linspace(0.5, 0.9, 40, dtype=np.float32)            # ... but one column with 40 row-winners
    col_over[0, :40, 18] = 0.45
    run(col_over, 3)
    # the banded form against the oracle directly (grid form of the estimator)
    from onnx_image_processing_amd.pytorch_model.geometry import EssentialMatrixEstimator
    g = load_golden("essential_matrix")
    est = EssentialMatrixEstimator(K=torch.from_numpy(g["grid_K"]), image_shape=(32, 32)).to(DEV)
    for i in range(3):
        assert _e_close(est(gpu(g[f"grid{i}_P"])).cpu().numpy(), O.essential_matrix_grid(g[f"grid{i}_P"], g["grid_K"]))


@pytest.mark.parametrize("name", ["st", "st_soft", "ak"])
def test_essential_matrix_composites_vs_golden(mods, name):
    from onnx_image_processing_amd.pytorch_model.feature_detection import (
        AKAZESparseBADSinkhornWithEssentialMatrix, ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix)
    g = load_golden("essential_matrix")
    a, b = synth_batch(int(g["pair_seed"]), 1, 120, 160)
    cfg = cfg_of(g, name + "_cfg")
    cls = AKAZESparseBADSinkhornWithEssentialMatrix if name == "ak" else ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix
    model = cls(K=torch.from_numpy(g["cam_K"]), **cfg).to(DEV)
    k1, k2, p, e = [t.cpu().numpy() for t in model(gpu(a), gpu(b))]
    assert e.shape == (3, 3) and p.shape == (1, cfg["max_keypoints"] + 1, cfg["max_keypoints"] + 1)
    assert np.array_equal(k1, g[name + "_k1"]) and np.array_equal(k2, g[name + "_k2"])
    # the head itself: E from OUR P and keypoints through the oracle (same inputs, so only K10 is compared)
    eo = O.essential_matrix_keypoints(p[0], k1[0], k2[0], k1[0][:, 0] >= 0, k2[0][:, 0] >= 0, g["cam_K"])
    assert _e_close(e, eo)
    # end to end against the reference's E (P agrees to 1e-4, so the weight mask is the same here)
    assert _e_close(e, g[name + "_E"], tol=2e-3), np.abs(e - g[name + "_E"]).max()
    # two pairs at once (extension): (B,3,3), each equal to the single-pair result
    a2, b2 = np.concatenate([a, b]), np.concatenate([b, a])
    e2 = model(gpu(a2), gpu(b2))[3].cpu().numpy()
    assert e2.shape == (2, 3, 3) and _e_close(e2[0], e)


def test_vo_model_480x640_k512_reference_fixture(mods):
    """The visual-odometry model (sample/visual_odometry.py:520-545's session) at its
    deployment size against the recorded reference run: ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix, Angle
    export-CLI values (block 5, 512 hard pairs, epsilon 0.05, NMS 5), 640x480, K = 512, pinhole K.  Keypoint sets equal
    (sequence too on this input), P through maxima / argmaxima / dustbins / sample rows to 1e-4, the MNN match set with
    the strict tie rule, E within 1e-4 * max|E| of the reference's."""
    from onnx_image_processing_amd.pytorch_model.feature_detection import ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix
    g = load_golden("angle_vo_480x640_k512")
    cfg = cfg_of(g)
    k = int(g["k"])
    a, b = synth_batch(int(g["seed"]), 1, int(g["h"]), int(g["w"]))
    model = ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix(K=torch.from_numpy(g["cam_K"]), **cfg).to(DEV)
    k1, k2, p, e = model(gpu(a), gpu(b))
    k1n, k2n, pn, en = [t.cpu().numpy() for t in (k1, k2, p, e)]
    assert {tuple(x) for x in k1n[0]} == {tuple(x) for x in g["k1"][0]} and {tuple(x) for x in k2n[0]} == {tuple(x) for x in g["k2"][0]}
    if np.array_equal(k1n, g["k1"]) and np.array_equal(k2n, g["k2"]):
        core = pn[:, :k, :k]
        assert np.array_equal(core.argmax(2), g["P_rowarg"]) and np.array_equal(core.argmax(1), g["P_colarg"])
        for mine, key in ((core.max(2), "P_rowmax"), (core.max(1), "P_colmax"), (pn[:, :, k], "P_dustcol"),
                          (pn[:, k, :], "P_dustrow"), (pn[:, :8], "P_rows_0_8")):
            ok, worst = p_close(mine, g[key])
            assert ok, (key, worst)
        mcfg = cfg_of(g, "mnn_cfg")
        mk = [t.cpu().numpy() for t in mods["MutualNearestNeighborMatcher"](mcfg["max_matches"], mcfg["threshold"])(p, k1, k2)]
        check_match_sets(match_dict(mk[0][0], mk[1][0], mk[2][0], mk[3][0]),
                         match_dict(g["mk1"][0], g["mk2"][0], g["mscores"][0], g["mvalid"][0]), mcfg["max_matches"])
    else:
        pytest.fail("block-5 keypoint order differs from the reference's on the fixture input (sets equal): re-record or relax")
    assert en.shape == (3, 3) and _e_close(en, g["E"], tol=1e-3), np.abs(en - g["E"]).max()
    # uint8 frames (converted on the device for this family): identical outputs
    for x, y in zip(model(gpu(a.astype(np.uint8)), gpu(b.astype(np.uint8))), (k1, k2, p, e)):
        assert torch.equal(x, y)


@pytest.mark.parametrize("b,n,m,bits", [(3, 512, 512, 512), (2, 300, 77, 256), (1, 40, 56, 256), (2, 520, 700, 512)])
def test_essential_matrix_from_the_sinkhorn_solution_equals_from_p(mods, b, n, m, bits):
    """mi_essential_matrix_dots (the head on the uint16 dot products + the Sinkhorn duals, P never written) against
    mi_essential_matrix on the P mi_sinkhorn_dots writes: every entry is rebuilt with the solver's own final-pass
    expression, so E is identical BIT FOR BIT -- banded and dense form, with and without validity masks, n != m, m > 512."""
    from onnx_image_processing_amd import ops
    rng = np.random.default_rng(n * 3 + m)
    b1 = rng.integers(0, 2 ** 32, size=(b, n, bits // 32), dtype=np.uint64).astype(np.uint32)
    b2 = rng.integers(0, 2 ** 32, size=(b, m, bits // 32), dtype=np.uint64).astype(np.uint32)
    k = min(n, m) * 3 // 4
    b2[:, :k] = b1[:, :k]                                            # true correspondences
    flip = rng.integers(0, bits // 32, size=(b, k))
    for bi in range(b):
        b2[bi, np.arange(k), flip[bi]] ^= np.uint32(1) << rng.integers(0, 32, size=k).astype(np.uint32)
    t1, t2 = gpu(b1.view(np.int32)), gpu(b2.view(np.int32))
    p, u, v, state = ops.sinkhorn_bits(t1, t2, True, 0.05, 1.0, 20, want_p=True, return_state=True)
    q1 = gpu((rng.random((b, n, 2)) - 0.5).astype(np.float32))
    q2 = q1[:, :m].clone() if m <= n else gpu((rng.random((b, m, 2)) - 0.5).astype(np.float32))
    if m > n:
        q2[:, :n] = q1
    q2 = (q2 + 0.01).contiguous()
    v1 = gpu(rng.random((b, n)) > 0.1)
    v2 = gpu(rng.random((b, m)) > 0.1)
    for top_k in (1, 3, 4):
        for masks in ((None, None), (v1, v2)):
            for banded in (True, False):
                want = ops.essential_matrix(p, q1, q2, masks[0], masks[1], top_k, 30, 10, banded=banded)
                got = ops.essential_matrix_dots(state, m, 0.05, u, v, q1, q2, masks[0], masks[1], top_k, 30, 10, banded=banded)
                assert torch.isfinite(want).all() and torch.equal(got, want), (top_k, masks[0] is not None, banded)


def test_vo_model_extensions_equal_forward(mods):
    """_EssentialHead.essential / match_and_essential (extensions: E, or matches + E, straight from the Sinkhorn solution
    with P never written) against forward() + MutualNearestNeighborMatcher, bit for bit, batch of 3."""
    from onnx_image_processing_amd.pytorch_model.feature_detection import ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix
    g = load_golden("angle_vo_480x640_k512")
    a, b = synth_batch(9100, 3, 240, 320)
    model = ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix(K=torch.from_numpy(g["cam_K"]), **{**cfg_of(g), "max_keypoints": 128}).to(DEV)
    k1, k2, p, e = model(gpu(a), gpu(b))
    x1, x2, e2 = model.essential(gpu(a), gpu(b))
    assert torch.equal(x1, k1) and torch.equal(x2, k2) and torch.equal(e2, e)
    out = model.match_and_essential(gpu(a), gpu(b), 50, 0.1)
    want = mods["MutualNearestNeighborMatcher"](50, 0.1)(p, k1, k2)
    for x, y in zip(out[:4], want):
        assert torch.equal(x, y)
    assert torch.equal(out[4], e) and int(want[3].sum()) > 30
    # few pairs per call share the front end's launches (both images as one batch of 2B): the separate calls' outputs
    assert model.pair_launches
    model.pair_launches = False
    for x, y in zip(model(gpu(a), gpu(b)), (k1, k2, p, e)):
        assert torch.equal(x, y)


# ------------------------------------------------------------------ FAST / DoG detectors
def test_fast_and_dog_detectors(mods):
    from onnx_image_processing_amd.pytorch_model.detector import DoGDetector, DoGDetectorWithScore, FASTScore
    g = load_golden("detectors")
    img = np.stack([synth_image(int(g["seed"]) + i, int(g["h"]), int(g["w"])) for i in range(2)])[:, None].astype(np.float32)
    img[1] += np.float32(0.37)
    for thr in (20, 7):
        got = FASTScore(threshold=thr).to(DEV)(gpu(img)).cpu().numpy()
        assert np.array_equal(got, O.fast_score(img, thr))                                    # bit-exact vs oracle
        assert np.array_equal(np.packbits(got != 0), g[f"fast_t{thr}"])                       # and vs the reference
    nms = FASTScore(threshold=20, use_nms=True, nms_radius=3).to(DEV)(gpu(img)).cpu().numpy()
    assert np.array_equal(np.packbits(nms != 0), g["fast_nms"])
    assert set(FASTScore().state_dict()) == {"circle_offsets", "powers_of_2"}
    big = np.stack([synth_image(3600 + i, 130, 200) for i in range(3)])[:, None].astype(np.float32)   # several tiles
    assert np.array_equal(FASTScore(12).to(DEV)(gpu(big)).cpu().numpy(), O.fast_score(big, 12))
    dog = DoGDetector().to(DEV)
    d = dog(gpu(img)).cpu().numpy()
    assert d.shape == (2, 4, int(g["h"]), int(g["w"])) and set(dog.state_dict()) == {"gaussian_kernels"}
    np.testing.assert_allclose(d, O.dog_responses(img), rtol=0, atol=2e-3)
    np.testing.assert_allclose(d[:, :, ::3, ::3], g["dog_default"], rtol=0, atol=2e-3)
    small = DoGDetector(num_scales=3, sigma_base=1.0, sigma_ratio=1.5, kernel_size=9).to(DEV)(gpu(img)).cpu().numpy()
    np.testing.assert_allclose(small, g["dog_small"], rtol=0, atol=2e-3)
    sc = DoGDetectorWithScore().to(DEV)(gpu(img)).cpu().numpy()
    np.testing.assert_allclose(sc, g["dog_score"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(sc, np.abs(d).max(axis=1, keepdims=True), rtol=0, atol=0)      # same kernel, same sums
    with pytest.raises(ValueError):
        DoGDetector(num_scales=1)
    with pytest.raises(ValueError):
        DoGDetector(kernel_size=8)
    with pytest.raises(ValueError):
        dog(gpu(np.zeros((1, 3, 16, 16), np.float32)))


# ------------------------------------------------------------------ SparseBAD(sampling_mode="bilinear")
def test_sparse_bad_bilinear(mods):
    g = load_golden("bad_bilinear")
    a, _ = synth_batch(int(g["seed"]), 2, 96, 128)
    zero = np.zeros((2, 40), np.float32)
    for name, kw, pairs in (("raw", dict(normalize_descriptors=False), 256),
                            ("soft", dict(binarize=True, soft_binarize=True), 256),
                            ("hard", dict(binarize=True, soft_binarize=False), 512)):
        box, thr = bad_tables(pairs)
        mod = mods["SparseBAD"](pairs, sampling_mode="bilinear", **kw).to(DEV)
        for tag, kpts, ori, th in (("int", g["kp"], None, zero), ("frac", g["kf"], None, zero),
                                   ("ori", g["kf"], g["ang"], O.sample_nearest(g["ang"], g["kf"]))):
            got = mod(gpu(a), gpu(kpts), gpu(ori) if ori is not None else None).cpu().numpy()
            want = O.sparse_bad_oriented(a, kpts, th, box, thr, sampling_mode="bilinear", **kw)
            ref = g[f"{name}_{tag}"]
            if name == "hard":
                bits_mismatch(got != 0, want != 0, ALLOW[f"gpu_bilinear_{tag}_vs_oracle"], f"bilinear hard {tag} vs oracle")
                bits_mismatch(got != 0, ref != 0, ALLOW[f"gpu_bilinear_{tag}_vs_reference"], f"bilinear hard {tag} vs reference")
                bits = mod.forward_bits(gpu(a), gpu(kpts), gpu(ori) if ori is not None else None).cpu().numpy()
                assert np.array_equal(bits.view(np.uint32), O.pack_bits(got != 0))            # packed == float form
            else:
                # same exact box means; with an angle, cosf/sinf (GPU vs numpy) move the sample position by ~1e-6 px
                tol = (2e-3 if name == "raw" else 1e-4) if tag == "ori" else 5e-5
                np.testing.assert_allclose(got, want, rtol=0, atol=tol, err_msg=f"{name}/{tag}")
                np.testing.assert_allclose(got, ref, rtol=0, atol=2e-3 if name == "raw" else 1e-4, err_msg=f"{name}/{tag}")
    # a matcher built with sampling_mode="bilinear" runs end to end
    from onnx_image_processing_amd.pytorch_model.feature_detection import ShiTomasiSparseBADSinkhornMatcher
    m = ShiTomasiSparseBADSinkhornMatcher(max_keypoints=32, sampling_mode="bilinear").to(DEV)
    k1, k2, p = m(gpu(a[:1]), gpu(a[1:]))
    assert p.shape == (1, 33, 33) and bool(torch.isfinite(p).all())
    with pytest.raises(ValueError):
        mods["SparseBAD"](256, sampling_mode="cubic")


def test_dense_oriented_bad(mods):
    from onnx_image_processing_amd.pytorch_model.descriptor.bad import BADDescriptor
    g = load_golden("bad_bilinear")
    small = synth_image(3701, 21, 30)[None, None].astype(np.float32)
    box, thr = bad_tables(256)
    got = BADDescriptor(256).to(DEV)(gpu(small), gpu(g["dense_ang"])).cpu().numpy()
    np.testing.assert_allclose(got, O.bad_dense_oriented(small, g["dense_ang"], box, thr), rtol=0, atol=2e-3)
    np.testing.assert_allclose(got, g["dense_raw"], rtol=0, atol=2e-3)
    hard = BADDescriptor(512, binarize=True, soft_binarize=False).to(DEV)(gpu(small), gpu(g["dense_ang"])).cpu().numpy()
    ref = np.unpackbits(g["dense_hard"])[: hard.size].reshape(hard.shape)
    bits_mismatch(ref, hard != 0, ALLOW["gpu_dense_oriented_vs_reference"], "dense oriented hard vs reference")
    # several tiles, batch 2, angle 0 everywhere == the non-oriented map up to the bilinear round trip
    img = np.stack([synth_image(3710 + i, 45, 70) for i in range(2)])[:, None].astype(np.float32)
    zero = np.zeros_like(img)
    a0 = BADDescriptor(256).to(DEV)(gpu(img), gpu(zero)).cpu().numpy()
    plain = BADDescriptor(256).to(DEV)(gpu(img)).cpu().numpy()
    np.testing.assert_allclose(a0, plain, rtol=0, atol=2e-3)
    rng = np.random.default_rng(4)
    ang = ((rng.random(img.shape).astype(np.float32)) * 2 - 1) * np.float32(np.pi)
    box, thr = bad_tables(256)
    got = BADDescriptor(256, binarize=True, soft_binarize=True).to(DEV)(gpu(img), gpu(ang)).cpu().numpy()
    np.testing.assert_allclose(got, O.bad_dense_oriented(img, ang, box, thr, binarize=True, soft_binarize=True),
                               rtol=0, atol=2e-3)


@pytest.mark.parametrize("eps", [0.004, 0.006, 0.02])
def test_duplicate_descriptors_at_small_epsilon(mods, eps):
    """Identical normalised descriptors: the reference clamps the (rounding-noise) negative cost at 0
    (sinkhorn.py:103).  The packed uint16-dot form drops that clamp and is therefore only used for
    epsilon >= 0.005 (MI_DOTS_MIN_EPSILON); below, the clamped fp32-Z form runs.  Both against the fp64 oracle."""
    from onnx_image_processing_amd import ops
    rng = np.random.default_rng(77)
    n, words = 96, 16
    bits = rng.integers(0, 2 ** 32, size=(1, n, words), dtype=np.uint64).astype(np.uint32)
    bits2 = bits.copy()
    bits2[0, n // 2:] = rng.integers(0, 2 ** 32, size=(n - n // 2, words), dtype=np.uint64).astype(np.uint32)
    d1 = unpack_bits(bits, 512).astype(np.float64)
    d2 = unpack_bits(bits2, 512).astype(np.float64)
    d1 /= np.linalg.norm(d1, axis=2, keepdims=True)
    d2 /= np.linalg.norm(d2, axis=2, keepdims=True)
    want = O.sinkhorn_match(d1, d2, 20, eps, 1.0, "l2", dtype=np.float64)
    m = mods["SinkhornMatcher"](iterations=20, epsilon=eps)
    assert ops.dots_supported(1, n, n, eps) == (eps >= 0.005)
    got = m.forward_bits(gpu(bits.view(np.int32)), gpu(bits2.view(np.int32)), True).cpu().numpy()
    ok, worst = p_close(got, want)
    assert ok, worst
    assert got[0, :n // 2, :n // 2].diagonal().min() > 0.5        # the duplicates are matched (the dustbin takes the rest)


# ------------------------------------------------------------------ u8 ingest
@pytest.mark.parametrize("shape", [(2, 480, 640), (3, 96, 128), (1, 37, 64), (2, 61, 83), (1, 8, 8), (1, 200, 136),
                                   (2, 50, 132), (1, 33, 45)])
def test_corner_u8_equals_f32_and_oracle(mods, shape):
    """mi_corner_response_u8: the score map from uint8 frames is the float32 path's, bit for bit (streaming kernel for
    w % 4 == 0, generic kernel otherwise), for blocks 3 / 5."""
    n, h, w = shape
    img = np.stack([synth_image(700 + i, h, w) for i in range(n)])[:, None]
    assert img.dtype == np.uint8
    for bs in (3, 5):
        det = mods["ShiTomasiScore"](bs)
        got = det(gpu(img))
        assert got.dtype == torch.float32 and torch.equal(got, det(gpu(img.astype(np.float32))))
        if bs == 3:
            assert np.array_equal(got.cpu().numpy(), O.shi_tomasi_score(img.astype(np.float32), 3))
    view = gpu(np.concatenate([img, img], 1))[:, 1:]                  # a non-contiguous uint8 view is accepted too
    assert torch.equal(mods["ShiTomasiScore"](3)(view), mods["ShiTomasiScore"](3)(gpu(img)))


def test_u8_pipeline_equals_f32_on_the_c2_fixture(mods):
    """The north-star pair as uint8 frames: keypoints, packed bits, P, matches all `torch.equal` to the float32 path
    (which the other tests pin to the oracle and to the reference output), module path and the one-call form."""
    from onnx_image_processing_amd.synth import synth_batch_u8
    g = load_golden("c2_pair_480x640_k512")
    cfg = cfg_of(g)
    a8, b8 = synth_batch_u8(int(g["seed"]), 1, 480, 640)
    a, b = _images(g)
    assert np.array_equal(a8.astype(np.float32), a)
    model = mods["ShiTomasiSparseBADSinkhornMatcher"](max_keypoints=512, **cfg).to(DEV)
    want = model(gpu(a), gpu(b))
    got = model(gpu(a8), gpu(b8))
    for x, y in zip(got, want):
        assert torch.equal(x, y)
    perm = tie_canonical_perm(g["kpts1"][0], g["kscores1"][0], 640)
    assert np.array_equal(got[0][0].cpu().numpy(), g["kpts1"][0][perm])            # and hence the reference's keypoints
    for im8, im, kp in ((a8, a, got[0]), (b8, b, got[1])):
        assert torch.equal(model.descriptor.forward_bits(gpu(im8), kp), model.descriptor.forward_bits(gpu(im), kp))
        model.descriptor.use_fast_path = False                                       # the general kernel on uint8
        slow = model.descriptor.forward_bits(gpu(im8), kp)
        model.descriptor.use_fast_path = True
        assert torch.equal(slow, model.descriptor.forward_bits(gpu(im8), kp))
    wrap = mods["MatchExtractionWrapper"](model, max_matches=100, match_threshold=0.1)
    m32, m8 = wrap(gpu(a), gpu(b)), wrap(gpu(a8), gpu(b8))
    one8 = wrap.forward_single_call(gpu(a8), gpu(b8), want_keypoints=True)
    one32 = wrap.forward_single_call(gpu(a), gpu(b), want_keypoints=True)
    for x, y, z, t in zip(m8, m32, one8[2:], one32[2:]):
        assert torch.equal(x, y) and torch.equal(z, y) and torch.equal(t, y)
    assert torch.equal(one8[0], want[0]) and torch.equal(one8[1], want[1])
    mixed = wrap.forward_single_call(gpu(a8), gpu(b), want_keypoints=True)           # one uint8, one float32 frame
    assert torch.equal(mixed[2], m32[0])


@pytest.mark.parametrize("h,w,k", [(97, 131, 40), (64, 72, 32), (120, 160, 96), (50, 132, 24)])
def test_u8_pipeline_odd_sizes_and_other_modes(mods, h, w, k):
    """uint8 frames on widths that are not multiples of 4 / 16, batch 3, keypoints at the border (margin 0), soft and raw
    descriptors (general BAD kernel), and the variants without a uint8 kernel (device-side conversion)."""
    from onnx_image_processing_amd.pytorch_model.feature_detection import ShiTomasiAngleSparseBADSinkhornMatcher
    from onnx_image_processing_amd.synth import synth_batch_u8
    a8, b8 = synth_batch_u8(9100 + h, 3, h, w, noise=1)
    a, b = a8.astype(np.float32), b8.astype(np.float32)
    for cfg in (dict(num_pairs=256, binarize=True, soft_binarize=False, epsilon=0.1, nms_radius=2, border_margin=0),
                dict(num_pairs=512, binarize=True, soft_binarize=True, epsilon=0.5, nms_radius=3),
                dict(num_pairs=256, binarize=False, normalize_descriptors=False, epsilon=30.0, nms_radius=2)):
        model = mods["ShiTomasiSparseBADSinkhornMatcher"](max_keypoints=k, sinkhorn_iterations=6, **cfg).to(DEV)
        for x, y in zip(model(gpu(a8), gpu(b8)), model(gpu(a), gpu(b))):
            assert torch.equal(x, y)
    ang = ShiTomasiAngleSparseBADSinkhornMatcher(max_keypoints=k, num_pairs=256, sinkhorn_iterations=5).to(DEV)
    for x, y in zip(ang(gpu(a8), gpu(b8)), ang(gpu(a), gpu(b))):
        assert torch.equal(x, y)


@pytest.mark.parametrize("h,w", [(97, 131), (64, 70), (33, 45)])
def test_odd_image_sizes_vs_oracle(mods, h, w):
    """Widths that are not a multiple of 4 take the generic corner / NMS kernels; keypoints near every border."""
    a, b = synth_batch(4100 + h, 2, h, w)
    sc = O.shi_tomasi_score(a, 3)[:, 0]
    for r in (2, 5):
        assert np.array_equal(mods["apply_nms_maxpool"](gpu(sc), r).cpu().numpy(), O.nms_mask(sc, r))
    cfg = dict(block_size=3, num_pairs=256, binarize=True, soft_binarize=False, sinkhorn_iterations=8, epsilon=0.1,
               nms_radius=2, border_margin=0)
    k = 40
    model = mods["ShiTomasiSparseBADSinkhornMatcher"](max_keypoints=k, **cfg).to(DEV)
    k1, k2, p = [t.cpu().numpy() for t in model(gpu(a), gpu(b))]
    box, thr = bad_tables(256)
    kw = {kk: v for kk, v in cfg.items() if kk != "num_pairs"}
    o1, o2, op = O.match_pair(a, b, box, thr, k, **kw)
    assert np.array_equal(k1, o1) and np.array_equal(k2, o2)
    ok, worst = p_close(p, op)
    assert ok, worst


# ------------------------------------------------------------------ the whole path in one C-ABI call


def test_bench_pairs_match_sets_equal_the_reference(mods):
    """The 64 pairs bench.py's live `parity` object checks (seeds 1000..1063, 640x480, K=512, export-CLI values) against
    what the REFERENCE produced for them (tests/golden/bench_seeds_matches.npz, written by make_golden.py --round3-only
    from the imported reference): same match set per pair; where the sets differ it must be a tie at the max_matches
    cut -- scores within 1e-4 of the cut score, and the extra match is one of the reference's own mutual matches with the
    same score (the fixture holds ALL of them) -- checked match by match (VERDICT r2 weak #1b, next #7).  All four forms:
    module path and mi_match_pairs, float32 and uint8 frames."""
    from onnx_image_processing_amd.synth import synth_batch_u8
    g = load_golden("bench_seeds_matches")
    n, cfg, mcfg = int(g["pairs"]), cfg_of(g), cfg_of(g, "mnn_cfg")
    a8, b8 = synth_batch_u8(int(g["first_seed"]), n, int(g["h"]), int(g["w"]))
    model = mods["MatchExtractionWrapper"](mods["ShiTomasiSparseBADSinkhornMatcher"](max_keypoints=int(g["k"]), **cfg),
                                           max_matches=mcfg["max_matches"], match_threshold=mcfg["threshold"]).to(DEV)
    outs = {"module f32": model(gpu(a8).float(), gpu(b8).float()), "module u8": model(gpu(a8), gpu(b8)),
            "one call f32": model.forward_single_call(gpu(a8).float(), gpu(b8).float()),
            "one call u8": model.forward_single_call(gpu(a8), gpu(b8))}
    first = None
    for name, out in outs.items():
        mk1, mk2, sc, valid = [t.cpu().numpy() for t in out]
        if first is None:
            first = (mk1, mk2, sc, valid)
        else:                                                     # the four forms agree bit for bit
            assert all(np.array_equal(x, y) for x, y in zip((mk1, mk2, sc, valid), first)), name
    mk1, mk2, sc, valid = first
    kinds = []
    for i in range(n):
        mutual = {tuple(map(float, r[:4])): float(r[4]) for r in g["mutual"][i][:int(g["n_mutual"][i])]}
        assert len(mutual) == int(g["n_mutual"][i])
        want = match_dict(g["mk1"][i], g["mk2"][i], g["mscores"][i], g["mvalid"][i])
        assert set(want) <= set(mutual)
        kinds.append(check_match_sets(match_dict(mk1[i], mk2[i], sc[i], valid[i]), want, mcfg["max_matches"], mutual=mutual))
    if os.environ.get("MI_REPORT"):
        print(f"[bench pairs vs reference] identical {kinds.count('same')}, cut ties {kinds.count('cut')} of {n}")
    # the MEASURED count on MI355X, like every other allowance of tests/helpers.py: 61 identical, 3 ties at the cut
    assert kinds.count("cut") <= ALLOW["gpu_bench_pairs_cut_ties"], kinds.count("cut")


def test_dense_variant_480x640_k512_reference_fixture(mods):
    """ShiTomasiBADSinkhornMatcher (BASELINE configs[2]'s "dense BAD" reading, what `bench.py --workload c3dense` times)
    at 640x480, K=512, P=512 against the recorded reference run.  The reference samples a dense map built from an fp32
    integral image that is inexact above 2^24 (255 x 480 x 640 = 7.8e7), so its own bits are wrong where the response
    is within that error of zero; this build evaluates the same responses exactly at the keypoints.  Hence:
    keypoints identical (no border margin); descriptor bits identical except where the reference's own raw response is
    within 1.0 intensity units of zero (measured: 103 + 108 of 2 x 262,144 bits, all with |raw| < 0.5); the matcher on
    the REFERENCE's bits reproduces the reference's P to 1e-4 (maxima, argmaxima, dustbins, marginals, eight full rows)
    and its MNN match set; end to end every row's best match is the reference's."""
    from onnx_image_processing_amd.pytorch_model.feature_detection import ShiTomasiBADSinkhornMatcher
    g = load_golden("dense_c3_480x640_k512")
    cfg = cfg_of(g)
    k = int(g["k"])
    a, b = synth_batch(int(g["seed"]), 1, int(g["h"]), int(g["w"]))
    model = ShiTomasiBADSinkhornMatcher(**cfg).to(DEV)
    k1, k2, p = model(gpu(a), gpu(b))
    assert np.array_equal(k1.cpu().numpy(), g["k1"]) and np.array_equal(k2.cpu().numpy(), g["k2"])
    kk = k1.cpu().numpy()[0]
    assert kk[:, 0].min() < 7 or kk[:, 1].min() < 7 or kk[:, 0].max() > 480 - 8 or kk[:, 1].max() > 640 - 8   # margin 0 is exercised
    bad = model.detector.descriptor
    total = 0
    for tag, im, kp in (("1", a, k1), ("2", b, k2)):
        _, bits = ops_sparse_bits(bad, gpu(im), kp)
        mine, ref = unpack_bits(bits.cpu().numpy().view(np.uint32), 512), unpack_bits(g["bits" + tag], 512)
        near = {tuple(i): v for i, v in zip(g["near_idx" + tag], g["near_val" + tag])}
        diff = np.argwhere(mine != ref)
        total += len(diff)
        for idx in diff:
            assert tuple(idx) in near and abs(near[tuple(idx)]) < 1.0, (tag, idx)
    assert total <= 2 * 262144 // 1000                           # < 0.1 %; measured 211
    # the matcher on the reference's own bits: P as recorded
    rb = [gpu(g["bits" + t].view(np.int32)) for t in "12"]
    pr = model.matcher.forward_bits(rb[0], rb[1], True).cpu().numpy()
    core = pr[:, :k, :k]
    assert np.array_equal(core.argmax(2), g["P_rowarg"]) and np.array_equal(core.argmax(1), g["P_colarg"])
    for mine_, key in ((core.max(2), "P_rowmax"), (core.max(1), "P_colmax"), (pr[:, :, k], "P_dustcol"), (pr[:, k, :], "P_dustrow"),
                       (pr.sum(-1), "P_rowsum"), (pr.sum(-2), "P_colsum"), (pr[:, :8], "P_rows_0_8")):
        ok, worst = p_close(mine_, g[key], atol=2e-4 if key.endswith("sum") else 1e-4)
        assert ok, (key, worst)
    mcfg = cfg_of(g, "mnn_cfg")
    mk = [t.cpu().numpy() for t in mods["MutualNearestNeighborMatcher"](mcfg["max_matches"], mcfg["threshold"])(gpu(pr), k1, k2)]
    check_match_sets(match_dict(mk[0][0], mk[1][0], mk[2][0], mk[3][0]),
                     match_dict(g["mk1"][0], g["mk2"][0], g["mscores"][0], g["mvalid"][0]), mcfg["max_matches"])
    # end to end (exact bits): every row's best match is the reference's; the probabilities move with the ~0.04 % bits
    # that differ between the reference's inexact fp32 integral image and the exact evaluation here -- inherent to the
    # reference (its own two runs of different conv algorithms differ the same way); the matcher ON THE REFERENCE'S BITS
    # is held to 1e-4 above, so 0.06 on the row maxima is the bound of the descriptor stage, not of the solver
    mine = p.cpu().numpy()[:, :k, :k]
    assert np.array_equal(mine.argmax(2), g["P_rowarg"])
    assert np.abs(mine.max(2) - g["P_rowmax"]).max() < 0.06


def ops_sparse_bits(bad, image, kp):
    from onnx_image_processing_amd import ops
    return ops.sparse_bad(image, kp, bad.pair_geom, bad.pair_thr, bad.mode, bad.temperature, True, want_desc=False, want_bits=True)


