"""CPU-only checks: the C-ABI library loads and exports every symbol the header declares (no
compute call is made without a GPU), the nn.Module mirrors keep the reference's constructor
contract, and the product refuses CPU tensors instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HEADER = os.path.join(ROOT, "include", "mi355x_match.h")
DEBUG_HEADER = os.path.join(ROOT, "include", "mi355x_match_debug.h")


@pytest.fixture(scope="module")
def lib_path():
    from onnx_image_processing_amd.build import build
    return build(verbose=False)          # hipcc cross-compiles gfx950 without a GPU


def header_functions(path=HEADER):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    names = header_functions()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/mi355x_match.h but not exported"
    assert lib.mi_abi_version() == 3
    lib.mi_error_string.restype = ctypes.c_char_p
    assert lib.mi_error_string(0) == b"ok" and b"NULL" in lib.mi_error_string(-1)


def test_binding_covers_the_header(lib_path):
    from onnx_image_processing_amd import _native
    assert sorted(_native.SIGNATURES) == header_functions()
    assert sorted(_native.DEBUG_SIGNATURES) == header_functions(DEBUG_HEADER) == [
        "mi_debug_akaze_math_check", "mi_debug_bad_plan_passes", "mi_debug_clock_probe", "mi_debug_set",
        "mi_debug_sinkhorn_dots_form", "mi_debug_topk_stamps", "mi_debug_tuner_script"]
    _native.load()
    with _native.debug_library() as dbg:                     # the debug build exports both headers
        assert dbg.mi_abi_version() == 3


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())


def test_product_library_exports_the_header_and_nothing_else(lib_path):
    """-fvisibility=hidden + the version script: the product library's dynamic symbols are exactly the MI_API
    declarations of include/mi355x_match.h -- no C++-mangled internals, no libstdc++ instantiations, and none of the
    process-wide kernel selectors (mi_debug_*), which live in the separate debug library."""
    from onnx_image_processing_amd.build import DEBUG_LIB
    assert _exported(lib_path) == header_functions()
    assert _exported(DEBUG_LIB) == sorted(header_functions() + header_functions(DEBUG_HEADER))
    import subprocess
    syms = subprocess.run(["nm", lib_path], capture_output=True, text=True).stdout
    assert "mi_hooks" not in syms and "mi_debug" not in syms                 # not even as local symbols


def test_sinkhorn_form_decision_is_a_host_check(lib_path):
    """Which Sinkhorn form runs is decided on the host from (extents, caller flags, what the device can hold): the
    single-launch form only when its whole grid -- 16 workgroups of 512 threads per pair, pairs rounded up to 8 -- fits
    occupancy x compute units; a CU-masked / partitioned device or MI_SOLVER_MULTI_LAUNCH takes the multi-launch form
    (ADVICE r2 medium #1).  Pure function, no GPU."""
    from onnx_image_processing_amd import _native
    with _native.debug_library() as dbg:
        form = dbg.mi_debug_sinkhorn_dots_form
        assert form(1, 512, 512, 0, 4, 256) == 1 and form(8, 512, 512, 0, 4, 256) == 1      # a whole MI355X
        assert form(8, 512, 512, 0, 1, 128) == 1 and form(8, 512, 512, 0, 1, 127) == 0        # 128 workgroups needed
        assert form(1, 512, 512, 0, 3, 42) == 0 and form(1, 512, 512, 0, 4, 32) == 1          # one pair still launches 128
        assert form(8, 512, 512, 0, 0, 256) == 0                                              # occupancy query failed
        assert form(8, 512, 512, 1, 4, 256) == 0                                              # the caller's flag
        assert form(9, 512, 512, 0, 4, 256) == 0 and form(8, 513, 512, 0, 4, 256) == 0        # not a single-launch shape
        assert form(8, 512, 600, 0, 4, 256) == 0
        assert form(2, 100, 100, 0, 1, 32) == 1 and form(2, 100, 100, 0, 1, 31) == 0          # 4 bands x 8 slots
        assert dbg.mi_debug_set(7, 0) == 0 and form(1, 512, 512, 0, 4, 256) == 0              # the test hook
    with _native.debug_library() as dbg:                      # leaving the block reset the hook
        assert dbg.mi_debug_sinkhorn_dots_form(1, 512, 512, 0, 4, 256) == 1
    lib = _native.load()
    ptr = ctypes.cast(ctypes.create_string_buffer(64), ctypes.c_void_p)
    assert lib.mi_sinkhorn_dots(ptr, ptr, ptr, 1, 8, 8, 8, 0.05, 1.0, 1.0, 5, ptr, ptr, None, ptr, 1 << 20, 8, None) == -3   # unknown flag
    base = ctypes.create_string_buffer(1 << 16)
    addr = ctypes.addressof(base)
    assert lib.mi_sinkhorn_dots_status_word(None, 1, 8, 8) is None
    assert lib.mi_sinkhorn_dots_status_word(addr, 1, 8, 2000) is None                        # not a dots shape
    lib.mi_sinkhorn_dots_workspace_bytes.restype = ctypes.c_size_t
    for shape in ((1, 8, 8), (8, 512, 512), (9, 512, 512), (64, 512, 1024)):
        w = lib.mi_sinkhorn_dots_status_word(addr, *shape)
        total = lib.mi_sinkhorn_dots_workspace_bytes(*shape)
        assert addr < w and w % 4 == 0 and w + 16 <= addr + total, shape                     # inside the workspace


def test_no_environment_variable_changes_which_kernels_run(lib_path):
    """The test hook is reachable only through an explicit mi_debug_set call."""
    src = open(os.path.join(ROOT, "onnx_image_processing_amd", "_native.py")).read()
    assert "MI_DEBUG_SET" not in src and "mi_debug_set" not in open(HEADER).read()
    for f in ("bench.py", "__graft_entry__.py"):
        assert "debug_set" not in open(os.path.join(ROOT, f)).read(), f


def test_argument_validation_happens_before_any_launch(lib_path):
    """Bad arguments are rejected on the host (negative MI_E_* codes): safe to call without a GPU."""
    lib = ctypes.CDLL(lib_path)
    lib.mi_corner_response.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p]
    assert lib.mi_corner_response(None, 1, 8, 8, 3, None, None) == -1            # MI_E_NULL
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.mi_corner_response(p, 0, 8, 8, 3, p, None) == -2                  # MI_E_SHAPE
    assert lib.mi_corner_response(p, 1, 8, 8, 4, p, None) == -3                  # MI_E_PARAM (even block)
    lib.mi_sinkhorn_workspace_bytes.restype = ctypes.c_size_t
    assert lib.mi_sinkhorn_workspace_bytes(2, 512, 512) == 2 * 17 * 513 * 8 + 2 * (512 + 4) * 4   # partials + padded v
    assert lib.mi_sinkhorn_workspace_bytes(2, 512, 5000) == 0
    seg, cap = ctypes.c_int(), ctypes.c_int()
    assert lib.mi_candidate_layout(480, 640, ctypes.byref(seg), ctypes.byref(cap)) == 0
    assert (seg.value, cap.value) == (75, 4096) and seg.value * cap.value >= 480 * 640


def test_constructor_contract_matches_reference():
    from onnx_image_processing_amd.pytorch_model.descriptor.bad import SparseBAD
    from onnx_image_processing_amd.pytorch_model.detector import ShiTomasiScore
    from onnx_image_processing_amd.pytorch_model.feature_detection import (
        MatchExtractionWrapper, ShiTomasiSparseBADSinkhornMatcher)
    from onnx_image_processing_amd.pytorch_model.matching import SinkhornMatcher
    # reference detector/shi_tomasi.py:37-41
    with pytest.raises(ValueError, match="sobel_size must be 3"):
        ShiTomasiScore(3, 5)
    for bad in (0, 4, -1):
        with pytest.raises(ValueError, match="positive odd integer"):
            ShiTomasiScore(bad)
    # reference descriptor/bad.py:385-392
    with pytest.raises(ValueError, match="num_pairs must be 256 or 512"):
        SparseBAD(128)
    with pytest.raises(ValueError, match="sampling_mode"):
        SparseBAD(256, sampling_mode="bicubic")
    # reference matching/sinkhorn.py:66-77
    with pytest.raises(ValueError, match="iterations must be positive"):
        SinkhornMatcher(iterations=0)
    with pytest.raises(ValueError, match="epsilon must be positive"):
        SinkhornMatcher(epsilon=-1.0)
    with pytest.raises(ValueError, match="distance_type"):
        SinkhornMatcher(distance_type="cos")
    assert SinkhornMatcher(distance_type="L2").distance_type == "l2"
    m = ShiTomasiSparseBADSinkhornMatcher(max_keypoints=64)
    assert m.border_margin == 7 and m.descriptor.max_radius == 7        # :120-124 (None -> max radius)
    assert ShiTomasiSparseBADSinkhornMatcher(64, border_margin=0).border_margin == 0
    assert (m.descriptor.num_pairs, m.descriptor.binarize, m.matcher.epsilon, m.nms_radius) == (256, False, 1.0, 3)
    w = MatchExtractionWrapper(m)
    assert (w.match_extractor.max_matches, w.match_extractor.threshold) == (100, 0.1)
    # buffer names / shapes / dtypes (state_dict round trip with the reference)
    sd = ShiTomasiSparseBADSinkhornMatcher(64, num_pairs=512).state_dict()
    expect = {"corner_detector.sobel_xy": (2, 1, 3, 3), "corner_detector.sum_kernel_grouped": (3, 1, 3, 3),
              "descriptor.offset_x1": (512,), "descriptor.radii": (512,), "descriptor.thresholds_v": (1, 1, 512),
              "descriptor.radius_select": (8, 512), "descriptor.box_kernel_bank": (8, 1, 15, 15)}
    for k, shape in expect.items():
        assert tuple(sd[k].shape) == shape, k
    assert sd["descriptor.radii"].dtype == torch.int64 and len(sd) == 15
    assert torch.allclose(sd["descriptor.box_kernel_bank"].sum((1, 2, 3)), torch.ones(8))


def test_bad_tables_geometry_invariants():
    """Properties the K4 window proof and the fast path rely on."""
    from onnx_image_processing_amd.pytorch_model.descriptor.bad_params import _get_bad_learned_params
    for p in (256, 512):
        box, thr = _get_bad_learned_params(p)
        assert box.shape == (p, 5) and thr.shape == (p,) and box.dtype == torch.float32
        x1, x2, y1, y2, r = box.T
        for c in (x1, x2, y1, y2):
            assert (c - r).min() >= 0 and (c + r).max() <= 31
        assert r.min() >= 1 and r.max() == 7
        # thresholds are odd multiples of 0.05: (S1 - S2)/area == thr can never hold for integers
        k = torch.round(thr.double() * 20)
        assert torch.allclose(k / 20, thr.double(), atol=1e-5) and bool((k.long() % 2 != 0).all())
    with pytest.raises(ValueError):
        _get_bad_learned_params(64)


def test_cpu_tensors_are_refused_not_emulated(lib_path):
    from onnx_image_processing_amd.pytorch_model.detector import ShiTomasiScore
    from onnx_image_processing_amd.pytorch_model.matching import SinkhornMatcher
    from onnx_image_processing_amd.pytorch_model.utils import apply_nms_maxpool
    with pytest.raises(RuntimeError, match="no CPU path"):
        ShiTomasiScore()(torch.zeros(1, 1, 16, 16))
    with pytest.raises(RuntimeError, match="no CPU path"):
        apply_nms_maxpool(torch.zeros(1, 16, 16), 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        SinkhornMatcher()(torch.zeros(1, 4, 8), torch.zeros(1, 4, 8))
    with pytest.raises(RuntimeError, match=r"\(N, 1, H, W\)"):
        ShiTomasiScore()(torch.zeros(1, 3, 16, 16))


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "onnx_image_processing_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(d, f)).read()
                assert "numpy_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_synthetic_inputs_are_deterministic():
    from onnx_image_processing_amd.synth import synth_batch, synth_image, synth_pair
    a = synth_image(1000)
    assert a.dtype == np.uint8 and a.shape == (480, 640)
    assert int(a[:4, :4].astype(np.int64).sum()) == int(synth_image(1000)[:4, :4].astype(np.int64).sum())
    assert (int(a.min()), int(a.max())) == (0, 253) or a.max() <= 255
    p, q = synth_pair(7, 64, 96)
    assert np.array_equal(np.roll(p, (3, 5), (0, 1)), q)
    x, y = synth_batch(5, 2, 32, 48, noise=2)
    assert x.shape == (2, 1, 32, 48) and x.dtype == np.float32 and np.abs(y[0, 0] - np.roll(x[0, 0], (3, 5), (0, 1))).max() <= 2
    import hashlib
    assert hashlib.sha256(synth_image(1000).tobytes()).hexdigest()[:16] == hashlib.sha256(a.tobytes()).hexdigest()[:16]


def test_graft_entry_smoke_arguments_are_oracle_compatible():
    """smoke() hands its hyper-parameters to the oracle: keep that call valid without a GPU."""
    import inspect
    import re

    from oracle import numpy_oracle as O
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    cfg_src = re.search(r"cfg = dict\((.*?)\)\n", src, re.S).group(1)
    keys = set(re.findall(r"(\w+)=", cfg_src)) - {"num_pairs"}
    assert keys <= set(inspect.signature(O.match_pair).parameters), keys
    assert 'if k != "num_pairs"' in src


def test_argument_validation_of_the_widened_entries(lib_path):
    """Host-side MI_E_* checks of the later entry points (no launch happens: safe without a GPU)."""
    lib = ctypes.CDLL(lib_path)
    buf = ctypes.create_string_buffer(256)
    p = ctypes.cast(buf, ctypes.c_void_p)
    vp, ci, cf, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double
    lib.mi_essential_matrix.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp, ci, ci, ci, vp, vp, ctypes.c_size_t, vp]
    assert lib.mi_essential_matrix(None, 1, 8, 8, p, p, None, None, 3, 30, 10, p, None, 0, None) == -1       # NULL P
    assert lib.mi_essential_matrix(p, 1, 8, 8, p, p, p, None, 3, 30, 10, p, None, 0, None) == -1             # one validity mask only
    assert lib.mi_essential_matrix(p, 1, 2000, 8, p, p, None, None, 3, 30, 10, p, None, 0, None) == -3       # n > 1024
    assert lib.mi_essential_matrix(p, 1, 8, 8, p, p, None, None, 9, 30, 10, p, None, 0, None) == -3          # top_k > 8
    assert lib.mi_essential_matrix(p, 1, 2, 8, p, p, None, None, 3, 30, 10, p, None, 0, None) == -3          # top_k > n
    lib.mi_essential_matrix_workspace_bytes.restype = ctypes.c_size_t
    assert lib.mi_essential_matrix_workspace_bytes(1, 512, 512, 5) == 0                                   # banded form: top_k <= 4
    need = lib.mi_essential_matrix_workspace_bytes(2, 512, 512, 3)
    assert need >= 2 * (512 * 4 + 512 + 512 * 8 * 8 + 16 * 512 * 3 * 4)
    assert lib.mi_essential_matrix(p, 2, 512, 512, p, p, None, None, 3, 30, 10, p, p, need - 1, None) == -4  # workspace too small
    assert lib.mi_essential_matrix(p, 2, 512, 512, p, p, None, None, 3, 30, 10, p, ctypes.c_void_p(p.value + 4), need, None) == -5
    lib.mi_mnn_duals_workspace_bytes.restype = ctypes.c_size_t
    lib.mi_mnn_duals_workspace_bytes.argtypes = [ci, ci, ci]
    assert lib.mi_mnn_duals_workspace_bytes(2, 512, 512) == (2 * 512 + 2 * 512 + 2 * 16 * 512) * 8
    assert lib.mi_mnn_duals_workspace_bytes(2, 512, 2000) == 0                                       # m > 1024: unsupported
    lib.mi_mnn_from_duals.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, vp, ci, cf, vp, ctypes.c_size_t, vp, vp, vp, vp, vp, vp]
    assert lib.mi_mnn_from_duals(None, 1, 8, 8, 8, p, p, p, p, 4, 0.1, p, 4096, p, p, p, p, None, None) == -1
    assert lib.mi_mnn_from_duals(p, 1, 8, 8, 6, p, p, p, p, 4, 0.1, p, 4096, p, p, p, p, None, None) == -5    # pitch < m
    assert lib.mi_mnn_from_duals(p, 1, 8, 8, 8, p, p, p, p, 4, 0.1, p, 8, p, p, p, p, None, None) == -4       # workspace too small
    lib.mi_akaze_diffuse.argtypes = [vp, ci, ci, ci, cf, cf, vp, vp]
    assert lib.mi_akaze_diffuse(p, 1, 8, 8, 0.05, 0.25, p, None) == -1                               # in place is refused
    lib.mi_akaze_hessian_scores.argtypes = [vp, ci, ci, ci, cf, ci, vp, vp]
    assert lib.mi_akaze_hessian_scores(p, 1, 8, 8, 0.001, 4, p, None) == -3                          # even NMS window
    lib.mi_dog_responses.argtypes = [vp, ci, ci, ci, vp, ci, ci, vp, vp, vp]
    assert lib.mi_dog_responses(p, 1, 8, 8, p, 1, 9, p, None, None) == -3                            # fewer than 2 scales
    assert lib.mi_dog_responses(p, 1, 8, 8, p, 3, 8, p, None, None) == -3                            # even kernel
    assert lib.mi_dog_responses(p, 1, 8, 8, p, 3, 9, None, None, None) == -1                         # no output at all
    lib.mi_sparse_bad_oriented.argtypes = [vp, ci, ci, ci, vp, ci, vp, vp, vp, vp, ci, ci, cf, ci, ci, cf, vp, vp, vp, vp]
    assert lib.mi_sparse_bad_oriented(p, 1, 64, 64, p, 4, p, p, p, p, 256, 2, 10.0, 1, 0, 0.0, p, None, None, None) == -1   # two angle sources
    assert lib.mi_sparse_bad_oriented(p, 1, 64, 64, p, 4, p, None, p, p, 100, 2, 10.0, 1, 0, 0.0, p, None, None, None) == -3  # pairs % 64
    assert lib.mi_sparse_bad_oriented(p, 1, 64, 64, p, 4, p, None, p, p, 256, 2, 10.0, 1, 0, -1.0, p, None, None, None) == -3  # negative reach


def test_match_pairs_host_side_checks(lib_path):
    """mi_match_pairs_workspace_bytes and the argument checks of mi_match_pairs run on the host (no GPU touched)."""
    from onnx_image_processing_amd import _native as N
    lib = N.load()
    fake = ctypes.create_string_buffer(64)
    ptr = ctypes.cast(fake, ctypes.c_void_p).value
    prm = N.MatchParams(3, 5, 512, 0.0, 7, 512, ptr, ptr, None, 1, 0.05, 1.0, 20, 100, 0.1)
    per_pair = lib.mi_match_pairs_workspace_bytes(1, 480, 640, ctypes.byref(prm))
    # up to 32 pairs both images share one launch per stage: 2 x (score map 1.2 MB + candidates 2.4 MB) + dots 0.5 MB +
    # the single-launch Sinkhorn's granules 0.13 MB + ...
    assert 7_500_000 < per_pair < 9_500_000
    assert lib.mi_match_pairs_workspace_bytes(8, 480, 640, ctypes.byref(prm)) >= 8 * (per_pair - 32768)   # fixed part: alignment + K1 ticket counters
    big = lib.mi_match_pairs_workspace_bytes(64, 480, 640, ctypes.byref(prm))     # one image side at a time
    assert 64 * 4_000_000 < big < 64 * 6_000_000
    assert prm.flags == 0
    for field, bad in (("max_keypoints", 2000), ("num_pairs", 100), ("block_size", 4), ("sinkhorn_iterations", 0),
                       ("epsilon", 0.0), ("max_matches", 0), ("flags", 4)):
        good = getattr(prm, field)
        setattr(prm, field, bad)
        assert lib.mi_match_pairs_workspace_bytes(1, 480, 640, ctypes.byref(prm)) == 0, field
        setattr(prm, field, good)
    assert lib.mi_match_pairs(None, None, 1, 480, 640, ctypes.byref(prm), None, None, None, None, None, None, None, None,
                              0, None) == -1                     # MI_E_NULL before anything else


def test_stream_registry_native_unit(tmp_path):
    """csrc/stream_registry.h (owner of mi_sinkhorn_dots' helper streams, keyed by (device, caller stream)): distinct
    callers get distinct resources, concurrent first use is race-free -- compiled and run with g++ (+ ThreadSanitizer
    when the toolchain has it), no HIP and no GPU involved."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    src = os.path.join(ROOT, "tests", "native", "test_stream_registry.cpp")
    exe = str(tmp_path / "test_stream_registry")
    base = ["g++", "-std=c++17", "-O1", "-pthread", src, "-o", exe]
    if subprocess.run(base[:4] + ["-fsanitize=thread"] + base[4:], capture_output=True).returncode != 0:
        subprocess.run(base, check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "stream_registry ok" in r.stdout, r.stdout + r.stderr


def _tuner(script):
    """Run [(op, arg, val), ...] through mi_debug_tuner_script (csrc/sk_tuner.h on a fresh state); returns the outputs."""
    from onnx_image_processing_amd import _native as N
    n = len(script)
    op = (ctypes.c_int * n)(*[o for o, _, _ in script])
    arg = (ctypes.c_int * n)(*[a for _, a, _ in script])
    val = (ctypes.c_double * n)(*[float(v) for _, _, v in script])
    out = (ctypes.c_int * n)()
    with N.debug_library() as lib:
        assert lib.mi_debug_tuner_script(n, op, arg, val, out) == 0
    return list(out)


def _decode(x):
    sched, slot, entry = x & 0xff, (x >> 8) & 0xff, (x >> 16) & 0xff
    return sched, (None if slot == 0xff else slot), (None if entry == 0xff else entry)


def test_stream_schedule_tuner_decision_logic(lib_path):
    """The decision logic of mi_sinkhorn_dots' stream-schedule tuner with INJECTED timings (no GPU): round-robin trials,
    minimum per schedule (one noisy sample does not pin a slow schedule), a trial abandoned between its two events still
    lets the window close (round 3's review: `finished` could never reach the trial count), trials of different shapes
    are never compared, a capture before the decision gets the unsplit schedule, pin / unpin, expiry."""
    BEGIN, FINISH, ABANDON, CURRENT, CAPTURE, SET = range(6)
    # nine eager calls of one shape: schedules 0 1 2 0 1 2 0 1 2, slots 0..8, undecided meanwhile
    out = _tuner([(CAPTURE, 448, 0)] + [(BEGIN, 448, 0)] * 9 + [(CURRENT, 448, 0), (BEGIN, 448, 0), (CAPTURE, 448, 0)])
    assert out[0] == 2                                                     # capture while undecided: unsplit
    assert [_decode(x) for x in out[1:10]] == [(i % 3, i, 0) for i in range(9)]
    assert out[10] == -1 and _decode(out[11]) == (0, None, None) and out[12] == 2   # all issued, none harvested yet
    # timings: schedule 1 is the fastest; one of its samples is hit by a noisy neighbour (10x) -- the minimum decides
    times = {0: [1.00, 1.01, 0.99], 1: [0.90, 9.0, 0.91], 2: [1.10, 1.12, 1.11]}
    fin = [(FINISH, (0 << 8) | i, times[i % 3][i // 3]) for i in range(9)]
    out = _tuner([(BEGIN, 448, 0)] * 9 + fin + [(CURRENT, 448, 0), (BEGIN, 448, 0), (CAPTURE, 448, 0), (CURRENT, 64, 0), (CAPTURE, 64, 0)])
    assert out[18] == 1 and _decode(out[19]) == (1, None, None) and out[20] == 1
    assert out[21] == -1 and out[22] == 2                                  # another shape: nothing decided, unsplit capture
    # a failed join before close_trial: that slot is abandoned, the window still closes on the other eight
    ops = [(BEGIN, 448, 0)] * 9 + [(ABANDON, 4, 0)] + [f for f in fin if f[1] != 4] + [(CURRENT, 448, 0)]
    assert _tuner(ops)[-1] == 1
    # every trial abandoned (events could not be recorded): the window closes on the eager default, not "never"
    assert _tuner([(BEGIN, 448, 0)] * 9 + [(ABANDON, i, 0) for i in range(9)] + [(CURRENT, 448, 0)])[-1] == 0
    # started == trials with fewer finished: undecided, eager calls run schedule 0, and late results still decide
    out = _tuner([(BEGIN, 448, 0)] * 9 + fin[:5] + [(CURRENT, 448, 0), (BEGIN, 448, 0)] + fin[5:] + [(CURRENT, 448, 0)])
    assert out[14] == -1 and _decode(out[15]) == (0, None, None) and out[-1] == 1
    # two shapes interleaved: each has its own window and decision (64 pairs: unsplit wins; 448: schedule 0)
    script, slots = [], {64: 0, 448: 0}
    for i in range(9):
        for b in (64, 448):
            script.append((BEGIN, b, 0))
    outs = _tuner(script)
    ent = {64: _decode(outs[0])[2], 448: _decode(outs[1])[2]}
    assert ent[64] != ent[448]
    fin2 = []
    for i in range(9):
        fin2.append((FINISH, (ent[64] << 8) | i, [0.30, 0.31, 0.25][i % 3]))
        fin2.append((FINISH, (ent[448] << 8) | i, [0.90, 1.00, 1.10][i % 3]))
    out = _tuner(script + fin2 + [(CURRENT, 64, 0), (CURRENT, 448, 0)])
    assert out[-2:] == [2, 0]
    # a fifth shape while four windows have trials in flight: runs unsplit untried; once a window has closed its entry
    # can be evicted (least recently used) and the new shape is tuned
    four = [(BEGIN, b, 0) for b in (64, 128, 256, 448)]
    out = _tuner(four + [(BEGIN, 512, 0)])
    assert _decode(out[-1]) == (2, None, None)
    close64 = [(BEGIN, 64, 0)] * 8 + [(FINISH, (0 << 8) | i, 1.0) for i in range(9)]
    out = _tuner(four + close64 + [(BEGIN, 512, 0), (CURRENT, 64, 0)])
    assert _decode(out[-2]) == (0, 0, 0) and out[-1] == -1                 # entry 0 re-used for the new shape
    # pin / unpin
    out = _tuner([(SET, 2, 0), (BEGIN, 448, 0), (CURRENT, 448, 0), (CAPTURE, 64, 0), (SET, -1, 0), (CURRENT, 448, 0), (BEGIN, 448, 0), (SET, 3, 0)])
    assert _decode(out[1]) == (2, None, None) and out[2] == 2 and out[3] == 2 and out[5] == -1
    assert _decode(out[6]) == (0, 0, 0) and out[7] == -1
    # expiry: after 8192 calls on a decision a new window opens; the old decision stays in force until it closes
    script = [(BEGIN, 448, 0)] * 9 + fin + [(BEGIN, 448, 0)] * 8191
    out = _tuner(script + [(BEGIN, 448, 0), (CURRENT, 448, 0)])
    assert all(_decode(x) == (1, None, None) for x in out[18:18 + 8191])
    assert _decode(out[-2]) == (0, 0, 0) and out[-1] == 1


def test_library_does_not_import_hip_memset(lib_path):
    """No entry point clears memory with hipMemset*: a memset call captured into a hipGraph became a node that zeroed its
    range on the first replay only on ROCm 7.2 (tools/graph_memset_probe.py); the library's clears
    are kernels (csrc/common.h mi_zero_async).  The import table of both libraries is the cheapest place to hold that."""
    import subprocess
    from onnx_image_processing_amd import _native
    for path in (lib_path, _native.DEBUG_LIB_PATH):
        syms = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
        assert "hipLaunchKernel" in syms or "hipModuleLaunchKernel" in syms or "__hipPushCallConfiguration" in syms
        assert not [ln for ln in syms.splitlines() if "hipMemset" in ln], path


def test_dots_form_refuses_small_epsilon(lib_path):
    """MI_DOTS_MIN_EPSILON: the packed Sinkhorn form drops the cost clamp and is refused below epsilon = 0.005 on the
    host, before any launch."""
    from onnx_image_processing_amd import _native as N, ops
    lib = N.load()
    fake = ctypes.create_string_buffer(4096)
    ptr = ctypes.cast(fake, ctypes.c_void_p).value
    assert lib.mi_sinkhorn_dots(ptr, ptr, ptr, 1, 8, 8, 8, 0.004, 1.0, 1.0, 5, ptr, ptr, None, ptr, 1 << 20, 0, None) == -3
    prm = N.MatchParams(3, 5, 512, 0.0, 7, 512, ptr, ptr, None, 1, 0.004, 1.0, 20, 100, 0.1)
    assert lib.mi_match_pairs_workspace_bytes(1, 480, 640, ctypes.byref(prm)) == 0
    prm.epsilon = 0.005
    assert lib.mi_match_pairs_workspace_bytes(1, 480, 640, ctypes.byref(prm)) > 0
    assert ops.DOTS_MIN_EPSILON == 0.005
    hdr = open(HEADER).read()
    assert "#define MI_DOTS_MIN_EPSILON 0.005" in hdr


def test_reference_import_lines_resolve_to_this_package(lib_path):
    """`pytorch_model.*` (the reference's package name) aliases onnx_image_processing_amd.pytorch_model.*: same module
    objects; the module names of feature_detection/__init__.py:4-9 exist; out-of-scope sub-packages are absent."""
    import importlib
    import onnx_image_processing_amd.pytorch_model.detector as real
    alias = importlib.import_module("pytorch_model.detector")
    assert alias is real
    from pytorch_model.descriptor.bad import SparseBAD                                       # noqa: F401
    from pytorch_model.feature_detection import ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix as A
    from pytorch_model.feature_detection.akaze_sparse_bad_sinkhorn_essential_matrix import (  # noqa: F401
        AKAZESparseBADSinkhornWithEssentialMatrix)
    from pytorch_model.feature_detection.shi_tomasi_angle_sparse_bad_sinkhorn_essential_matrix import (
        ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix as B)
    from pytorch_model.matching.outlier_filters import dustbin_margin_filter, probability_ratio_filter  # noqa: F401
    assert A is B
    with pytest.raises(ImportError):
        importlib.import_module("pytorch_model.vo")
    with pytest.raises(RuntimeError, match="no CPU path"):
        probability_ratio_filter(torch.rand(4, 4))


def test_argument_validation_of_the_round2_entries(lib_path):
    """Host-side MI_E_* checks of the entry points added in round 2 (no launch happens: safe without a GPU)."""
    lib = ctypes.CDLL(lib_path)
    buf = ctypes.create_string_buffer(256)
    p = ctypes.cast(buf, ctypes.c_void_p)
    q = ctypes.c_void_p(p.value + 64)
    vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    lib.mi_corner_response_u8.argtypes = [vp, ci, ci, ci, ci, vp, vp]
    assert lib.mi_corner_response_u8(None, 1, 8, 8, 3, p, None) == -1
    assert lib.mi_corner_response_u8(p, 1, 0, 8, 3, p, None) == -2
    assert lib.mi_corner_response_u8(p, 1, 8, 8, 2, p, None) == -3
    lib.mi_corner_response_balanced.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, vp]
    assert lib.mi_corner_response_balanced(None, 0, 1, 8, 8, 3, p, p, None) == -1
    assert lib.mi_corner_response_balanced(p, 1, 1, 8, 0, 3, p, p, None) == -2
    assert lib.mi_corner_response_balanced(p, 0, 1, 8, 8, 4, p, None, None) == -3
    assert lib.mi_corner_response_balanced(p, 0, 1, 8, 8, 3, p, ctypes.c_void_p(p.value + 2), None) == -5    # counter block alignment
    lib.mi_convert_u8_f32.argtypes = [vp, ctypes.c_longlong, vp, vp]
    assert lib.mi_convert_u8_f32(p, 0, p, None) == -2 and lib.mi_convert_u8_f32(None, 4, p, None) == -1
    lib.mi_sparse_bad_u8.argtypes = [vp, ci, ci, ci, vp, ci, vp, vp, ci, ci, cf, ci, vp, vp, vp, vp, vp]
    assert lib.mi_sparse_bad_u8(p, 1, 64, 64, p, 4, p, p, 100, 2, 0.0, 1, p, None, None, None, None) == -3   # pairs % 64
    assert lib.mi_sparse_bad_u8(p, 1, 64, 64, p, 4, p, p, 256, 0, 0.0, 1, p, p, None, None, None) == -3     # bits need HARD
    lib.mi_match_filter_masks.argtypes = [vp, ci, ci, ci, ci, cf, cf, vp, vp]
    assert lib.mi_match_filter_masks(p, 1, 4, 4, 0, 2.0, 0.3, p, None) == -3        # a margin test needs the dustbin column
    assert lib.mi_match_filter_masks(None, 1, 4, 4, 1, 2.0, 0.3, p, None) == -1
    lib.mi_akaze_scale.argtypes = [vp, ci, ci, ci, ci, cf, cf, cf, ci, vp, vp, vp, vp]
    assert lib.mi_akaze_scale(p, 1, 8, 8, 3, 0.05, 0.25, 0.001, 5, p, q, None, None) == -1     # l_out aliases l_in
    assert lib.mi_akaze_scale(p, 1, 8, 8, 0, 0.05, 0.25, 0.001, 5, q, q, None, None) == -3     # no iterations
    assert lib.mi_akaze_scale(p, 1, 8, 8, 3, 0.05, 0.25, 0.001, 4, q, q, None, None) == -3     # even NMS window
    assert lib.mi_akaze_scale(p, 1, 8, 8, 4, 0.05, 0.25, 0.001, 5, q, q, None, None) == -1     # unfused form needs tmp
    assert lib.mi_akaze_scale_fused(3, 5) == 1 and lib.mi_akaze_scale_fused(4, 5) == 0 and lib.mi_akaze_scale_fused(2, 9) == 0
    lib.mi_sinkhorn_dots_workspace_bytes.restype = ctypes.c_size_t
    small, big = lib.mi_sinkhorn_dots_workspace_bytes(8, 512, 512), lib.mi_sinkhorn_dots_workspace_bytes(9, 512, 512)
    assert small > big * 8 // 9          # up to 8 pairs the workspace also holds the single-launch form's hand-off area
    lib.mi_release_stream_resources.argtypes = [vp]


def test_bad_gather_schedule_is_lane_local_and_cuts_bank_conflicts(lib_path):
    """csrc/bad_plan_opt.h through mi_debug_bad_plan_passes (host only): on the reference's pair tables the scheduled
    gathers need far fewer LDS passes than the table as it stands, every lane keeps its own pairs (the entry point
    checks exec_pair % 64 == lane, which is what lets the kernel rebuild the canonical bit order), and the schedule
    is a deterministic function of the table."""
    from onnx_image_processing_amd import _native as N
    from onnx_image_processing_amd.pytorch_model.descriptor.bad import SparseBAD
    with N.debug_library() as lib:                # mi_debug_* live in the debug build only
        _gather_schedule_checks(lib, SparseBAD)


def _gather_schedule_checks(lib, SparseBAD):
    for pairs, bound in ((512, 200), (256, 110)):
        geom = np.ascontiguousarray(SparseBAD(num_pairs=pairs).pair_geom.numpy().astype(np.uint32))
        got = []
        for _ in range(2):
            canonical, scheduled = ctypes.c_int(0), ctypes.c_int(0)
            assert lib.mi_debug_bad_plan_passes(geom.ctypes.data, pairs, ctypes.byref(canonical), ctypes.byref(scheduled)) == 0
            got.append((canonical.value, scheduled.value))
        assert got[0] == got[1]
        assert pairs // 4 <= got[0][1] <= bound < got[0][0], (pairs, got[0])
    a, b = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.mi_debug_bad_plan_passes(None, 512, ctypes.byref(a), ctypes.byref(b)) == -1
    assert lib.mi_debug_bad_plan_passes(geom.ctypes.data, 100, ctypes.byref(a), ctypes.byref(b)) == -3


def _build_c_host(tmp_path):
    """tests/native/host_match_pairs.c with plain gcc (C99): the header is valid C, the host needs the HIP runtime only."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/lib/libamdhip64.so"):
        pytest.skip("gcc / libamdhip64 not available")
    exe = str(tmp_path / "host_match_pairs")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{os.path.join(ROOT, 'include')}",
           os.path.join(ROOT, "tests", "native", "host_match_pairs.c"), "-o", exe, "-L/opt/rocm/lib", "-lamdhip64", "-ldl",
           "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_host_compiles_against_the_header_only(tmp_path, lib_path):
    """A host without Python or torch: plain C99 + the HIP runtime, the library opened with dlopen.  Here (no GPU) it must
    compile warning-free and depend on nothing of this repository or of torch at link time; the GPU suite runs it."""
    import subprocess
    exe = _build_c_host(tmp_path)
    deps = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libamdhip64" in deps and "mi355x" not in deps and "torch" not in deps and "python" not in deps
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr
