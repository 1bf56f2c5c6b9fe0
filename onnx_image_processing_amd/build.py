"""Build the gfx950 shared library (C ABI of include/mi355x_match.h) in-tree.

    python -m onnx_image_processing_amd.build [--force]

hipcc cross-compiles without a GPU.  Output: onnx_image_processing_amd/lib/libmi355x_match.so -- the product: exports
exactly include/mi355x_match.h -- and lib/libmi355x_match_debug.so (the same ABI plus the kernel-variant test hooks of
include/mi355x_match_debug.h; tests/ and tools/ only).  Both are git-ignored and travel to the GPU box with the working
tree.  Objects are rebuilt only when a source or header is newer.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libmi355x_match.so")
# the same sources with -DMI_DEBUG_HOOKS: exports include/mi355x_match_debug.h as well; loaded by tests/ and tools/ only
DEBUG_LIB = os.path.join(LIBDIR, "libmi355x_match_debug.so")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")

SOURCES = ["corner.hip", "nms.hip", "topk.hip", "bad.hip", "bad_oriented.hip", "bad_dense.hip", "orient.hip", "cost.hip", "sinkhorn.hip", "sinkhorn_dots.hip", "mnn.hip", "akaze.hip", "akaze_stream.hip", "essential.hip", "detectors.hip", "match_pairs.hip"]
# sources that read a hook of csrc/hooks.h: compiled a second time for the debug library; every other object is shared
HOOKED = ["corner.hip", "topk.hip", "sinkhorn.hip", "sinkhorn_dots.hip", "akaze.hip", "bad_oriented.hip", "cost.hip", "mnn.hip"]
DEBUG_ONLY = ["hooks.hip"]
# -ffp-contract=off: the corner response must not fuse a*b+c (bit parity with the reference's
# op-by-op fp32); IEEE sqrt/div are hipcc's defaults and are relied upon.
# -fvisibility=hidden: only the MI_API declarations of include/*.h are exported (no C++-mangled internals).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wall",
         "-Wno-unused-function", f"-I{INCLUDE}"]
# per-file extras.  corner.hip: SLP packs the stencil's fp32 adds/muls into v_pk_* ops, which issue
# at half rate on gfx950 and need extra moves to pair operands (+18 % VALU slots measured).
# -save-temps=obj keeps corner's gfx950 assembly next to its object: tests/test_k1_isa.py lints the hand-scheduled
# region of the ticket kernel in it (CPU only).
EXTRA_FLAGS = {"corner.hip": ["-fno-slp-vectorize", "-save-temps=obj"], "akaze.hip": ["-fno-slp-vectorize"], "akaze_stream.hip": ["-fno-slp-vectorize"],
               "bad_oriented.hip": ["-fno-slp-vectorize"]}
# -Bsymbolic: calls between the library's own entry points bind inside the library (two builds of the same ABI can be
# loaded into one process -- product and debug -- without one's calls landing in the other)
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic",
              f"-Wl,--version-script={os.path.join(CSRC, 'exports.map')}"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Build both libraries (objects are rebuilt only when a source or header is newer); returns the product library."""
    os.makedirs(OBJDIR, exist_ok=True)
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))) + [os.path.abspath(__file__)]
    link_deps = [os.path.join(CSRC, "exports.map")]
    hipcc = _hipcc()
    # development hook for on-box parameter sweeps: MI_BUILD_DEFINES="-DAS_WAVES=16 ..." (forces a full rebuild)
    extra_defines = os.environ.get("MI_BUILD_DEFINES", "").split()
    force = force or bool(extra_defines)
    jobs = []
    objs, debug_objs = [], []

    def want(src: str, suffix: str, defines: list[str]) -> str:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", suffix + ".o"))
        if force or _newer(o, [s] + headers):
            extra = [f for f in EXTRA_FLAGS.get(src, []) if not (defines and f.startswith("-save-temps"))]   # temps: product only
            jobs.append([hipcc, *FLAGS, *extra, *defines, *extra_defines, "-c", s, "-o", o])
        return o

    for src in SOURCES:
        o = want(src, "", [])
        objs.append(o)
        debug_objs.append(want(src, ".dbg", ["-DMI_DEBUG_HOOKS"]) if src in HOOKED else o)
    for src in DEBUG_ONLY:
        debug_objs.append(want(src, ".dbg", ["-DMI_DEBUG_HOOKS"]))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJDIR)
        if r.returncode != 0:
            raise RuntimeError(f"build failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        list(ex.map(run, jobs))
    # Link to a temporary name and rename onto the final path: a process that waits for the library to appear (bench.py's
    # other ranks, a concurrent test run) must never dlopen a file the linker is still writing.  The debug library is
    # linked FIRST, so that the product library's appearance means both are complete.
    for lib, members in ((DEBUG_LIB, debug_objs), (LIB, objs)):
        if jobs or force or _newer(lib, members + link_deps):
            tmp = f"{lib}.{os.getpid()}.tmp"
            try:
                run([hipcc, *LINK_FLAGS, *members, "-o", tmp])
                os.replace(tmp, lib)                     # atomic on one filesystem
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
