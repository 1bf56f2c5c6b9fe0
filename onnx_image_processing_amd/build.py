"""Build the gfx950 shared library (C ABI of include/mi355x_match.h) in-tree.

    python -m onnx_image_processing_amd.build [--force]

hipcc cross-compiles without a GPU.  Output: onnx_image_processing_amd/lib/libmi355x_match.so
(git-ignored; it travels to the GPU box with the working tree).  Objects are rebuilt only when
a source or header is newer.
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
INCLUDE = os.path.join(os.path.dirname(PKG), "include")

SOURCES = ["corner.hip", "nms.hip", "topk.hip", "bad.hip", "bad_oriented.hip", "bad_dense.hip", "orient.hip", "cost.hip", "sinkhorn.hip", "sinkhorn_dots.hip", "mnn.hip", "akaze.hip", "essential.hip", "detectors.hip", "match_pairs.hip"]
# -ffp-contract=off: the corner response must not fuse a*b+c (bit parity with the reference's
# op-by-op fp32); IEEE sqrt/div are hipcc's defaults and are relied upon.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", f"-I{INCLUDE}"]
# per-file extras.  corner.hip: SLP packs the stencil's fp32 adds/muls into v_pk_* ops, which issue
# at half rate on gfx950 and need extra moves to pair operands (+18 % VALU slots measured).
EXTRA_FLAGS = {"corner.hip": ["-fno-slp-vectorize"]}


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
    os.makedirs(OBJDIR, exist_ok=True)
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))) + [os.path.abspath(__file__)]
    hipcc = _hipcc()
    # development hook for on-box parameter sweeps: MI_BUILD_DEFINES="-DAS_WAVES=16 ..." (forces a full rebuild)
    extra_defines = os.environ.get("MI_BUILD_DEFINES", "").split()
    force = force or bool(extra_defines)
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), *extra_defines, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"build failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _newer(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
