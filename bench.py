#!/usr/bin/env python3
"""Benchmark of the hot path: image-pairs/sec (640x480, K=512) -- BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs-per-gpu B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for N > 1: under torchrun the ranks exist already; a bare `python bench.py --gpus N` starts N fresh rank
processes itself (launch_ranks) before touching the GPU.  The N > 1 line records the world size the backend actually
formed (`ranks_seen`, `backend`) and every rank's own ms per step.  The process group is joined after two untimed
steps (join_group: a communicator created before the first step changed the runtime's hardware-queue mapping and cost the
Sinkhorn call its stream overlap), stdout carries nothing but the JSON line (RCCL's version banner goes to stderr), and
`MI_BENCH_FORCE_DIST=1 python bench.py` runs the N-rank control flow with a real RCCL group of ONE rank on a one-GPU box.

One "step" = one pass of the whole path (Shi-Tomasi -> NMS/top-k -> sparse BAD -> cost -> Sinkhorn -> mutual-NN match
extraction; the reference's MatchExtractionWrapper form) over a batch of B synthetic pairs per GPU that is already
resident in HBM as float32 (the reference's input type), followed (N > 1) by the RCCL gather of the match records to
rank 0.  Configuration = BASELINE.json configs[1] hyper-parameters (the export-CLI values) with
K=512; pairs are independent, so ranks hold different pairs (weak scaling).  Prints ONE JSON line on rank 0.

Beside `value` the line carries (N = 1 only, all measured after the timed region, none of them part of `value`):
  roofline      K1 corner response (the stencil north_star sets the 60 % target on), float32 input, 8 B/px
  u8_ingest     the same workload on uint8 frames resident in HBM (5 B/px K1), with its own roofline object
  p_materialised  the same workload with P written by forward() and read by the extractor (the reference's two modules)
  streamed      uint8 frames in pinned host memory, H2D on a copy stream double-buffered against compute
  latency       one pair per call and eight pairs per call (the reference harness's pattern,
                sample/image_matching.py:313-328): eager and hipGraph replay, submit -> results on the device
  cpu_baseline  the reference CPU path (oracle/torch_cpu.py, pinned to the reference's recorded outputs) on this
                box's host cores with the reference harness's 5 + 10 protocol, plus the live match-set parity
  side_workloads  BASELINE configs[2] (c3: 1920x1080, K=1024), configs[3] (c4: the AKAZE matcher) and the visual-odometry
                model (vo), 128 pairs per step, ten steps each: value, roofline of the dominant bandwidth-type kernel,
                per-call times and the workload's own reference CPU path (one pair per call); `--workload c3|c3dense|c4|vo`
                gives each a full line of its own, `--workload vo --stream` the one-pair-per-call form per rank
  sinkhorn_schedule  the stream schedule mi_sinkhorn_dots tuned itself to during the untimed steps
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
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
FP4_MFMA_PEAK_TOPS = 10000.0  # dense FP6 / FP4 MFMA peak (MI355X_MICROARCH.md; the 2:1-sparsity headline is not used): the
                              # packed descriptors' dot products run on v_mfma_f32_32x32x64_f8f6f4 (bit -> nibble 1.0 / 0.0, exact)


# ----------------------------------------------------------------------------------------------- timing plumbing
class StepClock:
    """Per-step timestamps on the launch stream: HIP events on a GPU, perf_counter on the CPU (the gloo dry run of the
    N > 1 control flow in tests/test_distributed_gloo.py)."""

    def __init__(self, steps: int, cuda: bool):
        self.cuda = cuda
        self.marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if cuda else []
        self.stamps: list[float] = []

    def mark(self, i: int) -> None:
        if self.cuda:
            self.marks[i].record()
        else:
            self.stamps.append(time.perf_counter())

    def per_step_ms(self) -> list[float]:
        if self.cuda:
            return [a.elapsed_time(b) for a, b in zip(self.marks, self.marks[1:])]
        return [(b - a) * 1e3 for a, b in zip(self.stamps, self.stamps[1:])]


class PipelinedGather:
    """The step's result gather with ONE collective in flight: step i issues its gather (gather_records_async) and
    returns the gathered records of step i - 1; drain() -- called inside the timed region, before its closing
    synchronise -- completes the last one.  Every step's records are gathered and waited for before the clock stops;
    what changes is that the slabs of step i travel while step i + 1 computes instead of in front of it.  With one
    process (no process group) the gather is the identity and nothing is deferred."""

    def __init__(self, total: int | None = None):
        self.pending = None
        self.total = total
        self.last = None

    def __call__(self, rec):
        from onnx_image_processing_amd import distributed as D
        handle = D.gather_records_async(rec, dst=0, total=self.total)
        if not dist.is_initialized():
            self.last = handle.wait()
            return self.last
        if self.pending is not None:
            self.last = self.pending.wait()
        self.pending = handle
        return self.last

    def drain(self):
        if self.pending is not None:
            self.last = self.pending.wait()
            self.pending = None
        return self.last


def join_group(step, drain, world: int, forced: bool) -> None:
    """Two untimed steps -- code objects loaded, torch's and the library's streams created -- and only then the process
    group (see main(): a communicator created before them costs the Sinkhorn's stream overlap)."""
    from onnx_image_processing_amd import distributed as D
    if world == 1 and not forced:
        return
    for _ in range(2):
        step()
    drain()
    torch.cuda.synchronize()
    D.init(force=forced)
    dist.barrier()


def run_timed(step, steps: int, warmup: int, world: int, device, sync, drain=None) -> tuple[float, list[float], object, float]:
    """The contract's timed region: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by a barrier + device
    synchronisation on both sides; returns (elapsed ms, MAX over ranks; per-step ms of this rank; last step's output;
    this rank's own elapsed ms up to its last synchronise, before the closing barrier).  drain: completes whatever the
    steps left in flight (PipelinedGather.drain) -- called after the warm-up and, inside the timed region, after the
    last step; its return value replaces the last step's output.
    `step()` returns what rank 0 needs (the gathered records); `sync()` is torch.cuda.synchronize on a GPU."""
    from onnx_image_processing_amd import distributed as D
    out = None
    for _ in range(warmup):
        out = step()
    if drain is not None:
        out = drain()
    sync()
    if dist.is_initialized():          # world > 1, or a forced group of one (MI_BENCH_FORCE_DIST=1: the one-GPU rehearsal)
        dist.barrier()
    sync()
    clock = StepClock(steps, torch.device(device).type == "cuda")
    t0 = time.perf_counter()
    clock.mark(0)
    for i in range(steps):
        out = step()
        clock.mark(i + 1)
    if drain is not None:
        out = drain()
    sync()
    own_ms = (time.perf_counter() - t0) * 1e3
    if dist.is_initialized():
        dist.barrier()
    elapsed_ms = (time.perf_counter() - t0) * 1e3
    return D.barrier_max_ms(elapsed_ms, device), clock.per_step_ms(), out, own_ms


_LINE_OUT = None


def claim_stdout() -> None:
    """Keep this process's stdout for the ONE JSON line: everything else that writes to file descriptor 1 from here on --
    RCCL prints a five-line version banner there when its first communicator comes up -- goes to stderr."""
    global _LINE_OUT
    if _LINE_OUT is None:
        sys.stdout.flush()
        _LINE_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(line: dict) -> None:
    out = _LINE_OUT or sys.stdout
    print(json.dumps(line), file=out, flush=True)


def step_stats(per_step_ms: list[float]) -> dict:
    return {"min": min(per_step_ms), "median": statistics.median(per_step_ms), "max": max(per_step_ms),
            "mean": statistics.fmean(per_step_ms)}


def pmc_traffic(kernel_prefix: str, pairs_per_gpu: int, workload: str = "bench"):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/*_<workload>_pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this script at the recorded pairs per GPU, reads
    doubled per the gfx950 note).  None when no profile of this batch size exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{workload}_pmc_traffic.json")))
    for path in reversed(files):
        try:
            prof = json.load(open(path))
            if int(prof.get("pairs_per_gpu", 256)) != pairs_per_gpu:
                continue
            for name, row in prof["kernels"].items():
                if name.startswith(kernel_prefix):
                    return row["total_MB"] * 1e6, os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def k1_roofline(ms_per_launch: float, bytes_per_px: float, images: int, kernel: str, prefix: str, pairs: int) -> dict:
    nbytes = bytes_per_px * images * H * W
    achieved = nbytes / (ms_per_launch * 1e-3) / 1e9
    traffic, src = pmc_traffic(prefix, pairs)
    return {"kernel": kernel, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src, "bytes_per_launch": nbytes,
            "bytes_per_pixel": bytes_per_px, "ms_per_launch": ms_per_launch}


# ----------------------------------------------------------------------------------------------- CPU baseline
def compare_match_sets(got: dict, want: dict, max_matches: int, mutual: dict | None = None, tol: float = 1e-4) -> int:
    """0 = same set and scores within tol; 1 = the sets differ only by a tie at the max_matches cut (both full, every
    match in one set but not the other scores within tol of the cut score, and -- when the reference's complete mutual
    set is given -- the match only `got` has is one of them with the same score); 2 = anything else."""
    if any(abs(got[k] - want[k]) > tol for k in set(got) & set(want)):
        return 2
    if set(got) == set(want):
        return 0
    if len(got) != max_matches or len(want) != max_matches:
        return 2
    cut = min(min(got.values()), min(want.values()))
    both = {**want, **got}
    if any(both[k] > cut + tol for k in set(got) ^ set(want)):
        return 2
    if mutual is not None and any(k not in mutual or abs(mutual[k] - got[k]) > tol for k in set(got) - set(want)):
        return 2
    return 1


def physical_cores() -> int:
    """Physical cores of this host (unique (package, core) pairs of /proc/cpuinfo; os.cpu_count() // 2 if unreadable)."""
    try:
        seen, pkg = set(), None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                pkg = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                seen.add((pkg, ln.split(":")[1].strip()))
        if seen:
            return len(seen)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def host_parallel(workers: int, pairs_per_worker: int) -> dict:
    """W one-thread processes of the reference CPU path, one pair per call each (oracle/cpu_worker.py): pairs are
    independent, so this is what the host's cores can do together.  The workers are fresh CPU-only child processes;
    all of them warm up, then time the same wall-clock window (released together)."""
    import subprocess
    worker = os.path.join(ROOT, "oracle", "cpu_worker.py")
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    procs = [subprocess.Popen([sys.executable, worker, str(5000 + 10 * i), str(pairs_per_worker), str(H), str(W), str(K)],
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=env) for i in range(workers)]
    try:
        for pr in procs:
            if pr.stdout.readline().strip() != "ready":
                raise RuntimeError("a cpu_baseline worker failed to start")
        t0 = time.perf_counter()
        for pr in procs:
            pr.stdin.write("go\n")
            pr.stdin.flush()
        rows = [json.loads(pr.stdout.readline()) for pr in procs]
        wall = time.perf_counter() - t0
    finally:
        for pr in procs:
            try:
                pr.stdin.close()
            except Exception:
                pass
            pr.wait(timeout=60)
    total = sum(r["pairs"] for r in rows)
    slowest = max(r["seconds"] for r in rows)
    return {"workers": workers, "threads_per_worker": 1, "pairs_per_worker": pairs_per_worker,
            "pairs_per_sec": total / slowest, "slowest_worker_s": slowest, "fastest_worker_s": min(r["seconds"] for r in rows),
            "wall_s_incl_release": wall, "mean_valid_matches_per_pair": sum(r["valid_matches"] for r in rows) / total,
            "what": "W single-thread processes of oracle/torch_cpu.py, one pair per call each, released together; "
                    "total pairs / slowest worker's time"}


def cpu_baseline(parity_pairs: int, gpu_records=None, parallel: bool = True) -> dict:
    """The reference CPU path on this host's cores: oracle/torch_cpu.py -- the same ATen CPU
    kernels in the reference's order, pinned to the reference's recorded outputs by tests/test_oracle_golden.py.
      sweep          torch intra-op threads in {1, 8, 16, 32, 64, physical cores} for one pair per call, {8, 32, physical}
                     for eight pairs per call, short protocol (1 warm-up + 3 timed calls) -- the reference relies on
                     torch's intra-op threads, and how many is the host's choice;
      value          the BEST single-process configuration of the sweep, re-timed with the reference harness's full
                     protocol (5 warm-up + 10 timed calls, mean; sample/image_matching.py:313-328); `cores` = its threads;
      host_parallel  W = physical cores one-thread worker processes, one pair per call each (pairs are independent).
    gpu_records: the (pairs, max_matches, 6) match records the GPU path produced for the same pairs; the numpy oracle's
    match sets for the first `parity_pairs` of them are compared after the clocks stop."""
    from onnx_image_processing_amd.synth import synth_batch
    from oracle import numpy_oracle as O
    from oracle.torch_cpu import TorchCpuPath, time_protocol
    t = np.load(os.path.join(ROOT, "onnx_image_processing_amd", "data", "bad_tables.npz"))
    kw = {k: v for k, v in CFG.items() if k not in ("num_pairs", "sampling_mode", "distance_type")}
    path = TorchCpuPath(t["box_512"], t["thr_512"], K, **kw)
    a, b = synth_batch(1000, max(8, parity_pairs), H, W)
    ta, tb = torch.from_numpy(a), torch.from_numpy(b)
    default_threads = torch.get_num_threads()
    cores = physical_cores()
    try:
        model = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:
        model = "unknown"
    sweep = []
    for pairs_per_call, counts in ((1, (1, 8, 16, 32, 64, cores)), (8, (8, 32, cores))):
        for th in sorted({c for c in counts if 1 <= c <= max(cores, 1)}):
            torch.set_num_threads(th)
            sec = time_protocol(lambda: path.match(ta[:pairs_per_call], tb[:pairs_per_call], **MNN), warmup=1, timed=3)
            sweep.append({"pairs_per_call": pairs_per_call, "threads": th, "ms_per_call": sec * 1e3,
                          "pairs_per_sec": pairs_per_call / sec})
    best = max(sweep, key=lambda r: r["pairs_per_sec"])
    torch.set_num_threads(best["threads"])
    n = best["pairs_per_call"]
    full = time_protocol(lambda: path.match(ta[:n], tb[:n], **MNN))          # 5 + 10, the reference harness's protocol
    torch.set_num_threads(default_threads)
    out = {"value": n / full, "unit": "image-pairs/sec", "cores": int(best["threads"]), "kind": "port",
           "sample": f"oracle/torch_cpu.py (torch-CPU restatement of the reference path, pinned to the reference's recorded "
                     f"outputs), 640x480 K=512 pairs seeds 1000..: best single-process configuration of the thread sweep "
                     f"({n} pair(s) per call on {best['threads']} thread(s)), protocol of sample/image_matching.py:313-328 "
                     f"(5 warm-up + 10 timed calls, mean); the sweep itself uses 1 + 3 calls per configuration",
           "best_configuration": {"pairs_per_call": n, "threads": int(best["threads"]), "ms_per_call": full * 1e3},
           "sweep": sweep,
           "host": {"cpu": model, "logical_cpus": os.cpu_count(), "physical_cores": cores,
                    "torch_default_threads": int(default_threads), "torch": torch.__version__}}
    if parallel:
        out["host_parallel"] = host_parallel(cores, 2)
    if gpu_records is not None and parity_pairs > 0:
        # match-set parity of the GPU's own output for these very pairs, against (a) the numpy oracle run here and (b) what
        # the REFERENCE produced for the same seeds (tests/golden/bench_seeds_matches.npz, recorded by make_golden.py
        # --round3-only from the imported reference): same matched coordinates, same validity, scores within 1e-4.  A
        # pair with more than max_matches mutual matches keeps the max_matches best: two scores closer than the 1e-4
        # bound that straddle that cut may legitimately swap ("cut ties", checked match by match); anything else is a
        # real difference.
        okw = {k: v for k, v in CFG.items() if k not in ("num_pairs", "sampling_mode")}
        fixture = None
        fpath = os.path.join(ROOT, "tests", "golden", "bench_seeds_matches.npz")
        if os.path.exists(fpath):
            fixture = np.load(fpath)
            if int(fixture["first_seed"]) != 1000 or (int(fixture["h"]), int(fixture["w"]), int(fixture["k"])) != (H, W, K):
                fixture = None
        tally = {"oracle": [0, 0, 0], "reference": [0, 0, 0]}
        worst, nmatch = 0.0, 0
        for i in range(parity_pairs):
            g = gpu_records[i]
            got = {tuple(map(float, g[j, 0:4])): float(g[j, 4]) for j in np.nonzero(g[:, 5] > 0.5)[0]}
            k1, k2, p = O.match_pair(a[i:i + 1], b[i:i + 1], t["box_512"], t["thr_512"], K, **okw)
            mk1, mk2, sc, valid, _ = O.mnn_extract(p, k1, k2, **MNN)
            want = {(*map(float, mk1[0, j]), *map(float, mk2[0, j])): float(sc[0, j]) for j in np.nonzero(valid[0])[0]}
            nmatch += len(want)
            worst = max([worst] + [abs(want[k] - got[k]) for k in set(want) & set(got)])
            tally["oracle"][compare_match_sets(got, want, MNN["max_matches"])] += 1
            if fixture is not None and i < int(fixture["pairs"]):
                fv = fixture["mvalid"][i]
                ref = {(*map(float, fixture["mk1"][i, j]), *map(float, fixture["mk2"][i, j])): float(fixture["mscores"][i, j])
                       for j in np.nonzero(fv)[0]}
                mutual = {tuple(map(float, r[:4])): float(r[4]) for r in fixture["mutual"][i][:int(fixture["n_mutual"][i])]}
                tally["reference"][compare_match_sets(got, ref, MNN["max_matches"], mutual)] += 1
        out["parity"] = {"checker": "oracle/numpy_oracle.py", "pairs_checked": parity_pairs,
                         "pairs_with_identical_match_set": tally["oracle"][0],
                         "pairs_differing_only_by_ties_at_the_max_matches_cut": tally["oracle"][1],
                         "pairs_differing_otherwise": tally["oracle"][2],
                         "matches_checked": nmatch, "max_abs_score_diff": worst, "bound": 1e-4}
        if fixture is not None:
            out["parity"]["vs_reference_fixture"] = {
                "fixture": "tests/golden/bench_seeds_matches.npz (the imported reference's match sets for seeds 1000..1063)",
                "pairs_checked": sum(tally["reference"]), "pairs_with_identical_match_set": tally["reference"][0],
                "pairs_differing_only_by_ties_at_the_max_matches_cut": tally["reference"][1],
                "pairs_differing_otherwise": tally["reference"][2]}
    return out


# ----------------------------------------------------------------------------------------------- N = 1 extras
def measure_latency(model, img1, img2, a8, b8, iters: int = 200) -> dict:
    """One call = submit -> results complete on the device (host synchronised after every call), the reference
    harness's pattern (sample/image_matching.py:313-328).  Three forms of the same forward, outputs identical:
      module       the nn.Module path (about 20 C-ABI calls, one launch per stage and image);
      single_call  ONE C-ABI call, mi_match_pairs: both images per launch, top-k by merge-rank sort, the 20 Sinkhorn
                   iterations as one persistent launch (bands exchange column sums as tagged granules);
      single_call_u8  the same on uint8 frames (mi_match_pairs_u8).
    Each eager and replayed as one hipGraph."""
    from onnx_image_processing_amd.graph import GraphedModule

    def timed(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    def one_call(x, y):
        return model.forward_single_call(x, y)

    out = {}
    for b in (1, 8):
        f1, f2 = img1[:b].contiguous(), img2[:b].contiguous()
        u1, u2 = torch.from_numpy(a8[:b]).to(img1.device), torch.from_numpy(b8[:b]).to(img1.device)
        row = {}
        for name, fn, x, y in (("module", model, f1, f2), ("single_call", one_call, f1, f2), ("single_call_u8", one_call, u1, u2)):
            eager = timed(lambda: fn(x, y))
            want = [t.clone() for t in fn(x, y)]
            graphed = GraphedModule(fn, x, y)
            graph = timed(graphed.graph.replay)
            # what the timed replays (host synchronised after each, other launches in between) left in the graph's output
            # buffers is the eager result, bit for bit: a replay that returned all-invalid matches would time the same
            same = all(torch.equal(g, w) for g, w in zip(graphed.static_outputs, want))
            if not same:
                raise RuntimeError(f"latency/{name}/{b}: the replayed graph's outputs differ from the eager call's")
            row[name] = {"eager_ms": eager, "graph_ms": graph, "eager_pairs_per_sec": b / (eager * 1e-3),
                         "graph_pairs_per_sec": b / (graph * 1e-3), "graph_equals_eager": True,
                         "valid_matches": int(want[3].sum().item())}
        if b == 1:
            # the reference harness's own situation (sample/image_matching.py:313-328): frames are HOST arrays and the
            # results are wanted on the host -- uint8 frames from pinned memory in, keypoint / match records out
            h1, h2 = torch.from_numpy(a8[:1]).pin_memory(), torch.from_numpy(b8[:1]).pin_memory()
            graphed8 = GraphedModule(one_call, u1, u2)
            rec_host = torch.empty((1, 100, 6), dtype=torch.float32).pin_memory()

            def from_host():
                graphed8.static_inputs[0].copy_(h1, non_blocking=True)
                graphed8.static_inputs[1].copy_(h2, non_blocking=True)
                graphed8.graph.replay()
                o = graphed8.static_outputs
                rec_host.copy_(torch.cat([o[0], o[1], o[2].unsqueeze(-1), o[3].float().unsqueeze(-1)], -1), non_blocking=True)
            row["single_call_u8_from_host"] = {"graph_ms": timed(from_host), "what": "H2D of both uint8 frames + replay + D2H "
                                               "of the 100 match records, host synchronised after every call"}
        out[f"pairs_per_call_{b}"] = row
    out["what"] = ("MatchExtractionWrapper(ShiTomasiSparseBADSinkhornMatcher), frames resident in HBM, host synchronised "
                   f"after every call, mean of {iters} calls; forms: see bench.py measure_latency")
    return out


def measure_two_step(model, img1: torch.Tensor, img2: torch.Tensor, steps: int) -> dict:
    """The default workload with the (K+1) x (K+1) matrix P WRITTEN by the matcher's forward() and read back by the
    extractor -- feature_detection/match_extraction_wrapper.py:82-113 taken literally (mi_sinkhorn_dots with a P output, then
    mi_mnn_extract on it) instead of the matches straight from the Sinkhorn solution.  Same records (asserted in the GPU
    suite); 1.8 KB more HBM traffic per matrix row."""
    from onnx_image_processing_amd import distributed as D
    B = img1.shape[0]
    fused = model.fuse_extraction
    model.fuse_extraction = False
    try:
        for _ in range(3):
            rec = D.pack_records(*model(img1, img2))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            rec = D.pack_records(*model(img1, img2))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
    finally:
        model.fuse_extraction = fused
    return {"value": B / (ms * 1e-3), "unit": "image-pairs/sec", "ms_per_step": ms, "steps": steps,
            "mean_valid_matches_per_pair": float(rec[..., 5].sum().item()) / B,
            "what": "float32 frames resident in HBM; the matcher's forward() writes P (B, K+1, K+1), MutualNearestNeighborMatcher reads it"}


def measure_u8_and_streamed(model, a8: np.ndarray, b8: np.ndarray, steps: int) -> tuple[dict, dict]:
    """(u8_ingest, streamed): the default workload on uint8 frames -- resident in HBM, and streamed from pinned host
    buffers (two slots; the copy of step i+1 runs on a copy stream while step i computes)."""
    from onnx_image_processing_amd import _native, distributed as D
    dev = torch.device("cuda", torch.cuda.current_device())
    B = a8.shape[0]
    host = torch.from_numpy(np.stack([a8, b8])).pin_memory()                   # (2, B, 1, H, W) uint8
    slots = [torch.empty_like(host, device=dev) for _ in range(2)]
    slots[0].copy_(host)
    slots[1].copy_(host)
    torch.cuda.synchronize()

    def step(s):
        return D.pack_records(*model(slots[s][0], slots[s][1]))

    for _ in range(3):
        step(0)
    torch.cuda.synchronize()
    _native.enable_timing(True, only={"mi_corner_response_pair"})
    t0 = time.perf_counter()
    for i in range(steps):
        rec = step(i & 1)
    torch.cuda.synchronize()
    resident_ms = (time.perf_counter() - t0) * 1e3 / steps
    k1 = _native.timings_ms()["mi_corner_response_pair"]
    _native.enable_timing(False)
    u8 = {"value": B / (resident_ms * 1e-3), "unit": "image-pairs/sec", "ms_per_step": resident_ms, "steps": steps,
          "what": "the default workload on uint8 frames resident in HBM (mi_corner_response_pair / mi_sparse_bad_pair on uint8 pixels)",
          "mean_valid_matches_per_pair": float(rec[..., 5].sum().item()) / B,
          "roofline": k1_roofline(float(np.mean(k1)), 5.0, 2 * B, "corner_stream_kernel<3,5,uint8> (mi_corner_response_pair: both images of every pair in one launch)",
                                  "corner_stream_kernel<3,5,true>", B)}

    main = torch.cuda.current_stream()
    copier = torch.cuda.Stream()

    def streamed(host_t, slot_ts, n_steps):
        """n_steps steps with the H2D copy of step i+1 under the compute of step i; -> (ms per step, last records)"""
        copied = [torch.cuda.Event() for _ in range(2)]
        done = [torch.cuda.Event() for _ in range(2)]

        def loop(n):
            rec = None
            for i in range(n):
                s = i & 1
                with torch.cuda.stream(copier):
                    if i >= 2:
                        copier.wait_event(done[s])                              # the step that last used this slot is finished
                    slot_ts[s].copy_(host_t, non_blocking=True)
                    copied[s].record(copier)
                main.wait_event(copied[s])
                rec = D.pack_records(*model(slot_ts[s][0], slot_ts[s][1]))
                done[s].record(main)
            return rec

        loop(4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rec = loop(n_steps)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / n_steps, rec

    ms, rec = streamed(host, slots, steps)
    nbytes = host.numel()
    out = {"value": B / (ms * 1e-3), "unit": "image-pairs/sec", "ms_per_step": ms, "steps": steps,
           "h2d_bytes_per_step": nbytes, "pcie_GBps_achieved": nbytes / (ms * 1e-3) / 1e9,
           "what": "uint8 frames in pinned host memory, H2D inside the timed region on a copy stream, double-buffered "
                   "against compute (two device slots); 0.61 MB per pair instead of 2.46 MB for float32 frames",
           "mean_valid_matches_per_pair": float(rec[..., 5].sum().item()) / B}
    # the same with float32 frames (what the reference's hosts hand over), for comparison: four times the bytes
    del slots
    fsteps = max(4, steps // 5)
    host32 = torch.empty(host.shape, dtype=torch.float32).pin_memory()
    host32.copy_(host)
    slots32 = [torch.empty_like(host32, device=dev) for _ in range(2)]
    ms32, _ = streamed(host32, slots32, fsteps)
    out["float32_frames"] = {"value": B / (ms32 * 1e-3), "ms_per_step": ms32, "steps": fsteps,
                             "h2d_bytes_per_step": host32.numel() * 4,
                             "pcie_GBps_achieved": host32.numel() * 4 / (ms32 * 1e-3) / 1e9}
    return u8, out


# ----------------------------------------------------------------------------------------------- configs[2] / [3] / VO
SIDE_WORKLOADS = ("c3", "c3dense", "c4", "vo")
VO_CFG = dict(block_size=5, num_pairs=512, binarize=True, soft_binarize=False, sinkhorn_iterations=20, epsilon=0.05,
              unused_score=1.0, distance_type="l2", nms_radius=5, score_threshold=0.0, normalize_descriptors=True)
# export_akaze_sparse_bad_sinkhorn.py defaults: 256 pairs, no binarisation, NMS radius 3
C4_CFG = dict(num_pairs=256, binarize=False, sinkhorn_iterations=20, epsilon=0.05, unused_score=1.0, distance_type="l2",
              nms_radius=3, score_threshold=0.0, normalize_descriptors=True, sampling_mode="nearest")
CAM_K = [[500.0, 0.0, 320.0], [0.0, 500.0, 240.0], [0.0, 0.0, 1.0]]


def side_model(name: str):
    """(module, h, w, k, description, (timed entry point, kernel label, bytes per pixel, profile prefix, images per launch
    as a multiple of the pairs per step))"""
    from onnx_image_processing_amd.pytorch_model.feature_detection import (AKAZESparseBADSinkhornMatcher,
                                                                           ShiTomasiSparseBADSinkhornMatcher)
    k1 = ("mi_corner_response_pair", "corner_stream_kernel<3,4> (mi_corner_response_pair: both images of every pair in one launch)", 8.0, "corner_stream_kernel<3,4,false>", 2)
    if name == "c3":
        return (ShiTomasiSparseBADSinkhornMatcher(max_keypoints=1024, **CFG), 1080, 1920, 1024,
                "Shi-Tomasi sparse pipeline, 1920x1080, K=1024 (BASELINE configs[2])", k1)
    if name == "c3dense":
        # BASELINE configs[2] in its "dense BAD cost matrix" reading: the reference's ShiTomasiBADSinkhornMatcher
        # (feature_detection/shi_tomasi_bad_sinkhorn.py:162-219) -- NMS / top-k WITHOUT border margin, descriptors = the dense
        # response map sampled at the keypoints (evaluated there exactly; the 4.2 GB map is never built), K x K cost on
        # the MFMA path.  Pinned to the reference at 640x480 by tests/golden/dense_c3_480x640_k512.npz.
        from onnx_image_processing_amd.pytorch_model.feature_detection import ShiTomasiBADSinkhornMatcher
        dcfg = {kk: v for kk, v in CFG.items() if kk != "sampling_mode"}
        return (ShiTomasiBADSinkhornMatcher(max_keypoints=1024, **dcfg), 1080, 1920, 1024,
                "ShiTomasiBADSinkhornMatcher (dense-BAD variant: no border margin, responses at the keypoints), 1920x1080, "
                "K=1024, P=512 hard bits (BASELINE configs[2], dense reading)", k1)
    if name == "vo":
        # the visual-odometry model (sample/visual_odometry.py:520-545 runs it once per frame
        # pair): Shi-Tomasi(5) + angle at the keypoints + rotation-aware BAD + Sinkhorn + essential-matrix head, Angle
        # export-CLI values, pinned to the reference by tests/golden/angle_vo_480x640_k512.npz
        from onnx_image_processing_amd.pytorch_model.feature_detection import ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix
        return (ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix(K=torch.tensor(CAM_K), max_keypoints=K, **VO_CFG), H, W, K,
                "ShiTomasiAngleSparseBADSinkhornWithEssentialMatrix (the VO model: Shi-Tomasi(5) + keypoint angles + oriented "
                "BAD(512, hard) + Sinkhorn(20, eps 0.05) + essential-matrix head), 640x480, K=512, Angle export-CLI values; "
                "per pair what sample/visual_odometry.py:520-613 consumes: E and the 100 best mutual matches "
                "(match_and_essential: both straight from the Sinkhorn solution, P not written)",
                ("mi_corner_response_pair", "corner_stream_kernel<5,4,float> (block 5: the streaming LDS-DMA kernel; both images of every pair in one launch)", 8.0, "corner_stream_kernel<5,4,false>", 2))
    # c4: one scale per launch for BOTH images of every pair (2 B images): reads the previous scale's image, writes the
    # diffused image and the scale's score map -- 12 B/px (the middle scale, mi_akaze_scale; the first reads two batches,
    # the last folds the selection across scales in: 4 + 4 + 4 + 8 + 1 B/px)
    return (AKAZESparseBADSinkhornMatcher(max_keypoints=K, **C4_CFG), H, W, K,
            "AKAZE(3 scales x 3 steps) + oriented sparse BAD(256, raw) + Sinkhorn(20, eps 0.05), 640x480, K=512 "
            "(BASELINE configs[3], AKAZE export-CLI values)",
            ("mi_akaze_scale", "akaze_stream_kernel<3,2,-1> (mi_akaze_scale: 3 diffusion steps + Hessian + NMS per launch, rolling "
                               "window; both images of every pair in one launch)", 12.0, "akaze_stream_kernel<3,2,-1>", 2))


def cpu_baseline_side(name: str, threads: int | None = None) -> dict:
    """The reference CPU path of a side workload on this host's cores: its oracle/torch_cpu.py twin (pinned to the
    reference's recorded outputs by tests/test_oracle_golden.py::test_torch_cpu_restatement_*), ONE pair per call -- the
    reference's own usage (sample/image_matching.py:313-328) -- 1 warm-up + 3 timed calls (a bounded sample: 2-10 s)."""
    from onnx_image_processing_amd.synth import synth_batch
    from oracle.torch_cpu import TorchCpuAkazePath, TorchCpuPath, TorchCpuVoPath, time_protocol
    t = np.load(os.path.join(ROOT, "onnx_image_processing_amd", "data", "bad_tables.npz"))
    drop = ("num_pairs", "sampling_mode", "distance_type")
    if name in ("c3", "c3dense"):
        h, w = 1080, 1920
        kw = {k: v for k, v in CFG.items() if k not in drop}
        if name == "c3dense":
            kw["border_margin"] = 0          # shi_tomasi_bad_sinkhorn.py:200-205; the twin's sparse box means stand in for the dense map
        path = TorchCpuPath(t["box_512"], t["thr_512"], 1024, **kw)
        twin = "TorchCpuPath" + (" (sparse box means in place of the dense response map: a LOWER bound of the reference's cost)" if name == "c3dense" else "")
    elif name == "vo":
        h, w = H, W
        path = TorchCpuVoPath(t["box_512"], t["thr_512"], K, np.asarray(CAM_K, np.float32), **{k: v for k, v in VO_CFG.items() if k not in drop})
        twin = "TorchCpuVoPath"
    else:
        h, w = H, W
        path = TorchCpuAkazePath(t["box_256"], t["thr_256"], K, **{k: v for k, v in C4_CFG.items() if k not in drop})
        twin = "TorchCpuAkazePath"
    a, b = synth_batch(1000, 1, h, w)
    ta, tb = torch.from_numpy(a), torch.from_numpy(b)
    default_threads = torch.get_num_threads()
    # the main cpu_baseline's sweep finds torch's intra-op sweet spot at ~32 threads on the pool's hosts (its default, one
    # thread per logical CPU, is 2-3x slower): used here when the caller has no sweep result to hand over
    torch.set_num_threads(int(threads) if threads else max(1, min(32, physical_cores())))
    try:
        sec = time_protocol(lambda: path.match(ta, tb, **MNN), warmup=1, timed=3)
        used = torch.get_num_threads()
    finally:
        torch.set_num_threads(default_threads)
    return {"value": 1.0 / sec, "unit": "image-pairs/sec", "cores": int(used), "kind": "port",
            "sample": f"oracle/torch_cpu.py {twin}, one {w}x{h} pair per call (seed 1000), 1 warm-up + 3 timed calls, mean",
            "ms_per_call": sec * 1e3}


def run_side(name: str, B: int, steps: int, warmup: int, rank: int, world: int, dev, two_step: bool = False,
             forced: bool = False, extras: bool = True, stream: bool = False):
    """One side workload through the same modules as the default line: returns the JSON line (rank 0; None elsewhere).
    stream: the VO model as its host uses it (sample/visual_odometry.py:520-545) -- ONE frame pair per call per rank,
    replayed as a hipGraph, the host synchronised after every call, the call's record (100 matches + E) gathered to
    rank 0 pipelined by one call; `value` = calls per second over all ranks."""
    from onnx_image_processing_amd import _native, distributed as D
    from onnx_image_processing_amd.pytorch_model.feature_detection import MatchExtractionWrapper
    from onnx_image_processing_amd.synth import synth_batch
    base, h, w, k, what, roof = side_model(name)
    begin, _ = D.shard_range(B * world, rank, world)
    a, b = synth_batch(1000 + begin, B, h, w)
    img1, img2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    del a, b
    gather = PipelinedGather()
    if name == "vo":
        model = base.to(dev)

        def records(x, y):                            # per pair: 100 match records + the 3 x 3 essential matrix (two more rows)
            mk1, mk2, sc, valid, e = model.match_and_essential(x, y, MNN["max_matches"], MNN["threshold"])
            erows = torch.nn.functional.pad(e.reshape(-1, 9), (0, 3)).reshape(-1, 2, D.RECORD_FIELDS)
            return torch.cat([D.pack_records(mk1, mk2, sc, valid), erows], dim=1)
    else:
        model = MatchExtractionWrapper(base, max_matches=MNN["max_matches"], match_threshold=MNN["threshold"]).to(dev)
        model.fuse_extraction = not two_step

        def records(x, y):
            return D.pack_records(*model(x, y))

    per_step = B
    if stream:
        if name != "vo":
            raise SystemExit("--stream is the VO model's one-pair-per-call pattern: use it with --workload vo")
        from onnx_image_processing_amd.graph import GraphedModule
        graphed = GraphedModule(records, img1[:1].contiguous(), img2[:1].contiguous())
        per_step, cursor = 1, [0]

        def compute():                                # the next resident frame pair through the replayed model, host-synchronised
            i = cursor[0] % B
            cursor[0] += 1
            out = graphed(img1[i:i + 1], img2[i:i + 1])
            torch.cuda.synchronize()
            return out

        def step():
            return gather(compute())
    else:
        def compute():
            return records(img1, img2)

        def step():
            return gather(compute())

    join_group(step, gather.drain, world, forced)
    _native.enable_timing(True, only={roof[0]})
    elapsed_ms, per_step_ms, out, own_ms = run_timed(step, steps, warmup, world, dev, torch.cuda.synchronize, gather.drain)
    facts = world_facts(own_ms, steps, dev)
    timed = _native.timings_ms().get(roof[0], [])
    timed = timed[-len(timed) * steps // (steps + warmup):] if timed else timed     # drop the warm-up calls
    _native.enable_timing(True)
    for _ in range(3):
        step()
    gather.drain()
    per_call = _native.timings_ms()
    _native.enable_timing(False)
    no_gather_ms = None
    if stream:                                        # the same loop without the gather: what the per-call collective costs
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            compute()
        no_gather_ms = (time.perf_counter() - t0) * 1e3 / steps
    line = None
    if rank == 0:
        ms = elapsed_ms / steps
        cfg = {"workload": what + (" -- STREAM form: one pair per call per rank, hipGraph replay, host synchronised after every "
                                   "call, the record gathered to rank 0 per call" if stream else ""),
               "pairs_per_gpu_per_step": per_step}
        if name == "vo":
            cfg["finite_essential_matrices"] = int(torch.isfinite(out[:, -2:]).all(dim=(1, 2)).sum().item())
            cfg["mean_valid_matches_per_pair"] = float(out[:, :-2, 5].sum().item()) / (per_step * world)
        else:
            cfg["mean_valid_matches_per_pair"] = float(out[..., 5].sum().item()) / (per_step * world)
        line = {
            "metric": f"image-pairs/sec ({w}x{h}, K={k})", "value": per_step * world / (ms * 1e-3), "unit": "image-pairs/sec",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": cfg,
            "step_ms": step_stats(per_step_ms), **facts,
            "kernels": {kk: {"ms_per_step": float(np.sum(v)) / 3, "calls_per_step": len(v) / 3} for kk, v in per_call.items()}}
        if stream:
            line["stream"] = {"calls_per_sec_per_rank": [1e3 / x for x in facts["ms_per_step_per_rank"]],
                              "ms_per_call_with_gather": ms, "ms_per_call_without_gather_rank0": no_gather_ms,
                              "gather_cost_ms_per_call": ms - no_gather_ms,
                              "record_bytes_per_call": (MNN["max_matches"] + 2) * D.RECORD_FIELDS * 4}
        if timed and not stream:
            nbytes = roof[2] * B * roof[4] * h * w
            t_ms = float(np.mean(timed))
            traffic, tsrc = pmc_traffic(roof[3], B, name)
            if name == "vo" and world == 1 and extras:
                # the model's own use: ONE frame pair per call (sample/visual_odometry.py:520-545), host synchronised after each
                from onnx_image_processing_amd.graph import GraphedModule
                one = (img1[:1].contiguous(), img2[:1].contiguous())

                def timed_calls(fn, iters=100):
                    for _ in range(10):
                        fn()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(iters):
                        fn()
                        torch.cuda.synchronize()
                    return (time.perf_counter() - t0) / iters * 1e3
                eager = timed_calls(lambda: model(*one))
                want = [t.clone() for t in model(*one)]
                graphed = GraphedModule(model, *one)
                graph_ms = timed_calls(graphed.graph.replay)
                if not all(torch.equal(g, w_) for g, w_ in zip(graphed.static_outputs, want)):
                    raise RuntimeError("latency_one_pair: the replayed graph's outputs differ from the eager call's")
                line["latency_one_pair"] = {"eager_ms": eager, "graph_ms": graph_ms, "graph_equals_eager": True,
                                            "what": "one 640x480 pair per call through the VO model (forward: P and E), host synchronised after every call"}
            line["roofline"] = {"kernel": roof[1], "bound": "hbm", "achieved": nbytes / (t_ms * 1e-3) / 1e9,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "traffic": traffic, "traffic_source": tsrc, "bytes_per_launch": nbytes,
                                "bytes_per_pixel": roof[2], "ms_per_launch": t_ms}
    del img1, img2, model
    return line


def side_workload(args, rank, world, dev) -> None:
    """`--workload c3|c3dense|c4|vo`: BASELINE configs[2]/[3] and the VO model as lines of their own (value, per-call
    times, the roofline of the dominant bandwidth-type kernel, and -- N = 1 -- the reference CPU path timed beside it)."""
    line = run_side(args.workload, args.pairs_per_gpu, args.steps, args.warmup, rank, world, dev, two_step=args.two_step,
                    forced=getattr(args, "_forced_group", False), extras=not args.no_extras, stream=args.stream)
    if rank == 0:
        if world == 1 and args.cpu_pairs > 0 and not args.stream:
            print(f"[bench] cpu_baseline ({args.workload})", file=sys.stderr, flush=True)
            line["cpu_baseline"] = cpu_baseline_side(args.workload)
        emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------- N > 1 launcher
def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv: list[str]) -> int:
    """`python bench.py --gpus N` outside torchrun: THIS process -- which has made no GPU call (importing torch does
    not initialise HIP) and never will -- starts N fresh rank processes of this script, one per GPU, with the torchrun
    environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), relays rank 0's JSON line on stdout
    and returns non-zero if any rank fails.  Child processes, never an exec: a process that has touched the GPU must
    not be replaced (the pool's rule), and children started before any GPU call inherit no HIP state.  stderr of the
    ranks passes through.  When one rank dies the others are terminated (they would wait in a collective forever)."""
    import subprocess
    port = int(os.environ.get("MASTER_PORT", "0")) or _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's stdout is drained WHILE the ranks run (a reader thread): a rank blocked writing into a full pipe would
    # never exit and this loop would poll forever
    import threading
    chunks: list[str] = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    pending = set(range(n))
    kill_at = None                                                   # after a failure: terminate, then kill what is left
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and kill_at is None:
                print(f"[bench] rank {r} exited with code {code}", file=sys.stderr, flush=True)
                rc = rc or code or 1
                for q in pending:                                    # the exact children started above, by handle
                    procs[q].terminate()
                kill_at = time.time() + 10.0
        if pending and kill_at is not None and time.time() > kill_at:
            for q in pending:                                        # a rank stuck in a collective ignores SIGTERM
                print(f"[bench] rank {q} did not stop after SIGTERM: killing it", file=sys.stderr, flush=True)
                procs[q].kill()
            kill_at = time.time() + 3600.0
        if pending:
            time.sleep(0.05)
    reader.join(timeout=10.0)
    out = "".join(chunks)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    if rc == 0 and len(lines) != 1:
        print(f"[bench] expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr, flush=True)
        rc = 1
    for ln in lines:
        print(ln, flush=True)
    return rc


def world_facts(elapsed_ms_this_rank: float, steps: int, device) -> dict:
    """What the process group actually formed (not what --gpus asked for) and every rank's own ms per step."""
    if not dist.is_initialized():
        return {"ranks_seen": 1, "backend": None, "ms_per_step_per_rank": [elapsed_ms_this_rank / steps]}
    t = torch.tensor([elapsed_ms_this_rank / steps], dtype=torch.float64, device=device)
    every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(every, t)
    return {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(),
            "ms_per_step_per_rank": [float(x.item()) for x in every]}


def dry_run(args, rank: int, world: int) -> None:
    """--dry-run: the launcher, rendezvous, sharding, timed region, gather and output line with the per-pair compute
    STUBBED (deterministic records, no GPU, backend gloo).  It exists so that the N > 1 start-up path is tested on CPU
    (tests/test_distributed_gloo.py); its line says so and is not a measurement."""
    from onnx_image_processing_amd import distributed as D
    B = args.pairs_per_gpu
    begin, end = D.shard_range(B * world, rank, world)
    g = torch.Generator().manual_seed(1000 + begin)
    rec = torch.rand((end - begin, MNN["max_matches"], D.RECORD_FIELDS), generator=g)
    rec[..., 5] = 1.0
    rec[:, :, 0] = torch.arange(begin, end, dtype=torch.float32)[:, None]      # the global pair index, for the order check

    if os.environ.get("MI_BENCH_DRY_RUN_FAIL_RANK") == str(rank):              # test hook of the dry run only: a rank that dies
        raise SystemExit(3)

    gather = PipelinedGather(total=B * world)

    def step():
        time.sleep(0.001 * (rank + 1))
        return gather(rec)

    elapsed_ms, per_step, out, own_ms = run_timed(step, args.steps, args.warmup, world, "cpu", lambda: None, gather.drain)
    facts = world_facts(own_ms, args.steps, "cpu")
    if rank == 0:
        ms = elapsed_ms / args.steps
        ordered = bool(torch.equal(out[:, 0, 0], torch.arange(B * world, dtype=torch.float32)))
        emit({"metric": "image-pairs/sec (640x480, K=512)", "value": B * world / (ms * 1e-3),
                          "unit": "image-pairs/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "DRY RUN: stubbed compute on CPU over gloo -- not a measurement",
                          "dry_run": True, "gathered_in_global_pair_order": ordered,
                          "config": {"workload": "launcher / rendezvous / gather control flow only",
                                     "pairs_per_gpu_per_step": B, "global_pairs_per_step": B * world,
                                     "parallelism": f"pair-sharded x{world}"},
                          "step_ms": step_stats(per_step), **facts})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500,
                    help="timed steps (default 500: a timed region of about 1.1 s, so that a 2 %% change is not noise)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pairs-per-gpu", type=int, default=448,
                    help="pairs resident per GPU and processed per step (448: the Sinkhorn row kernel's workgroups of "
                         "each half-batch fill whole rounds of the 256 CUs, and top-k runs two workgroups per CU)")
    ap.add_argument("--cpu-pairs", type=int, default=64, help="pairs of the live match-set parity check (0 = skip cpu_baseline)")
    ap.add_argument("--no-host-parallel", action="store_true",
                    help="cpu_baseline without the W one-thread worker processes (W = physical cores)")
    ap.add_argument("--no-extras", action="store_true", help="skip u8_ingest / streamed / latency (N = 1 extras)")
    ap.add_argument("--single-call", action="store_true",
                    help="run the step as ONE C-ABI call (mi_match_pairs); informational: no per-stage timers, so the "
                         "line carries no roofline object")
    ap.add_argument("--two-step", action="store_true",
                    help="materialise P and run the extractor on it (default: matches straight from the duals)")
    ap.add_argument("--frames", choices=["f32", "u8"], default="f32",
                    help="pixel type of the frames resident in HBM for the main line (f32 = the reference's input form; "
                         "u8 is the line `u8_ingest` reports, selectable here so that profilers can be pointed at it)")
    ap.add_argument("--workload", choices=["c2", "c3", "c3dense", "c4", "vo"], default="c2",
                    help="c2 (default, the metric's configuration); c3 = 1080x1920 K=1024 sparse pipeline; c3dense = the same "
                         "size through the dense-BAD matcher; c4 = AKAZE front end; vo = the visual-odometry model "
                         "(Shi-Tomasi+Angle matcher with the essential-matrix head)")
    ap.add_argument("--stream", action="store_true",
                    help="with --workload vo: ONE pair per call per rank (hipGraph replay, host synchronised after every call, "
                         "the record gathered to rank 0 per call) -- BASELINE configs[4] as sample/visual_odometry.py runs it")
    ap.add_argument("--pin-schedule", type=int, choices=[0, 1, 2], default=None,
                    help="pin mi_sinkhorn_dots' stream schedule on the launch stream instead of letting it tune itself "
                         "(profiling passes: every row-kernel launch then has the same size; mi_sinkhorn_dots_set_schedule)")
    ap.add_argument("--no-side", action="store_true", help="skip the side_workloads object of the default line (c3, c4, vo)")
    ap.add_argument("--dry-run", action="store_true",
                    help="control flow only: stubbed compute on CPU over gloo (tests the N > 1 launcher; not a measurement)")
    args = ap.parse_args()

    # `python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks ourselves -- BEFORE anything touches the
    # GPU in this process (nothing above does; this process only waits for its children and relays rank 0's line)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    claim_stdout()
    from onnx_image_processing_amd import _native, distributed as D
    from onnx_image_processing_amd.pytorch_model.feature_detection import (MatchExtractionWrapper,
                                                                           ShiTomasiSparseBADSinkhornMatcher)
    from onnx_image_processing_amd.synth import synth_batch_u8

    # MI_BENCH_FORCE_DIST=1: form the process group even for one rank, so that a one-GPU box runs the N-rank control flow
    # of this script -- barriers, the pipelined RCCL gather of every step's records, the max reduction -- on real RCCL
    forced = os.environ.get("MI_BENCH_FORCE_DIST", "0") == "1"
    rank, world, local = D.env_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher's WORLD_SIZE={world}")
    if args.dry_run:
        D.init(backend="gloo")
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if not os.path.exists(_native.LIB_PATH):                     # a checkout without the (git-ignored) build product
        # rank 0 builds; build.py renames each finished library onto its path (the debug library first), so the product
        # library's appearance means both are complete.  The other ranks wait for it -- there is no process group yet --
        # and give up loudly when rank 0's build failed (its marker file) or takes longer than 15 minutes.
        failed = _native.LIB_PATH + ".build_failed"
        if local == 0:
            from onnx_image_processing_amd.build import build
            try:
                if os.path.exists(failed):
                    os.remove(failed)
                build(verbose=False)
            except Exception:
                os.makedirs(os.path.dirname(failed), exist_ok=True)
                open(failed, "w").close()
                raise
        else:
            deadline = time.time() + 900
            while not os.path.exists(_native.LIB_PATH):
                if os.path.exists(failed):
                    raise SystemExit("bench.py: rank 0's build of the HIP library failed")
                if time.time() > deadline:
                    raise SystemExit("bench.py: timed out waiting for rank 0 to build the HIP library")
                time.sleep(0.5)
    _native.load()
    if args.pin_schedule is not None:
        from onnx_image_processing_amd import ops as _pin_ops
        _pin_ops.set_sinkhorn_schedule(args.pin_schedule)
    # The process group is joined AFTER the pipeline has run once (join_group below).  Measured on MI355X with a forced
    # group of one rank: with the RCCL communicator created first, every HIP stream this process creates afterwards --
    # torch's, and the helper stream mi_sinkhorn_dots overlaps its two half-batches on -- lands on the hardware queues
    # differently and the Sinkhorn call went 0.83 -> 1.16 ms (175 k instead of 202 k pairs/s per rank; 141 k with
    # GPU_MAX_HW_QUEUES=8, 198 k with 2); with the streams in place before the communicator the step costs what it costs
    # without a group, plus ~2 % for the per-step gather.
    args._forced_group = forced

    B = args.pairs_per_gpu
    if args.workload != "c2":
        if args.steps == 500:
            args.steps = 200 if args.stream else 50
        if args.stream and args.pairs_per_gpu == 448:
            args.pairs_per_gpu = 16                              # resident frame pairs the per-call loop cycles through
        return side_workload(args, rank, world, dev)
    if args.stream:
        raise SystemExit("--stream needs --workload vo")
    begin, _ = D.shard_range(B * world, rank, world)            # this rank's pairs in the global order
    a8, b8 = synth_batch_u8(1000 + begin, B, H, W)
    # float32 [0,255] (B,1,H,W): the reference's input form, resident in HBM before timing
    img1, img2 = torch.from_numpy(a8).to(dev).float(), torch.from_numpy(b8).to(dev).float()
    u8_main = args.frames == "u8"
    if u8_main:
        img1, img2 = torch.from_numpy(a8).to(dev), torch.from_numpy(b8).to(dev)
    # the reference's deployment form of "matcher + match extraction" (match_extraction_wrapper.py:82-113)
    model = MatchExtractionWrapper(ShiTomasiSparseBADSinkhornMatcher(max_keypoints=K, **CFG),
                                   max_matches=MNN["max_matches"], match_threshold=MNN["threshold"]).to(dev)
    model.fuse_extraction = not args.two_step

    gather = PipelinedGather()          # N > 1: one RCCL gather in flight under the next step's kernels; N = 1: identity

    def step():
        rec = D.pack_records(*(model.forward_single_call(img1, img2) if args.single_call else model(img1, img2)))
        return gather(rec)

    # HIP events around the roofline kernel's calls only (two per step) inside the timed region; the per-stage table
    # below comes from extra steps after it, so its 24 events per step do not sit in the measurement
    join_group(step, gather.drain, world, forced)
    # mi_sinkhorn_dots tunes its stream schedule over the first 9 calls of a shape on a stream (3 schedules x 3 trials) and
    # collects the times on the calls after them: enough untimed steps that no trial lands in the timed region, whatever
    # --warmup says; the schedule it settled on is reported in the line (`sinkhorn_schedule`)
    from onnx_image_processing_amd import ops as _ops
    for _ in range(max(0, 14 - args.warmup - (2 if (world > 1 or forced) else 0))):
        step()
    gather.drain()
    _native.enable_timing(True, only=None if args.single_call else {"mi_corner_response_pair"})
    elapsed_ms, per_step, out, own_ms = run_timed(step, args.steps, args.warmup, world, dev, torch.cuda.synchronize, gather.drain)
    schedule = _ops.sinkhorn_schedule(B, K, K, CFG["sinkhorn_iterations"])
    facts = world_facts(own_ms, args.steps, dev)
    per_call = _native.timings_ms()
    _native.enable_timing(False)
    stage_steps = 3
    stages = {}
    if not args.single_call:                                     # per-stage times (informational), outside the timed region
        _native.enable_timing(True)
        for _ in range(stage_steps):
            step()
        out = gather.drain()
        stages = _native.timings_ms()
        _native.enable_timing(False)

    if rank == 0:
        ms_per_step = elapsed_ms / args.steps
        pairs_per_step = B * world
        kernels = {k: {"ms_per_step": float(np.sum(v)) / stage_steps, "calls_per_step": len(v) / stage_steps}
                   for k, v in stages.items()}
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
            "dtype": "f32",                                      # the arithmetic type of the path (uint8 frames too)
            "data": "synthetic",
            "config": {"workload": "Shi-Tomasi(3) + NMS(r=5)/top-k + sparse BAD(512, hard) + Sinkhorn(20, eps 0.05) "
                                   "+ MNN(100, 0.1), 640x480 gray pairs, K=512 (BASELINE configs[1])"
                                   + (", issued as one mi_match_pairs call per step" if args.single_call else ""),
                       "pairs_per_gpu_per_step": B, "global_pairs_per_step": pairs_per_step,
                       "height": H, "width": W, "max_keypoints": K, "parallelism": f"pair-sharded x{world}",
                       "result_gather": ("none (one process)" if not dist.is_initialized() else
                                         ("FORCED group of one rank (MI_BENCH_FORCE_DIST=1): " if world == 1 else "") +
                                         "RCCL gather of the match records to rank 0 every step, one collective in flight under "
                                         "the next step's kernels, all completed inside the timed region"),
                       "input": ("uint8" if u8_main else "float32") + " frames resident in HBM",
                       "mean_valid_matches_per_pair": float(out[..., 5].sum().item()) / pairs_per_step},
            "step_ms": step_stats(per_step),
            **facts,                                             # ranks_seen / backend / ms_per_step_per_rank: what ran
            # the stream schedule mi_sinkhorn_dots measured fastest for this shape on this stream (include/mi355x_match.h:
            # 0 = halves on {stream, helper}, 1 = on two helpers, 2 = unsplit; -1 = still undecided)
            "sinkhorn_schedule": schedule,
            "kernels": kernels,
        }
        if not args.single_call:
            # K1 corner response: 8 algorithmic bytes per pixel (4 read + 4 written); one launch covers BOTH images of
            # every pair of this rank (mi_corner_response_pair: 2 B images), one launch per step
            k1 = per_call["mi_corner_response_pair"][args.warmup:]
            if u8_main:
                line["roofline"] = k1_roofline(float(np.mean(k1)), 5.0, 2 * B, "corner_stream_kernel<3,5,uint8> (mi_corner_response_pair: both images of every pair in one launch)",
                                               "corner_stream_kernel<3,5,true>", B)
            else:
                line["roofline"] = k1_roofline(float(np.mean(k1)), 8.0, 2 * B, "corner_stream_kernel<3,4,float> (mi_corner_response_pair: both images of every pair in one launch)",
                                               "corner_stream_kernel<3,4,false>", B)
            # informational: the other stages by their algorithmic bytes / operations
            other = {}
            if "mi_nms_candidates" in stages:
                t = float(np.mean(stages["mi_nms_candidates"]))
                other["mi_nms_candidates"] = {"bytes_per_call": 4.0 * 2 * B * H * W, "unit": "GB/s",      # one call: 2 B score maps
                                              "achieved": 4.0 * 2 * B * H * W / (t * 1e-3) / 1e9, "bound": "hbm",
                                              "peak": HBM_PEAK_GBS, "frac": 4.0 * 2 * B * H * W / (t * 1e-3) / 1e9 / HBM_PEAK_GBS}
            if "mi_sinkhorn_dots" in stages:
                t = float(np.mean(stages["mi_sinkhorn_dots"]))
                nb = 2.0 * B * K * K * CFG["sinkhorn_iterations"]
                sk_traffic, sk_src = pmc_traffic("sk_band_dots_kernel", B)
                other["mi_sinkhorn_dots (20 iterations, 2 B/element/iteration)"] = {
                    "bytes_per_call": nb, "unit": "GB/s", "achieved": nb / (t * 1e-3) / 1e9, "bound": "hbm",
                    "peak": HBM_PEAK_GBS, "frac": nb / (t * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "row_kernel_traffic_per_launch": sk_traffic, "traffic_source": sk_src,
                    "note": "memory-bound: a 22 % cut of the row kernel's VALU instructions left the call time unchanged "
                            "(same-box A/B, DESIGN.md K6); row_kernel_traffic_per_launch = FETCH + WRITE counters of one "
                            "sk_band_dots_kernel launch (one iteration of the launch's pairs) from the committed profile"}
            if "mi_cost_dots_bits" in stages:
                t = float(np.mean(stages["mi_cost_dots_bits"]))
                ops_ = 2.0 * B * K * K * NUM_PAIRS
                other["mi_cost_dots_bits (FP4 MFMA, exact popcounts)"] = {
                    "ops_per_call": ops_, "unit": "Top/s", "achieved": ops_ / (t * 1e-3) / 1e12,
                    "peak": FP4_MFMA_PEAK_TOPS, "frac": ops_ / (t * 1e-3) / 1e12 / FP4_MFMA_PEAK_TOPS,
                    "bytes_written_per_call": 2.0 * B * K * K,
                    "hbm_frac_of_its_store_stream": 2.0 * B * K * K / (t * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "note": "bound by its uint16 store stream (2 B per matrix element), not by the matrix pipe; MFMA busy counters in profiles/ (DESIGN.md K5)"}
            line["roofline_other"] = other
        if world == 1 and not args.single_call and not u8_main:
            records = out[:args.cpu_pairs].cpu().numpy() if 0 < args.cpu_pairs <= B else None
            def stage(name):                                     # progress on stderr: locates a stage that hangs or faults
                print(f"[bench] {name}", file=sys.stderr, flush=True)
            if not args.no_extras:
                stage("u8 ingest + streamed input")
                line["u8_ingest"], line["streamed"] = measure_u8_and_streamed(model, a8, b8, min(args.steps, 100))
                stage("one-pair-per-call latency")
                line["latency"] = measure_latency(model, img1, img2, a8, b8)
                stage("P materialised (the reference's two modules used literally)")
                line["p_materialised"] = measure_two_step(model, img1, img2, min(args.steps, 50))
            if args.cpu_pairs > 0:
                stage("cpu_baseline")
                line["cpu_baseline"] = cpu_baseline(min(args.cpu_pairs, B), records, parallel=not args.no_host_parallel)
            if not args.no_side:
                # BASELINE configs[2] / [3] and the VO model, driver-visible: a few steps each AFTER everything above (never
                # part of `value`), each with its own roofline object and the reference CPU path timed beside it
                del img1, img2
                torch.cuda.empty_cache()
                threads = line.get("cpu_baseline", {}).get("cores")
                side = {}
                for name in ("c3", "c4", "vo"):
                    stage(f"side workload {name}")
                    sl = run_side(name, 128, 10, 12, 0, 1, dev, extras=False)
                    side[name] = {"metric": sl["metric"], "value": sl["value"], "unit": sl["unit"], "ms_per_step": sl["ms_per_step"],
                                  "steps": sl["steps"], "pairs_per_gpu_per_step": 128, "workload": sl["config"]["workload"],
                                  "mean_valid_matches_per_pair": sl["config"].get("mean_valid_matches_per_pair"),
                                  "roofline": sl.get("roofline"), "kernels": sl["kernels"]}
                    if args.cpu_pairs > 0:
                        side[name]["cpu_baseline"] = cpu_baseline_side(name, threads)
                line["side_workloads"] = side
        emit(line)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
